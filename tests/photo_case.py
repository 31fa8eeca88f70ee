"""Shared test plumbing: build a seeded loss-path case, run the oracle on it, and drive a
dd_photo_loss-compatible entry point (the HIP library on a GPU, or the host-math test library on CPU)
through the same DDPhotoArgs."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

import synth
import oracle.ref_loss as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
from hipops import abi  # noqa: E402

BASE_COEFS = dict(p_photo=1.0, d_smooth=1e-3, d_ground=0.1, c_smooth=1e-3, c_consistency=5.0, m_sparsity=0.04, m_smooth=0.1)
MODE_OF_PHASE = {"disp_init": abi.DD_MODE_RIGID, "motion_init": abi.DD_MODE_FLOW,
                 "mask_init": abi.DD_MODE_FLOW_MASK, "fine_tune": abi.DD_MODE_FLOW_MASK}


class Case:
    """Inputs + stand-in network outputs of one phase, plus the oracle's losses / gradients for a chosen
    subset of loss terms (the others get coefficient 0)."""

    def __init__(self, phase, B, H, W, scales, seed=7, ts=None, active=("p_photo", "c_consistency"), coef_scale=1.0,
                 noise_seed=99):
        self.phase, self.B, self.H, self.W, self.scales = phase, B, H, W, list(scales)
        self.mode = MODE_OF_PHASE[phase]
        self.cmpflow, self.motmask, self.optimised, self.automask = orc.PHASES[phase]
        ts = ts or {0: [1] * B, -1: [1] * B, 1: [1] * B}
        self.inputs = synth.make_inputs(seed, B, H, W, self.scales, ts=ts)
        self.leaves = synth.make_leaves(seed, B, H, W, self.scales)
        self.coefs = {k: (v * coef_scale if k in active else 0.0) for k, v in BASE_COEFS.items()}
        self.cfg = orc.LossConfig(H, W, self.scales, coefs=self.coefs)
        self.noise = None
        if self.automask:
            g = torch.Generator().manual_seed(noise_seed)
            self.noise = {s: torch.randn(B, 2, H, W, generator=g) for s in self.scales}

    def run_oracle(self, rand_idx=None, fp64=False):
        """fp32 oracle (values + autograd gradients).  fp64=True also runs the same oracle in double precision and keeps
        its gradients in self.grad64: the yardstick for ill-conditioned cases, where the fp32 oracle's own rounding error
        is the floor under any tolerance (check_grads then judges |kernel - fp64| against |fp32 oracle - fp64|)."""
        self.grad64 = None
        if fp64:
            self.grad64 = self._oracle_fp64(rand_idx)
        self.outputs = synth.leaves_to_outputs(self.leaves, self.scales, orc.pose_matrix, self.cmpflow, self.motmask)
        self.losses = orc.loss_path(self.cfg, dict(self.inputs), self.outputs, self.phase, self.noise, rand_idx)
        for v in self.leaves.values():
            v.grad = None
        self.losses["loss"].backward()
        return self

    def _oracle_fp64(self, rand_idx):
        dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v   # noqa: E731
        leaves = {k: v.detach().double().requires_grad_() for k, v in self.leaves.items()}
        inputs = {k: dbl(v) for k, v in self.inputs.items()}
        noise = None if self.noise is None else {s: v.double() for s, v in self.noise.items()}
        grid32 = orc.pixel_grid
        orc.pixel_grid = lambda *a, **k: grid32(*a, **k).double()
        try:
            outputs = synth.leaves_to_outputs(leaves, self.scales, orc.pose_matrix, self.cmpflow, self.motmask)
            losses = orc.loss_path(self.cfg, inputs, outputs, self.phase, noise, rand_idx)
            losses["loss"].backward()
        finally:
            orc.pixel_grid = grid32
        g = {k: v.grad for k, v in leaves.items()}
        for f in (-1, 1):
            g[("T", f)] = outputs[("cam_T_cam", 0, f)].grad
        return g

    # ---------------------------------------------------------------------------------------
    def photo_buffers(self, device, materialise=True, want_grad=True, shared=False, packed=False):
        """Allocates every tensor dd_photo_loss touches on `device`; returns (args, keepalive dict).

        shared=True hands the kernel what networks.Model publishes: ONE flow field read by both frames (the sign of
        frame -1 rides on ts) and ONE mask tensor, each with one gradient buffer -- the kernel's 5-plane fast path."""
        self.shared = bool(shared) and self.mode >= 1
        B, H, W, S = self.B, self.H, self.W, len(self.scales)
        dev = torch.device(device)
        f32 = dict(dtype=torch.float32, device=dev)
        t = {}
        t["target"] = self.inputs[("color", 0, 0)].to(dev).contiguous()
        t["source"] = [self.inputs[("color", f, 0)].to(dev).contiguous() for f in (-1, 1)]
        # packed=True: the kernel also gets the pixel-interleaved copies (B,H,W,3) of the source frames (dd_pack_rgb) and gathers from those
        t["source_packed"] = None
        if packed:
            from hipops.inputs import pack_rgb
            t["source_packed"] = [pack_rgb(x) for x in t["source"]]
        t["K"] = self.inputs[("K", 0)].to(dev).contiguous()
        t["inv_K"] = self.inputs[("inv_K", 0)].to(dev).contiguous()
        t["T"] = [self.outputs[("cam_T_cam", 0, f)].detach().to(dev).contiguous() for f in (-1, 1)]
        t["ts"] = [self.inputs[("ts", f)].float().to(dev).contiguous() * (float(f) if self.shared else 1.0) for f in (-1, 1)]
        t["g_T"] = [torch.zeros(B, 4, 4, **f32) for _ in range(2)]
        t["sums"] = torch.zeros(S, abi.DD_SUMS_STRIDE, **f32)
        scales = []
        nsc = float(S)
        for s in self.scales:
            h, w = H >> s, W >> s
            d = dict(shift=s, h=h, w=w)
            d["w_photo"] = self.coefs["p_photo"] / nsc / (B * H * W)
            d["w_cons"] = (self.coefs["c_consistency"] / nsc / (2 ** s) / 2 / (B * 3 * h * w)) if self.mode == 2 else 0.0
            d["disp"] = self.outputs[("disp", 0, s)].detach().to(dev).contiguous()
            if self.mode >= 1 and self.shared:
                d["flow"] = [self.outputs[("complete_flow", 1, s)].detach().to(dev).contiguous()] * 2
                d["g_flow"] = [torch.zeros(B, 3, h, w, **f32)] * 2
            elif self.mode >= 1:
                d["flow"] = [self.outputs[("complete_flow", f, s)].detach().to(dev).contiguous() for f in (-1, 1)]
                d["g_flow"] = [torch.zeros(B, 3, h, w, **f32) for _ in range(2)]
            if self.mode == 2 and self.shared:
                d["mask"] = [self.outputs[("motion_mask", 1, s)].detach().to(dev).contiguous()] * 2
                d["g_mask"] = [torch.zeros(B, 1, h, w, **f32)] * 2
            elif self.mode == 2:
                d["mask"] = [self.outputs[("motion_mask", f, s)].detach().to(dev).contiguous() for f in (-1, 1)]
                d["g_mask"] = [torch.zeros(B, 1, h, w, **f32) for _ in range(2)]
            if self.mode == 2:
                if materialise:
                    d["out_resid"] = [torch.zeros(B, 3, h, w, **f32) for _ in range(2)]
                d["out_delta"] = [torch.zeros(B, h, w, **f32) for _ in range(2)]
            if self.automask:
                d["noise"] = self.noise[s].to(dev).contiguous()
                if materialise:
                    d["out_idsel"] = torch.zeros(B, H, W, **f32)
            d["g_disp"] = torch.zeros(B, 1, h, w, **f32)
            if materialise:
                d["out_color"] = [torch.zeros(B, 3, H, W, **f32) for _ in range(2)]
                d["out_sample"] = [torch.zeros(B, H, W, 2, **f32) for _ in range(2)]
                d["out_depth"] = torch.zeros(B, 1, H, W, **f32)
            if not want_grad:
                for k in ("g_disp", "g_flow", "g_mask"):
                    d.pop(k, None)
            scales.append(d)
        t["scales"] = scales
        args = abi.fill_photo_args(
            B=B, H=H, W=W, mode=self.mode, automask=self.automask, want_grad=want_grad,
            min_depth=self.cfg.min_depth, max_depth=self.cfg.max_depth, ssim_weight=self.cfg.ssim_weight,
            eps=1e-7, disp_thr=self.cfg.mask_disp_thrd, target=t["target"], source=t["source"], K=t["K"],
            inv_K=t["inv_K"], T=t["T"], ts=t["ts"], g_T=t["g_T"] if want_grad else None, sums=t["sums"],
            workspace=None, scales=scales, source_packed=t["source_packed"])
        if dev.type == "cuda":
            import ctypes
            from hipops import lib as L
            need = L.load().dd_photo_workspace_bytes(ctypes.byref(args))
            t["workspace"] = torch.empty(max(need // 4, 1), **f32)
            args.workspace = abi.ptr(t["workspace"])
        return args, t

    # ---------------------------------------------------------------------------------------
    def check(self, t, rtol_grad=2e-3, report=None, resid_atol=1e-6):
        """Compares buffers filled by a dd_photo_loss-compatible call with the oracle.  Returns list of failures.
        resid_atol: absolute tolerance of the residual flow c - (T P - P); the subtraction cancels |P| (up to max_depth = 100 at
        disparity 0), so its fp32 rounding floor is ~|P| x 2^-23 in ANY implementation (edge cases pass what their depth needs)."""
        fails = []
        B, H, W = self.B, self.H, self.W

        def cmp(name, got, want, rtol, atol):
            got = got.detach().cpu().double().numpy()
            want = want.detach().cpu().double().numpy()
            err = np.abs(got - want)
            tol = atol + rtol * np.abs(want)
            bad = float((err > tol).mean())
            line = "%-34s max|err| %.3e  max|ref| %.3e  frac_bad %.2e" % (name, err.max(), np.abs(want).max(), bad)
            if report is not None:
                report.append(line)
            return bad, err.max()

        o = self.outputs
        for si, s in enumerate(self.scales):
            d = t["scales"][si]
            h, w = H >> s, W >> s
            # materialised maps: pixels whose sample lands within 1e-3 px of an integer may pick the other tap pair
            if "out_color" in d:
                for fi, f in enumerate((-1, 1)):
                    bad, _ = cmp("color[%d,%d]" % (f, s), d["out_color"][fi], o[("color", f, s)], 1e-4, 2e-5)
                    if bad > 2e-3:
                        fails.append("color %d %d" % (f, s))
                    bad, _ = cmp("sample[%d,%d]" % (f, s), d["out_sample"][fi], o[("sample", f, s)], 1e-4, 2e-5)
                    if bad > 1e-3:
                        fails.append("sample %d %d" % (f, s))
                bad, _ = cmp("depth[%d]" % s, d["out_depth"], o[("depth", 0, s)], 1e-5, 1e-6)
                if bad > 0:
                    fails.append("depth %d" % s)
            photo = float(t["sums"][si, 0]) / (B * H * W)
            # loss_term/p_photo is summed over scales; recompute the oracle's per-scale value
            want = self.oracle_photo(s)
            if report is not None:
                report.append("p_photo[%d] got %.7f want %.7f" % (s, photo, want))
            if abs(photo - want) > 2e-5 * max(1.0, abs(want)):
                fails.append("p_photo %d: %g vs %g" % (s, photo, want))
            if self.automask and "out_idsel" in d:
                idsel = o["identity_selection/%d" % s]
                mism = float((d["out_idsel"].cpu() != idsel).float().mean())
                if report is not None:
                    report.append("identity_selection[%d] mismatch frac %.2e" % (s, mism))
                if mism > 1e-3:
                    fails.append("idsel %d" % s)
            if self.mode == 2:
                for fi, f in enumerate((-1, 1)):
                    if "out_resid" in d:
                        bad, _ = cmp("residual_flow[%d,%d]" % (f, s), d["out_resid"][fi], o[("residual_flow", f, s)], 1e-4, resid_atol)
                        if bad > 1e-3:
                            fails.append("resid %d %d" % (f, s))
                    want = self.oracle_cons(f, s)
                    got = float(t["sums"][si, 1 + fi]) / (B * 3 * h * w)
                    if report is not None:
                        report.append("c_consistency[%d,%d] got %.7f want %.7f" % (f, s, got, want))
                    if abs(got - want) > 1e-4 * max(1e-3, abs(want)):
                        fails.append("cons %d %d" % (f, s))
                    want_delta = self.oracle_delta(f, s)
                    bad, _ = cmp("disp_mag[%d,%d]" % (f, s), d["out_delta"][fi], want_delta, 1e-3, 1e-7)
                    if bad > 2e-3:
                        fails.append("delta %d %d" % (f, s))
        return fails

    def check_grads(self, t, report=None, frac_tol=5e-3, only_T=False, t_slack=1.0):
        """Gradient parity.  A few pixels sit on argmin ties / clamp edges / tap boundaries where fp32 rounding
        flips a discrete choice, so the criterion is: relative L2 error small AND few outliers.
        only_T: judge the pose gradients only (the per-pixel gradients go through check_grads_masked); t_slack widens the
        fp64 yardstick for the adversarial cases, where a single flipped pixel near z = 0 carries a visible share of the sum."""
        fails = []

        def cmp(name, got, want, key=None):
            got = got.detach().cpu().double()
            want = (torch.zeros_like(got) if want is None else want.detach().cpu().double())
            g64 = None if (self.grad64 is None or key is None) else self.grad64.get(key)
            if g64 is not None and float(g64.norm()) > 0:
                # judged against the fp64 oracle, in units of the fp32 oracle's own distance from it
                g64 = g64.reshape(got.shape)
                e_kernel = ((got - g64).norm() / g64.norm()).item()
                e_oracle = ((want - g64).norm() / g64.norm()).item()
                if report is not None:
                    report.append("grad %-22s vs fp64 oracle: kernel %.3e, fp32 oracle %.3e" % (name, e_kernel, e_oracle))
                if e_kernel > t_slack * max(4.0 * e_oracle, 3e-4):
                    fails.append("grad %s (%.2e vs fp32-oracle floor %.2e)" % (name, e_kernel, e_oracle))
                return
            scale = want.abs().max().item() + 1e-30
            err = (got - want).abs()
            outlier = err > 1e-3 * scale + 1e-3 * want.abs()
            bad = outlier.double().mean().item()
            keep = ~outlier
            trimmed = (((got - want) * keep).norm() / ((want * keep).norm() + 1e-30)).item()
            rel_l2 = ((got - want).norm() / (want.norm() + 1e-30)).item()
            if report is not None:
                report.append("grad %-22s rel_l2 %.3e trimmed %.3e outliers %.2e max|ref| %.3e" % (name, rel_l2, trimmed, bad, scale))
            if name.startswith("T["):
                # 12 numbers, each a sum over B*H*W pixels with cancellation: judge the vector, not its entries.  Under the
                # auto-mask a single identity/warp tie flip moves the sum: the fp32 oracle itself differs from its fp64 run by
                # 1.1e-2 on disp_init 288x512 B=1 (one flipped pixel in 147k; scripts/probe_oracle_fp64.py).
                if rel_l2 > (5e-2 if self.automask else 1e-3):
                    fails.append("grad " + name)
            # ~10x measured (profiles/r02_parity_report.txt): bulk error 1e-5..8e-5, outliers <= 2e-3 at argmin ties / static-pixel
            # threshold flips, which also carry the total rel-L2 (<= 2.2e-2 under the auto-mask, <= 4e-3 otherwise)
            elif trimmed > 5e-4 or bad > frac_tol or rel_l2 > (1e-1 if self.automask else 2e-2):
                fails.append("grad " + name)

        for si, s in enumerate([] if only_T else self.scales):
            d = t["scales"][si]
            cmp("disp[%d]" % s, d["g_disp"], self.leaves[("disp", s)].grad, ("disp", s))
            if self.mode >= 1:
                # the two frames' flow gradients are -g(-1) + g(+1) on the shared leaf
                g = d["g_flow"][0] if getattr(self, "shared", False) else -d["g_flow"][0] + d["g_flow"][1]
                cmp("flow[%d]" % s, g, self.leaves[("flow", s)].grad, ("flow", s))
            if self.mode == 2:
                m = torch.sigmoid(self.leaves[("prob", s)].detach()).to(d["g_mask"][0].device)
                g = (d["g_mask"][0] if getattr(self, "shared", False) else d["g_mask"][0] + d["g_mask"][1]) * m * (1 - m)
                cmp("prob[%d]" % s, g, self.leaves[("prob", s)].grad, ("prob", s))
        for fi, f in enumerate((-1, 1)):
            cmp("T[%d]" % f, t["g_T"][fi], self.outputs[("cam_T_cam", 0, f)].grad, ("T", f))
        return fails

    def check_grads_masked(self, t, report=None, tau=1e-3, tol=1e-4):
        """Decision-masked gradient parity (needs run_oracle(fp64=True)).  The relative-L2 figures of check_grads are carried
        by the few pixels where an argmin / auto-mask / static-pixel / clip decision flips on a last-bit difference; a
        systematic 1e-3 error would hide under them.  Here those pixels are taken out: an element is `flipped` for an
        implementation when it differs from the fp64 oracle by more than 10 tau |g64| + tau rms(g64) (a decision moved a whole
        contribution, rounding does not).  Required: on the elements where neither the fp32 oracle nor the kernel flipped the
        kernel agrees with fp64 to `tol` relative L2 (or 3x the fp32 oracle's own distance on the same elements, where the
        case's conditioning puts that above `tol`), and the kernel flips no more often than the fp32 oracle does: 2x its count
        plus two decisions' worth of elements (its rounding differs from torch's, so the flipped sets differ; ONE full-resolution
        decision reaches up to 5x5 low-resolution elements through the SSIM window and the up-sampling adjoint)."""
        assert self.grad64 is not None, "run_oracle(fp64=True) first"
        fails = []
        self.kernel_flips = 0          # elements where the kernel (and not the fp32 oracle) took another decision than fp64

        def triple(got, want32, key):
            g64 = self.grad64.get(key)
            if g64 is None or float(g64.norm()) == 0:
                return None
            got = got.detach().cpu().double().reshape(g64.shape)
            want32 = (torch.zeros_like(got) if want32 is None else want32.detach().cpu().double()).reshape(g64.shape)
            # a moved decision changes an element by a good part of its own size; rounding stays relative to the element (and,
            # where contributions cancel, to the typical element: the rms term)
            bar = 10.0 * tau * g64.abs() + tau * g64.pow(2).mean().sqrt()
            return got, want32, g64, (want32 - g64).abs() > bar, (got - g64).abs() > bar

        for si, s in enumerate(self.scales):
            d = t["scales"][si]
            items = [("disp[%d]" % s, triple(d["g_disp"], self.leaves[("disp", s)].grad, ("disp", s)))]
            if self.mode >= 1:
                g = d["g_flow"][0] if getattr(self, "shared", False) else -d["g_flow"][0] + d["g_flow"][1]
                items.append(("flow[%d]" % s, triple(g, self.leaves[("flow", s)].grad, ("flow", s))))
            if self.mode == 2:
                m = torch.sigmoid(self.leaves[("prob", s)].detach()).to(d["g_mask"][0].device)
                g = (d["g_mask"][0] if getattr(self, "shared", False) else d["g_mask"][0] + d["g_mask"][1]) * m * (1 - m)
                items.append(("prob[%d]" % s, triple(g, self.leaves[("prob", s)].grad, ("prob", s))))
            items = [(n, x) for n, x in items if x is not None]
            if not items:
                continue
            # one full-resolution decision reaches a 5x5 block of low-resolution pixels (SSIM window + up-sampling adjoint) in
            # EVERY tensor of the scale (they share the sample position): its whole reach is taken out of all of them
            moved = None
            for _, (got, w32, g64, f32, fk) in items:
                m_ = (f32 | fk).double().amax(1, keepdim=True)
                moved = m_ if moved is None else torch.maximum(moved, m_)
            moved = torch.nn.functional.max_pool2d(moved, 5, 1, 2)
            for name, (got, w32, g64, flip32, flipk) in items:
                keep = (moved == 0).expand_as(g64)
                err = (((got - g64) * keep).norm() / ((g64 * keep).norm() + 1e-300)).item()
                err32 = (((w32 - g64) * keep).norm() / ((g64 * keep).norm() + 1e-300)).item()
                f32, fk = flip32.double().mean().item(), flipk.double().mean().item()
                self.kernel_flips += int((flipk & ~flip32).sum())
                if report is not None:
                    report.append("masked %-20s kernel %.3e (fp32 oracle %.3e) on %.4f of the elements; flipped: kernel %.2e, fp32 oracle %.2e"
                                  % (name, err, err32, keep.double().mean().item(), fk, f32))
                if err > max(tol, 3.0 * err32):
                    fails.append("masked grad %s: %.2e" % (name, err))
                if fk > 2.0 * f32 + 1e-4 + 50.0 / g64.numel():
                    fails.append("masked grad %s: kernel flips %.2e vs fp32 oracle %.2e" % (name, fk, f32))
        return fails

    # ---- oracle per-scale pieces (recomputed from oracle outputs) --------------------------
    def oracle_photo(self, s):
        o, cfg = self.outputs, self.cfg
        tgt = self.inputs[("color", 0, 0)]
        rep = torch.cat([orc.reprojection_loss(o[("color", f, s)], tgt, cfg.ssim_weight) for f in (-1, 1)], 1)
        if self.automask:
            idl = torch.cat([orc.reprojection_loss(self.inputs[("color", f, 0)], tgt, cfg.ssim_weight) for f in (-1, 1)], 1)
            rep = torch.cat([idl + self.noise[s] * 0.00001, rep], 1)
        return float(rep.min(1)[0].mean())

    def oracle_cons(self, f, s):
        o = self.outputs
        valid = (o[("disp", 0, s)] > self.cfg.mask_disp_thrd).float()
        return float((valid * (1 - o[("motion_mask", f, s)]) * o[("residual_flow", f, s)].abs()).mean())

    def oracle_delta(self, f, s):
        o = self.outputs
        h, w = self.H >> s, self.W >> s
        e = orc.resize_bilinear(o[("sample_ego", f, s)].permute(0, 3, 1, 2), (h, w))
        k = orc.resize_bilinear(o[("sample_complete", f, s)].permute(0, 3, 1, 2), (h, w))
        return ((e - k) ** 2).sum(1)


def build_host_lib():
    """g++-compiles tests/hostmath/photo_host.cpp (dd_math.h + dd_pair.h on the CPU) and loads it."""
    src = os.path.join(ROOT, "tests", "hostmath", "photo_host.cpp")
    out_dir = os.path.join(ROOT, "tests", "hostmath", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libphoto_host.so")
    hdrs = [os.path.join(ROOT, "dynamo-depth_amd", "csrc", h) for h in ("dd_math.h", "dd_pair.h")] + [os.path.join(ROOT, "include", "dynamo_hip.h")]
    newest = max(os.path.getmtime(f) for f in [src] + hdrs)
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", src, "-o", so])
    lib = C.CDLL(so)
    lib.dd_photo_loss_host.restype = C.c_int
    lib.dd_photo_loss_host.argtypes = [C.POINTER(abi.DDPhotoArgs)]
    return lib
