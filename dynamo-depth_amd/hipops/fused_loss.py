"""The whole view-synthesis loss of Trainer.process_batch as ONE autograd node over HIP kernels.

Replaces Trainer.generate_images_pred + Trainer.compute_losses of the reference (Trainer.py:215-411):
~2.5k-5.1k ATen launches per step there, here one fused photometric tile kernel
for all scales, a handful of streaming regulariser kernels and one assembling kernel -- no host
synchronisation, no intermediate full-resolution tensor in HBM, hipGraph-capturable.

Values and gradients are produced in the same pass: every kernel receives the weight its raw sum
carries in `loss` and writes d(loss)/d(input) directly, so backward() is a single scale-by-grad_output.
"""
import ctypes as C
import os

import torch

from . import abi
from . import lib as L

PACKED_CALLS = [0]      # fused_loss() calls whose photometric kernel gathered from pixel-interleaved source copies (bench.py reports it)

# bench.py sets this to a list: every dd_photo_loss launch is then bracketed by HIP events on the launching stream
PROFILE_EVENTS = None
# segments.SegmentedStep (time_tile_kernel) sets this to a callable while it records the loss into hipGraphs: it is handed the
# launcher of the photometric tile kernel (stream -> None) INSTEAD of the launch, ends the graph being recorded and begins the next
# one.  At replay the step issues graph | that launch | graph: the tile kernel is then an ordinary launch on the stream, which
# dd_photo_timing brackets with HIP events exactly as it does for the host-issued step (bench.py's roofline leg)
TILE_CUT = None

# "split": dd_photo_loss + dd_reg_losses_finish (round 4's launches) even where dd_fused_loss applies -- the A/B switch of the tests
PIPELINE = os.environ.get("DD_LOSS_PIPELINE", "fused")
LAST_PIPELINE = [None]          # what the last evaluation ran: "fused5" | "split" (bench.py and the tests read it)

TERMS = abi.TERM_NAMES
_T = {name: i for i, name in enumerate(TERMS)}


class LossPlan:
    """Static description of one phase (what is active, with which coefficient)."""

    def __init__(self, *, height, width, scales, min_depth, max_depth, ssim_weight, mask_disp_thrd,
                 gp_prior, gp_tol, gp_max_it, gp_np_per_it, cmpflow, motmask, automask, optimised, coefs):
        self.H, self.W, self.scales = height, width, list(scales)
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.ssim_weight, self.mask_disp_thrd = float(ssim_weight), float(mask_disp_thrd)
        self.gp_prior, self.gp_tol, self.gp_max_it, self.gp_np_per_it = float(gp_prior), float(gp_tol), int(gp_max_it), int(gp_np_per_it)
        self.cmpflow, self.motmask, self.automask = bool(cmpflow), bool(motmask), bool(automask)
        self.coefs = {k: float(coefs[k]) for k in TERMS}
        move_depth, move_flow, move_mask = "Depth" in optimised, "CmpFlow" in optimised, "MotMask" in optimised
        c = self.coefs
        # activation rules of Trainer.compute_losses (Trainer.py:355-402)
        self.on = {
            "p_photo": True,
            "d_smooth": move_depth and c["d_smooth"] > 0,
            "d_ground": move_depth and c["d_ground"] > 0 and self.motmask,
            "c_smooth": move_flow and self.cmpflow and c["c_smooth"] > 0,
            "c_consistency": move_flow and self.cmpflow and self.motmask and c["c_consistency"] > 0,
            "m_sparsity": move_mask and self.motmask and c["m_sparsity"] > 0,
            "m_smooth": move_mask and self.motmask and c["m_smooth"] > 0,
        }
        self.mode = abi.DD_MODE_FLOW_MASK if (self.cmpflow and self.motmask) else (abi.DD_MODE_FLOW if self.cmpflow else abi.DD_MODE_RIGID)
        if self.motmask and not self.cmpflow:
            raise L.DynamoHipError("bool_MotMask without bool_CmpFlow is not a phase of the reference (Trainer.py:466-490)")


def _f32(t, name):
    if not t.is_cuda:
        raise L.DynamoHipError("%s must be on the GPU: the fused loss has no CPU implementation" % name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _FusedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, run, *diff):
        want_grad = any(ctx.needs_input_grad[1:])
        loss, terms, grads = run(diff, want_grad)
        ctx.mark_non_differentiable(terms)
        ctx.grads = grads
        return loss, terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        grads = ctx.grads
        ctx.grads = None
        live = [g for g in grads if g is not None]
        scaled = iter(torch._foreach_mul(live, g_loss)) if live else iter(())
        out = [None if g is None else next(scaled) for g in grads]
        return (None,) + tuple(out)


def fused_loss(plan, inputs, outputs, frame_ids=(0, -1, 1), noise=None, rand_idx=None, materialise=False):
    """Returns the `losses` dict of Trainer.compute_losses (same keys; values are 0-dim device tensors).

    inputs / outputs follow the reference dict contracts.  With materialise=True the
    ('color',f,s), ('sample',f,s), ('depth',0,s), ('disp_scaled',0,s), ('residual_flow',f,s) and
    'identity_selection/s' entries are written into `outputs` as a side effect."""
    lib = L.load()
    src = list(frame_ids[1:])
    if len(src) != abi.DD_NUM_SRC:
        raise L.DynamoHipError("the fused loss is built for two source frames (frame_ids [0,-1,1]), got %r" % (frame_ids,))
    H, W, scales, S = plan.H, plan.W, plan.scales, len(plan.scales)
    mode = plan.mode
    target = _f32(inputs[("color", 0, 0)], "color")
    dev = target.device
    B = target.shape[0]
    sources = [_f32(inputs[("color", f, 0)], "color") for f in src]
    # pixel-interleaved copies of the source frames, when the input side made them (Trainer.process_inputs -> hipops.inputs.pack_rgb):
    # the photometric kernel gathers one 12-byte pixel per tap instead of three planes; same values, optional
    packed = [inputs.get(("color_packed", f)) for f in src]
    if any(p is None or p.shape != (B, H, W, 3) or p.dtype != torch.float32 or p.device != dev or not p.is_contiguous() for p in packed):
        packed = None
    PACKED_CALLS[0] += packed is not None
    K, inv_K = _f32(inputs[("K", 0)], "K"), _f32(inputs[("inv_K", 0)], "inv_K")
    ts = [_f32(inputs[("ts", f)], "ts") for f in src] if mode != abi.DD_MODE_RIGID else None

    # ---- differentiable inputs, de-duplicated by identity (mask/prob tensors are shared by both frames) ----
    diff, slot = [], {}

    def reg(key, tensor):
        for i, t in enumerate(diff):
            if t is tensor:
                slot[key] = i
                return
        slot[key] = len(diff)
        diff.append(tensor)

    for s in scales:
        reg(("disp", s), outputs[("disp", 0, s)])
    for f in src:
        reg(("T", f), outputs[("cam_T_cam", 0, f)])
    # networks.Model also publishes the un-negated motion field; frames -1/+1 then read the SAME tensor and the sign
    # rides on the time step (c_f = (sign_f * ts_f) * up(field)), which halves the flow traffic and smoothness work
    gap = abs(src[0])
    shared_flow = mode != abi.DD_MODE_RIGID and src[0] == -src[1] and all(("complete_flow_field", gap, s) in outputs for s in scales)
    if mode != abi.DD_MODE_RIGID:
        for s in scales:
            for f in src:
                reg(("flow", f, s), outputs[("complete_flow_field", gap, s)] if shared_flow else outputs[("complete_flow", f, s)])
        if shared_flow:
            ts = [ts[i] * (1.0 if f > 0 else -1.0) for i, f in enumerate(src)]
    if mode == abi.DD_MODE_FLOW_MASK:
        for s in scales:
            for f in src:
                reg(("mask", f, s), outputs[("motion_mask", f, s)])
                reg(("prob", f, s), outputs[("motion_prob", f, s)])

    def run(dtensors, want_grad):
        stream = L.current_stream()
        d = [_f32(t, "network output") for t in dtensors]
        f32 = dict(dtype=torch.float32, device=dev)
        # one zero-filled arena for every accumulate-type buffer (gradients, low-res side outputs)
        offs, total = {}, 0

        def carve(name, numel):
            nonlocal total
            offs[name] = (total, numel)
            total += (numel + 63) // 64 * 64

        if want_grad:
            for i, t in enumerate(d):
                carve(("g", i), t.numel())
        if mode == abi.DD_MODE_FLOW_MASK:
            for s in scales:
                n = (H >> s) * (W >> s)
                for fi in range(2):
                    carve(("delta", fi, s), B * n)
                    if materialise:
                        carve(("resid", fi, s), B * 3 * n)
        # (zero-filled below unless the five-launch pipeline writes every element itself: dd_fused_loss_supported == 2)
        arena = torch.empty(max(total, 1), **f32)

        def view(name, shape=None):
            o, n = offs[name]
            v = arena[o:o + n]
            return v if shape is None else v.view(shape)

        def g_of(key):
            return view(("g", slot[key]), d[slot[key]].shape) if want_grad else None

        # ---- photometric kernel ------------------------------------------------------------------
        nsc = float(S)
        c = plan.coefs
        # every raw sum of the step lives in one zero-filled record: [S x DD_REG_RES_STRIDE regulariser slots | S x DD_SUMS_STRIDE
        # photometric sums]; dd_assemble_losses reads it in place
        RS = abi.DD_REG_RES_STRIDE
        res = torch.zeros(S * RS + S * abi.DD_SUMS_STRIDE, **f32)
        sums = res[S * RS:].view(S, abi.DD_SUMS_STRIDE)
        g_T = [torch.empty(B, 4, 4, **f32) for _ in range(2)] if want_grad else None
        nz = None
        if plan.automask:
            nz = noise if noise is not None else torch.randn(S, B, 2, H, W, **f32)
        sc_list, mat = [], {}
        for si, s in enumerate(scales):
            h, w = H >> s, W >> s
            e = dict(shift=s, h=h, w=w, disp=d[slot[("disp", s)]], g_disp=g_of(("disp", s)))
            e["w_photo"] = c["p_photo"] / nsc / (B * H * W)
            e["w_cons"] = (c["c_consistency"] / nsc / (2 ** s) / 2 / (B * 3 * h * w)) if plan.on["c_consistency"] else 0.0
            if mode != abi.DD_MODE_RIGID:
                e["flow"] = [d[slot[("flow", f, s)]] for f in src]
                e["g_flow"] = [g_of(("flow", f, s)) for f in src]
            if mode == abi.DD_MODE_FLOW_MASK:
                e["mask"] = [d[slot[("mask", f, s)]] for f in src]
                e["g_mask"] = [g_of(("mask", f, s)) for f in src]
                e["out_delta"] = [view(("delta", fi, s), (B, h, w)) for fi in range(2)]
                if materialise:
                    e["out_resid"] = [view(("resid", fi, s), (B, 3, h, w)) for fi in range(2)]
            if plan.automask:
                e["noise"] = _f32((nz[si] if not isinstance(nz, dict) else nz[s]).to(dev), "noise")
                if materialise:
                    e["out_idsel"] = torch.empty(B, H, W, **f32)
            if materialise:
                e["out_color"] = [torch.empty(B, 3, H, W, **f32) for _ in range(2)]
                e["out_sample"] = [torch.empty(B, H, W, 2, **f32) for _ in range(2)]
                e["out_depth"] = torch.empty(B, 1, H, W, **f32)
            sc_list.append(e)
        args = abi.fill_photo_args(
            B=B, H=H, W=W, mode=mode, automask=plan.automask, want_grad=want_grad, min_depth=plan.min_depth,
            max_depth=plan.max_depth, ssim_weight=plan.ssim_weight, eps=1e-7, disp_thr=plan.mask_disp_thrd,
            target=target, source=sources, K=K, inv_K=inv_K, T=[d[slot[("T", f)]] for f in src], ts=ts, g_T=g_T,
            sums=sums, workspace=None, scales=sc_list, source_packed=packed)
        ws = torch.empty(max(lib.dd_photo_workspace_bytes(C.byref(args)) // 4, 1), **f32)
        args.workspace = abi.ptr(ws)

        # ---- regularisers: ONE entry point for all terms and scales (four launches + the assembling one); raw sums land in
        # fixed slots of `res`, the weighted gradients are added to the arena --------------------------------------------
        asm = abi.DDAssembleArgs()
        asm.num_scales = S
        for k, name in enumerate(TERMS):
            asm.coef[k] = c[name]
        for k in range(abi.DD_MAX_RES):
            asm.term_of[k] = -1
        keep = [ws, sums, res] + (packed or [])

        def record(o, term, si, norm):
            if o >= abi.DD_MAX_RES:
                raise L.DynamoHipError("too many loss records")
            asm.term_of[o], asm.scale_of[o], asm.norm[o] = _T[term], si, norm

        reg = abi.DDRegArgs()
        reg.abi_version = abi.DD_ABI_VERSION
        reg.B, reg.num_scales = B, S
        reg.np_per_it, reg.max_it = plan.gp_np_per_it, plan.gp_max_it
        reg.tol, reg.g_prior, reg.min_depth, reg.max_depth = plan.gp_tol, plan.gp_prior, plan.min_depth, plan.max_depth
        any_reg = False
        for si, s in enumerate(scales):
            h, w = H >> s, W >> s
            rs = reg.scale[si]
            rs.h, rs.w = h, w
            color = _f32(inputs[("color", 0, s)], "color pyramid")
            keep.append(color)
            rs.img = abi.ptr(color)
            disp = d[slot[("disp", s)]]
            nsm = [0]

            def smooth(tensor, g, term, weight, normalise=False, frames=1):
                k = nsm[0]
                nsm[0] += 1
                Bq, Cq = tensor.shape[0], tensor.shape[1]
                e = rs.smooth[k]
                e.inp, e.g_inp, e.C, e.normalise, e.weight = abi.ptr(tensor), abi.ptr(g), Cq, int(normalise), weight
                div = 1.0 if term == "d_smooth" else 0.5      # per-frame terms are divided by num_frames = 2
                record(si * RS + 2 * k, term, si, 1.0 / (Bq * Cq * h * (w - 1)) / (2 ** s) * div * frames)
                record(si * RS + 2 * k + 1, term, si, 1.0 / (Bq * Cq * (h - 1) * w) / (2 ** s) * div * frames)

            if plan.on["d_smooth"]:
                smooth(disp, g_of(("disp", s)), "d_smooth", c["d_smooth"] / nsc / (2 ** s), normalise=True)
            # smoothness of the flow / mask: both frames contribute the same value when they share the tensor
            # (mask always; flow through the shared field, |smooth(-v)| = |smooth(v)|) -> one entry with the summed weight
            for term, kind in (("c_smooth", "flow"), ("m_smooth", "mask")):
                if not plan.on[term]:
                    continue
                if slot[(kind, src[0], s)] == slot[(kind, src[1], s)]:
                    smooth(d[slot[(kind, src[0], s)]], g_of((kind, src[0], s)), term, c[term] / nsc / (2 ** s), frames=2)
                else:
                    for f in src:
                        smooth(d[slot[(kind, f, s)]], g_of((kind, f, s)), term, c[term] / nsc / (2 ** s) / 2)
            if plan.on["m_sparsity"]:
                for fi, f in enumerate(src):
                    rs.delta[fi] = abi.ptr(view(("delta", fi, s)))
                    rs.delta_sum[fi] = C.c_void_p(sums.data_ptr() + 4 * (si * abi.DD_SUMS_STRIDE + 3 + fi))
                    rs.prob[fi] = abi.ptr(d[slot[("prob", f, s)]])
                    rs.g_prob[fi] = abi.ptr(g_of(("prob", f, s)))
                    rs.w_sparsity[fi] = c["m_sparsity"] / nsc / (2 ** s) / 2
                    record(si * RS + 10 + 2 * fi, "m_sparsity", si, 1.0 / (2 ** s) / 2)
            if plan.on["d_ground"]:
                rows = int(plan.gp_prior * h)
                total_pts = plan.gp_max_it * plan.gp_np_per_it
                if rand_idx is not None:
                    ridx = torch.as_tensor(rand_idx[s]).to(device=dev, dtype=torch.int32).contiguous()
                else:
                    ridx = torch.randint(0, rows * w, (B, total_pts), device=dev, dtype=torch.int32)
                plane = torch.empty(B, 3, **f32)
                invk = _f32(inputs[("inv_K", s)], "inv_K")
                keep += [ridx, plane, invk]
                rs.disp, rs.g_disp, rs.inv_K = abi.ptr(disp), abi.ptr(g_of(("disp", s))), abi.ptr(invk)
                rs.rand_idx, rs.plane = abi.ptr(ridx), abi.ptr(plane)
                rs.w_ground = -c["d_ground"] / nsc / (2 ** s) / (B * h * w)
                record(si * RS + 14, "d_ground", si, -1.0 / (B * h * w) / (2 ** s))
                if materialise:
                    mat[("ground_plane", s)] = plane
            any_reg = any_reg or nsm[0] > 0 or plan.on["m_sparsity"] or plan.on["d_ground"]
        if any_reg:
            reg.res = abi.ptr(res)
            wsr = torch.empty(max(lib.dd_reg_workspace_bytes(C.byref(reg)) // 4, 1), **f32)
            keep.append(wsr)
            reg.workspace = abi.ptr(wsr)
        # the photometric / consistency sums sit behind the regulariser records (dd_photo_loss wrote them there)
        base = S * RS
        for si, s in enumerate(scales):
            h, w = H >> s, W >> s
            record(base + si * abi.DD_SUMS_STRIDE + 0, "p_photo", si, 1.0 / (B * H * W))
            if plan.on["c_consistency"]:
                for fi in range(2):
                    record(base + si * abi.DD_SUMS_STRIDE + 1 + fi, "c_consistency", si, 1.0 / (B * 3 * h * w) / (2 ** s) / 2)
        nres = [base + S * abi.DD_SUMS_STRIDE]
        asm.n = nres[0]
        loss = torch.empty(1, **f32)
        out = torch.empty(1 + abi.DD_NUM_TERMS + abi.DD_MAX_SCALES, **f32)
        # ---- the launches, back to back (everything above was host-side preparation: the kernels of the loss then follow one
        # another on the stream without waiting for Python in between) ------------------------------------------------------
        timed = None
        if PROFILE_EVENTS is not None and not torch.cuda.is_current_stream_capturing():
            timed = [torch.cuda.Event(enable_timing=True) for _ in range(3)]      # before | after dd_photo_loss | after the assembly
            timed[0].record()
        # dd_fused_loss: warp + SSIM + smoothness + regularisers + assembly in five launches (the smoothness inside the photometric tile
        # kernel / the footprint pass) when the request qualifies -- every training step of the four phases does; DD_LOSS_PIPELINE=split
        # keeps round 4's ten launches (dd_photo_loss + dd_reg_losses_finish), which also serve the no-gradient (validation) pass
        code = lib.dd_fused_loss_supported(C.byref(args), C.byref(reg)) if (any_reg and want_grad and PIPELINE != "split") else 0
        five = code > 0
        LAST_PIPELINE[0] = "fused5" if five else "split"
        # accumulate-type buffers start from zero -- except when every gradient element is plain-stored by the five launches (code 2)
        # AND a gradient was asked of motion_prob only where the sparsity term writes it (a 46 MB fill per step at the bench shape)
        prob_written = mode != abi.DD_MODE_FLOW_MASK or plan.on["m_sparsity"]
        if not (code == 2 and prob_written and not materialise):
            arena.zero_()
        fargs = (C.byref(args), C.byref(reg), C.byref(asm), abi.ptr(loss), abi.ptr(out))
        if five:
            if TILE_CUT is not None:
                TILE_CUT(lambda on_stream, fargs=fargs: L.check(lib.dd_fused_loss_part(*fargs, on_stream, 1), "dd_fused_loss_part"))
                if timed is not None:
                    timed[1].record()
                L.check(lib.dd_fused_loss_part(*fargs, stream, 2), "dd_fused_loss_part")
            elif timed is not None:
                L.check(lib.dd_fused_loss_part(*fargs, stream, 1), "dd_fused_loss_part")
                timed[1].record()                 # (behind the tile kernel: the bracket of `dd_photo_loss_us` no longer exists as such)
                L.check(lib.dd_fused_loss_part(*fargs, stream, 2), "dd_fused_loss_part")
            else:
                L.check(lib.dd_fused_loss(*fargs, stream), "dd_fused_loss")
        else:
            if TILE_CUT is not None and want_grad:
                TILE_CUT(lambda on_stream, args=args: L.check(lib.dd_photo_loss_part(C.byref(args), on_stream, 1), "dd_photo_loss_part"))
                L.check(lib.dd_photo_loss_part(C.byref(args), stream, 2), "dd_photo_loss_part")
            else:
                L.check(lib.dd_photo_loss(C.byref(args), stream), "dd_photo_loss")
            if timed is not None:
                timed[1].record()
            if any_reg:      # four regulariser launches + the assembling kernel (which also folds the ground-hinge partials)
                L.check(lib.dd_reg_losses_finish(C.byref(reg), C.byref(asm), abi.ptr(loss), abi.ptr(out), stream), "dd_reg_losses_finish")
            else:
                L.check(lib.dd_assemble_losses(abi.ptr(res), C.byref(asm), abi.ptr(loss), abi.ptr(out), stream), "dd_assemble_losses")
        if timed is not None:
            timed[2].record()
            PROFILE_EVENTS.append((timed[0], timed[1], want_grad, timed[2]))

        grads = None
        if want_grad:
            grads = []
            for i in range(len(d)):
                grads.append(view(("g", i), d[i].shape))
            for fi, f in enumerate(src):
                grads[slot[("T", f)]] = g_T[fi]
        if materialise:
            for si, s in enumerate(scales):
                e = sc_list[si]
                outputs[("depth", 0, s)] = e["out_depth"]
                outputs[("disp_scaled", 0, s)] = 1.0 / e["out_depth"]
                for fi, f in enumerate(src):
                    outputs[("color", f, s)] = e["out_color"][fi]
                    outputs[("sample", f, s)] = e["out_sample"][fi]
                    if mode == abi.DD_MODE_FLOW_MASK:
                        outputs[("residual_flow", f, s)] = e["out_resid"][fi]
                if plan.automask:
                    outputs["identity_selection/{}".format(s)] = e["out_idsel"]
            outputs.update(mat)
        return loss.reshape(()), out, grads or []

    loss, out = _FusedLossFn.apply(run, *diff)
    losses = {"loss": loss}
    for k, name in enumerate(TERMS):
        losses["loss_term/{}".format(name)] = out[1 + k]
        losses["loss_coef/{}".format(name)] = plan.coefs[name]
    for si, s in enumerate(scales):
        losses["loss_term/{}".format(s)] = out[1 + abi.DD_NUM_TERMS + si]
    return losses
