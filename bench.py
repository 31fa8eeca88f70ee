#!/usr/bin/env python
"""Headline benchmark: training images/s (192x640 triplets) of the Dynamo-Depth step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = Trainer.process_batch (3x depth net, 2x pose net, motion encoder + 2 motion decoders through MIOpen /
hipBLASLt, then the fused HIP view-synthesis loss) + backward + Adam, on one synthetic batch already resident in HBM.
Workload = BASELINE.json configs[1]: KITTI shape 192x640, litemono, batch 12 per GPU, fp32, phase fine_tune (all
networks optimised, every loss term active -- 20 of the schedule's 27 epochs).  Weak scaling: every rank processes its
own batch; the only collective is DDP's gradient all-reduce over RCCL/xGMI.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
import miopen_env  # noqa: E402

miopen_env.setup()      # HIP_FORCE_DEV_KERNARG (206 vs 192 img/s with 0), MIOpen find mode + the shipped find-db records

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (guide: ~6.3 TB/s achievable with a float4 copy)


def algorithmic_bytes(B, H, W, scales, motion):
    """per scale, inputs read once in forward and once in backward, gradients written once (fp32)."""
    N = H * W
    M = 1 if motion else 0
    total = single = 0
    for s in scales:
        n = N >> (2 * s)
        fwd = 4 * (9 * N + n * (1 + (3 if s > 0 else 0)) + M * 2 * 5 * n)
        bwd = fwd + 4 * (n + M * 2 * 5 * n)
        total += fwd + bwd
        # what the single-pass fused kernel has to move: frames + disp (+flow, mask) once, gradients (+disp_mag) once
        single += 4 * (9 * N + n + M * 2 * 4 * n) + 4 * (n + M * 2 * 4 * n + M * 2 * n)
    return B * total, B * single


def make_batch(trainer, seed):
    from torch.utils.data import DataLoader
    ds = trainer.get_dataset(["synthetic {}".format(i) for i in range(trainer.B)], is_train=False, seed=seed)
    batch = next(iter(DataLoader(ds, batch_size=trainer.B)))
    trainer.upload_inputs(batch)                      # the batch is HBM-resident before timing; the target pyramid
    return batch                                      # (Trainer.apply_img_resize) is built inside every timed step


def _queues_found():
    """How many of the step's side streams the picker could place on hardware queues of their own (hipops/queues.py)."""
    try:
        from hipops import queues
        return queues.found()
    except Exception:
        return None


def _small_conv_calls():
    try:
        from hipops import functions as HF
        return HF.small_conv_calls()
    except Exception:
        return 0


def _mlp_fused_calls():
    try:
        from hipops import functions as HF
        return HF.mlp_fused_calls()
    except Exception:
        return 0


def _mlp_recompute_calls():
    try:
        from hipops import functions as HF
        return HF.mlp_recompute_calls()
    except Exception:
        return 0


def _mlp_calls():
    try:
        from hipops import functions as HF
        return HF.mlp_calls()
    except Exception:
        return 0


def _mfma_conv_calls():
    try:
        from hipops import functions as HF
        return HF.mfma_conv_calls()
    except Exception:
        return 0


def _mfma_products():
    try:
        from hipops import functions as HF
        return HF.mfma_products()
    except Exception:
        return 6


def _half_conv_calls():
    try:
        from hipops import functions as HF
        return HF.half_conv_calls()
    except Exception:
        return 0


def _conv_mfma_roofline(reps=20):
    """Second roofline object, for the kernel that takes most of the step's time since round 5: dd_conv3x3_mfma's forward at the motion
    decoders' half-resolution shape (12 x 64 -> 64 x 96 x 320), timed here with events on the stream it is launched on.  bound "mfma":
    achieved = 2 * pixels * 9 * cin * cout / time (the fp32 FLOPs of the convolution), peak = the dense bf16 MFMA peak of
    MI355X_MICROARCH.md (2 500 TFLOP/s) / 6 partial products per fp32 multiply-add."""
    try:
        import torch
        from hipops import lib as L
        from hipops.functions import _p, _ws_bytes, _dense_nhwc, _nhwc_empty
        lib = L.load()
        B, cin, cout, H, W = 12, 64, 64, 96, 320
        x = _dense_nhwc(torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last))
        w = torch.randn(cout, cin, 3, 3, device="cuda") / 24
        b = torch.randn(cout, device="cuda")
        pf = torch.empty(_ws_bytes("dd_conv3x3_mfma_pack_bytes", cout, cin) // 4, device="cuda")
        sw = w.stride()
        st = L.current_stream()
        L.check(lib.dd_conv3x3_mfma_pack(_p(w), sw[0], sw[1], sw[2], sw[3], cout, cin, _p(pf), None, st), "dd_conv3x3_mfma_pack")
        y = _nhwc_empty(B, cout, H, W, x.device)
        for _ in range(3):
            L.check(lib.dd_conv3x3_mfma(_p(x), _p(pf), _p(b), B, H, W, cin, cout, 1, _p(y), st), "dd_conv3x3_mfma")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)       # the kernel goes out on torch's current stream
        e0.record()
        for _ in range(reps):
            lib.dd_conv3x3_mfma(_p(x), _p(pf), _p(b), B, H, W, cin, cout, 1, _p(y), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        flops = 2.0 * B * H * W * 9 * cin * cout
        peak = 2500.0 / 6.0
        tf = flops / us * 1e-6
        return {"bound": "mfma", "kernel": "dd::cm::conv_mfma_kernel<2> (dd_conv3x3_mfma forward, 12x64->64x96x320)", "achieved": round(tf, 1), "peak": round(peak, 1),
                "unit": "TFLOP/s (fp32-equivalent: six bf16 MFMA partial products per multiply-add)", "frac": round(tf / peak, 4),
                "avg_launch_us": round(us, 1), "launches_timed": reps, "traffic": None,
                "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": 2 * B * H * W * cin * 4}
    except Exception as exc:          # the second object is information, never a reason to lose the bench line
        return {"error": repr(exc)[:200]}



def synthetic_net_outputs(B, H, W, scales, cmpflow, motmask, device, low_frequency=True, seed=0):
    """SURVEY.md 8(d) stand-in network outputs for the loss-only figures: disp_s ~ 0.05 + 0.9 U, axis-angle ~ 0.01 N, translation ~ 0.1 N ->
    transformation_from_parameters(invert=True), complete_flow_s ~ 0.05 N, motion_prob_s ~ N(0,1), motion_mask = sigmoid(prob); what
    networks.Model publishes (ONE flow field / mask / prob tensor for both frames).  low_frequency: the same distributions drawn on an
    8x coarser grid and bilinearly up-sampled -- network outputs are smooth; the literal per-pixel draw scatters every tap of the warp
    and is reported beside it (`frac_white_noise_fields`).  Seeded CPU generator: the same fields on every box."""
    import torch.nn.functional as F
    from hipops.functions import PoseMatrixFn
    gen = torch.Generator().manual_seed(1000 + seed)

    def field(draw, c, h, w):
        if low_frequency and w >= 16:
            return F.interpolate(draw(B, c, (h + 7) // 8 + 1, (w + 7) // 8 + 1), (h, w), mode="bilinear", align_corners=False)
        return draw(B, c, h, w)

    def leaf(t):
        return t.to(device).contiguous().requires_grad_()
    rand = lambda *sz: torch.rand(*sz, generator=gen)        # noqa: E731
    randn = lambda *sz: torch.randn(*sz, generator=gen)      # noqa: E731
    out = {}
    for s in scales:
        h, w = H >> s, W >> s
        out[("disp", 0, s)] = leaf(0.05 + 0.9 * field(rand, 1, h, w))
        if cmpflow:
            fl = leaf(0.05 * field(randn, 3, h, w))
            out[("complete_flow_field", 1, s)] = fl
            out[("complete_flow", 1, s)], out[("complete_flow", -1, s)] = fl, -fl
        if motmask:
            prob = leaf(field(randn, 1, h, w))
            mask = torch.sigmoid(prob)
            for f in (-1, 1):
                out[("motion_prob", f, s)], out[("motion_mask", f, s)] = prob, mask
    for f in (-1, 1):
        aa, tt = leaf(0.01 * randn(B, 1, 3)), leaf(0.1 * randn(B, 1, 3))
        out[("axisangle", 0, f)], out[("translation", 0, f)] = aa, tt
        out[("cam_T_cam", 0, f)] = PoseMatrixFn.apply(aa, tt, True)
    return out


def allreduce_budget(seg_step, world):
    """What the one collective of the step costs IF NOTHING HIDES IT: the flat gradient buffer through RCCL's ring
    all-reduce, 2 (n-1)/n x bytes per rank over the ring's bus bandwidth -- 150 GB/s (one xGMI link's worth) to 300 GB/s (what RCCL
    reaches on a fully connected MI300-class node); scripts/scale.sh prints each N's measured ms/step beside it."""
    buf = getattr(seg_step, "flat_all", None) if seg_step is not None else None
    if buf is None:
        return None
    nbytes = int(buf.numel()) * 4
    out = {"gradient_bytes": nbytes, "ranks": world}
    for n in sorted(set([2, 4, 8, max(world, 2)])):
        vol = 2.0 * (n - 1) / n * nbytes
        out["exposed_ms_at_%d_gpus" % n] = [round(vol / 300e9 * 1e3, 3), round(vol / 150e9 * 1e3, 3)]
    return out


def reference_cpu_record():
    """The unmodified reference cannot travel to the GPU box; its timing in the build container is a committed record
    (scripts/time_reference_cpu.py -> profiles/rNN_reference_cpu_build_container.txt, the newest round's file): parsed, not restated."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu_build_container.txt")))
    if not files:
        return {"error": "no profiles/r*_reference_cpu_build_container.txt"}
    rec = {"source": "scripts/time_reference_cpu.py -> " + os.path.relpath(files[-1], ROOT)}
    try:
        for ln in open(files[-1]):
            m = re.search(r"(\d+) threads: median ([0-9.]+) s\s+\(([0-9.]+) img/s\)", ln)
            if not m:
                continue
            key = "loss_path_fwd_bwd" if "loss path" in ln else ("full_step" if "full step" in ln else None)
            if key:
                rec[key + "_img_per_s"], rec[key + "_median_s"], rec["threads"] = float(m.group(3)), float(m.group(2)), int(m.group(1))
                rec[key + "_what"] = ln.split(":")[0].strip()
    except OSError as exc:
        rec["error"] = repr(exc)
    return rec


def note(msg):
    print("[bench {:7.1f}s] {}".format(time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def cpu_baseline_guarded(opt_args, phase, sample_batch, hard_limit_s=150):
    """Runs cpu_baseline() in a child process with a hard time limit so that it can never take the bench line down."""
    import subprocess
    code = ("import sys, json; sys.path.insert(0, {root!r}); sys.path.insert(0, {prod!r}); import bench; "
            "print('CPUBASE ' + json.dumps(bench.cpu_baseline({args!r}, {phase!r}, {sb})))").format(
                root=ROOT, prod=os.path.join(ROOT, "dynamo-depth_amd"), args=list(opt_args), phase=phase, sb=sample_batch)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    try:
        res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             timeout=hard_limit_s, env=env, cwd=ROOT)
        for ln in res.stdout.splitlines():
            if ln.startswith("CPUBASE "):
                return json.loads(ln[8:])
        return {"error": "cpu baseline produced no result (rc={}): {}".format(res.returncode, res.stderr[-400:])}
    except subprocess.TimeoutExpired:
        return {"error": "cpu baseline exceeded {} s".format(hard_limit_s)}


def cpu_baseline(opt_args, phase, sample_batch, budget_s=20.0):
    """The reference's CPU path restated: this tree's networks on CPU + the oracle loss (oracle/ref_loss.py) + Adam,
    on a bounded sample of the same workload, all host cores."""
    sys.path.insert(0, ROOT)
    import oracle.ref_loss as orc
    import networks
    from options import DynamoOptions
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, int(os.environ.get("DD_CPU_BASELINE_THREADS", "16")))   # batch-2 convs stop scaling (and thrash) beyond ~16 threads
    torch.set_num_threads(cores)
    opt = DynamoOptions().parse(args=opt_args)
    model = networks.Model(opt)
    cmp, mot, nets, _ = orc.PHASES[phase]
    model.bool_CmpFlow, model.bool_MotMask = cmp, mot
    model.set_train()
    params = model.parameters_by_names(list(nets))
    adam = torch.optim.Adam(params, 1e-4)
    base = {k[2:]: v for k, v in vars(opt).items() if k[:2] == "g_"}
    cfg = orc.LossConfig(opt.height, opt.width, opt.scales, coefs=base)
    cfg.redraw_singular = True          # redraw a degenerate RANSAC sample instead of aborting the whole step (see oracle/ref_loss.py)
    from datasets import SyntheticTriplets
    from torch.utils.data import DataLoader
    ds = SyntheticTriplets(height=opt.height, width=opt.width, num_scales=len(opt.scales), length=sample_batch)
    batch = next(iter(DataLoader(ds, batch_size=sample_batch)))
    import torch.nn.functional as F
    for s in opt.scales:
        if s:
            batch[("color", 0, s)] = F.interpolate(batch[("color", 0, s - 1)], (opt.height >> s, opt.width >> s), mode="bicubic",
                                                   align_corners=False, antialias=True).clamp(0, 1)
    steps, attempts, t0, dt = 0, 0, time.time(), 0.0
    while attempts < 200:
        attempts += 1
        t_try = time.time()
        adam.zero_grad()
        outputs = model(batch)
        try:
            losses = orc.loss_path(cfg, batch, outputs, phase)
        except torch.linalg.LinAlgError:
            # the reference's RANSAC inverts a 3x3 built from 5 random points with torch.inverse and raises when a draw is
            # degenerate (tools.py:152); redraw.  (The HIP kernel yields a non-finite candidate that never wins instead.)
            continue
        losses["loss"].backward()
        adam.step()
        steps += 1
        dt += time.time() - t_try                      # redrawn attempts are not charged to the baseline
        if time.time() - t0 > budget_s:
            break
    if steps == 0:
        return {"error": "no RANSAC draw succeeded in {} attempts".format(attempts)}
    return {"value": sample_batch * steps / dt, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "{} full training steps (networks + oracle loss + Adam, fp32) at batch {} of the same {}x{} {} workload, {:.1f} s ({} attempts)".format(
                steps, sample_batch, opt.height, opt.width, phase, dt, attempts)}


def pmc_traffic(a, opt, motion):
    """`roofline.traffic`: bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate passes and corrected as MI355X_MICROARCH.md prescribes -- profiles/README.md). Counters cannot be collected
    from inside this process, so the figure is the one measured on this kernel revision and workload shape; null otherwise."""
    path = os.path.join(ROOT, "profiles", "photo_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        key = "{}x{}x{} scales={} phase={}".format(a.batch, opt.height, opt.width, len(opt.scales), a.phase)
        hit = rec["workloads"].get(key)
        if hit is None:
            return {"traffic": None}
        return {"traffic": hit["traffic_bytes_per_launch"], "traffic_source": "profiles/photo_traffic.json ({})".format(rec["collected"])}
    except (OSError, ValueError, KeyError):
        return {"traffic": None}


def main():
    if os.environ.get("DD_FAULT_AFTER"):            # debugging aid: dump every thread's Python stack after N seconds (a hung run)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DD_FAULT_AFTER"]), repeat=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--phase", default="fine_tune", choices=["disp_init", "motion_init", "mask_init", "fine_tune"])
    ap.add_argument("--depth_model", default="litemono")
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--mode", default="auto", choices=["auto", "eager", "graph"],
                    help="graph = the step replayed from per-network hipGraphs on their own streams (segments.py; train.py's default); "
                         "eager = every kernel issued by the host; auto (default) = time both during the warm-up and run the faster one")
    ap.add_argument("--single_stream", dest="multi_stream", action="store_false",
                    help="default: the independent network branches of the forward (3 depth passes, poses, motion encoder) run on separate HIP streams")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_miopen_find", dest="miopen_find", action="store_false", default=None,
                    help="default (train.py's): MIOpen Find on unless the workload's problems have shipped find-db records (miopen_db/recorded.json)")
    ap.add_argument("--miopen_find", dest="miopen_find", action="store_true", help="MIOpen Find on whatever the shipped records cover")
    ap.add_argument("--nchw", dest="channels_last", action="store_false",
                    help="default: NHWC networks (MIOpen's fp32 implicit-GEMM kernels are NHWC; NCHW inserts transposes)")
    ap.add_argument("--amp", default="none", choices=["none", "bf16", "fp16"], help="NOT the headline: reduced-precision networks (loss stays fp32)")
    ap.add_argument("--matmul_precision", default=None, choices=["highest", "high", "medium"],
                    help="NOT the headline unless 'highest' (the default): torch.set_float32_matmul_precision for the run -- 'high' computes the fp32 "
                         "3x3 convolutions from three bf16 partial products instead of six (bf16x3), 'medium' from one")
    ap.add_argument("--no_fused_loss", action="store_true", help="ablation: operator-by-operator loss path")
    ap.add_argument("--stats_only_side_frames", action="store_true",
                    help="NOT the headline: frames -1/+1 through the depth encoder only (same weights, statistics and losses; the reference also decodes them)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the loss path has no CPU implementation")
    # DD_BENCH_BACKEND=gloo: smoke test of the N > 1 code path on a one-GPU box (all ranks on cuda:0, CPU collectives);
    # RCCL refuses two ranks on one device.  Not a measurement.
    backend = os.environ.get("DD_BENCH_BACKEND", "nccl")
    share_device = backend != "nccl"
    torch.cuda.set_device(0 if share_device else local_rank)
    # DD_BENCH_FORCE_DIST=1: the N > 1 code path (process group, per-segment all-reduce, captures beside RCCL's watchdog thread) with
    # ONE rank -- RCCL accepts a world of one, which is how a one-GPU box exercises it (tests/test_bench_gpu.py)
    dist_on = world > 1 or os.environ.get("DD_BENCH_FORCE_DIST", "0") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world == 1:
            dist.init_process_group(backend=backend, rank=0, world_size=1)
        else:
            dist.init_process_group(backend=backend)

    from options import DynamoOptions
    from Trainer import Trainer
    from hipops import fused_loss as FL
    from hipops import lib as HL
    import ctypes as C
    opt_args = ["-d", a.dataset, "--depth_model", a.depth_model, "-b", str(a.batch), "--weights_init", "scratch", "--synthetic",
                "--num_workers", "0", "--log_dir", "/tmp/dd_bench_logs", "--no_train_vis"]
    # the fast configuration (channels-last, multi-stream, MIOpen Find, per-network hipGraphs) is train.py's default: the
    # flags below only switch parts of it OFF for ablations
    if a.no_fused_loss:
        opt_args.append("--no_fused_loss")
    if a.stats_only_side_frames:
        opt_args.append("--stats_only_side_frames")
    if a.mode == "eager":
        opt_args.append("--no_hip_graph")
    if not a.channels_last:
        opt_args.append("--nchw")
    if not a.multi_stream:
        opt_args.append("--single_stream")
    if a.miopen_find is not None:
        opt_args.append("--miopen_find" if a.miopen_find else "--no_miopen_find")
    if a.amp != "none":
        opt_args += ["--amp", a.amp]
    if a.matmul_precision:
        opt_args += ["--matmul_precision", a.matmul_precision]
    opt = DynamoOptions().parse(args=opt_args)
    opt.print_opt = False
    opt.local_world_size, opt.ddp = world, dist_on
    opt.local_rank = local_rank
    opt.cuda_ids = [0] * max(world, 1) if share_device else list(range(max(world, 1)))
    torch.manual_seed(1234 + rank)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the Trainer's banner goes to stderr: stdout carries the ONE JSON line and nothing else
        tr = Trainer(opt)
    # The timed step is train.py's step, graph for graph.  (DD_BENCH_SPLIT_LOSS=1: round 3's instrumented step -- the loss as
    # graph | tile kernel launched by the host | graph, so that HIP events bracket the kernel inside the timed region; it costs two
    # graph-launch latencies per step and is not what train.py runs.)
    tr.time_tile_kernel = os.environ.get("DD_BENCH_SPLIT_LOSS", "0") == "1"
    tr.num_steps_per_epoch = 1000
    tr.setup_phase(a.phase)
    tr.bool_automask = a.phase == "disp_init"
    tr.step = 1000                                   # past the ramp: loss weights at their full values
    tr.set_train()
    batch = make_batch(tr, seed=rank)
    motion = a.phase in ("mask_init", "fine_tune")

    def one_step():
        return tr.train_step(dict(batch))            # a fresh dict without the pyramid keys: every step builds the target pyramid

    note("trainer built; warm-up (first step compiles / selects the MIOpen kernels)")
    FL.PROFILE_EVENTS = []
    hip = HL.load()
    HL.check(hip.dd_photo_timing(1), "dd_photo_timing")          # HIP events around photo_tile_kernel alone, inside the library
    mode = "eager" if a.no_fused_loss and a.mode != "graph" else a.mode
    capture_fallback = None              # auto mode: why the replayed step was not available, if it was not
    if mode == "auto":
        # both ways of issuing the step, W warm-up steps each (all untimed); the faster one is then timed for K steps
        def timed(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                one_step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        opt.hip_graph = False
        for i in range(a.warmup):
            one_step()
            if i == 0:
                torch.cuda.synchronize()
                note("first eager step done (MIOpen solver selection, code objects of every kernel loaded)")
        t_eager = timed(max(a.warmup, 3))
        note("eager warm-up and probe done")
        opt.hip_graph = True
        try:
            for i in range(2):
                one_step()                       # captures, then replays
                if i == 0:
                    torch.cuda.synchronize()
                    note("per-network graphs captured")
            t_graph = timed(max(a.warmup, 3))
        except Exception as err:                 # a capture this stack refuses (never seen on one GPU, nor beside a one-rank RCCL group)
            capture_fallback = "{}: {}".format(type(err).__name__, str(err)[:400])
            note("the replayed step is not available here ({}); timing the eager step".format(capture_fallback))
            t_graph = float("inf")
            opt.hip_graph = False
            tr.drop_graphs()
            torch.cuda.synchronize()
        if dist_on:                          # every rank must take the same path: decide on the slowest rank's numbers
            t = torch.tensor([t_eager, min(t_graph, 1e9)], device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_eager, t_graph = float(t[0]), float(t[1])
        # the replayed step is the default of train.py and the steadier of the two: its host side is ~18 ms of graph launches
        # against ~45-50 ms of Python for the eager step, which sits within a few per cent of the GPU time and loses whenever
        # the host stumbles (one probe of the round: eager 47.95 ms, then 55.3 ms over the timed steps).  Eager only when it
        # is clearly faster.
        mode = "graph" if t_graph <= 1.05 * t_eager else "eager"
        opt.hip_graph = mode == "graph"
        note("auto mode: eager {:.2f} ms/step, hipGraph replay {:.2f} ms/step -> {}".format(t_eager * 1e3, t_graph * 1e3, mode))
        auto_note = {"eager_ms": round(t_eager * 1e3, 3), "graph_ms": round(t_graph * 1e3, 3) if t_graph < 1e8 else None}
    else:
        auto_note = None
        opt.hip_graph = mode == "graph"
        for _ in range(a.warmup):
            one_step()
    torch.cuda.synchronize()
    warm_events = FL.PROFILE_EVENTS
    FL.PROFILE_EVENTS = [] if mode == "eager" else None
    tile_us, tile_n = C.c_float(0), C.c_int(0)
    HL.check(hip.dd_photo_timing_read(C.byref(tile_us), C.byref(tile_n), 1), "dd_photo_timing_read")   # eager warm-up launches
    # (the timer stays on in both modes: the replayed step launches the tile kernel from the host between two graphs)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    seg_step = tr._graph if mode == "graph" and hasattr(tr._graph, "loss_events") else None
    if seg_step is not None and getattr(seg_step, "reduce_mode_chosen", "end") is None:
        # DD_SEG_REDUCE=auto (the default with more than one rank): the replayed step is still probing where the gradient all-reduce
        # goes (behind the last backward graph, or one collective per network behind its backward graph); let it decide before the
        # timed region -- untimed replays, same on every rank
        for _ in range(3 * seg_step.AUTO_PROBE):
            if seg_step.reduce_mode_chosen is not None:
                break
            one_step()
        torch.cuda.synchronize()
        note("gradient all-reduce placement: {} (probe, ms/step, max over ranks: {})".format(seg_step.reduce_mode_chosen, seg_step.reduce_probe_ms))
        if dist_on:
            dist.barrier()
    if seg_step is not None:
        seg_step.loss_events = []            # an event pair around the loss graph of every timed step (no host sync)
    note("warm-up done; timing {} steps".format(a.steps))
    if seg_step is not None:
        seg_step.host_wait_s = 0.0
    t0 = time.perf_counter()
    trace = [] if os.environ.get("DD_BENCH_TRACE_LOSS") == "1" else None     # diagnostics: one host sync per step
    for _ in range(a.steps):
        outputs, losses = one_step()
        if trace is not None:
            trace.append(round(float(losses["loss"].detach()), 5))
    if trace is not None:
        note("loss per timed step: {}".format(trace))
    t_enqueued = time.perf_counter() - t0          # host side done; the rest is the GPU draining its queue
    if seg_step is not None:
        t_enqueued -= seg_step.host_wait_s          # the replayed step keeps the host at most two steps ahead of the GPU: that wait is idle time
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    enqueue_per_rank = [round(t_enqueued / a.steps * 1e3, 3)]
    if dist_on:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        mine = torch.tensor([t_enqueued / a.steps * 1e3], device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        enqueue_per_rank = [round(float(x), 3) for x in every]
    loss_val = float(losses["loss"].detach())
    if seg_step is not None and seg_step.timing:             # DD_SEG_TIMING=1: where the graphs of the last step sat on the GPU
        rows = []
        for _ in range(5):
            one_step()
            rows.append(seg_step.timeline())
        for i, (name, _, _) in enumerate(rows[0]):
            b, e = sum(r[i][1] for r in rows) / len(rows), sum(r[i][2] for r in rows) / len(rows)
            note("  segment {:<12s} start {:7.2f} ms  end {:7.2f} ms  ({:6.2f} ms)".format(name, b, e, e - b))

    probe_n = 0
    legs = {}
    cmp_on, mot_on = bool(tr.base_model.bool_CmpFlow), bool(tr.base_model.bool_MotMask)

    def loss_leg(outputs, n):
        """n host-issued evaluations of the whole loss path (values AND gradients), alone on the stream: HIP events around the tile kernel
        (dd_photo_timing, inside the library, on the launching stream) and around the whole path; the first two are dropped."""
        torch.cuda.synchronize()
        FL.PROFILE_EVENTS = []
        us, cnt = C.c_float(0), C.c_int(0)
        HL.check(hip.dd_photo_timing_read(C.byref(us), C.byref(cnt), 0), "dd_photo_timing_read")      # drop what came before
        for _ in range(n + 2):
            tr.fused_losses(seg_step.batch, outputs)
        torch.cuda.synchronize()
        ev = [e for e in FL.PROFILE_EVENTS[2:] if e[2]]
        FL.PROFILE_EVENTS = None
        HL.check(hip.dd_photo_timing_read(C.byref(us), C.byref(cnt), 2), "dd_photo_timing_read")
        if not ev or cnt.value <= 0:
            return None
        if os.environ.get("DD_BENCH_DEBUG_LEGS") == "1":
            note("leg: tile {:.1f} us x {}; per evaluation [event 0 -> behind the tile kernel | -> behind the last launch] (us): {}".format(
                us.value, cnt.value, ["%.0f|%.0f" % (e[0].elapsed_time(e[1]) * 1e3, e[0].elapsed_time(e[3]) * 1e3) for e in ev]))
        # (the event brackets span host-issued launches: the MEDIAN over the evaluations -- one host hiccup between two launches of one
        # evaluation, seen once in a sweep as 18 ms inside a 0.4 ms bracket, must not pass for kernel time; the tile kernel's own bracket is
        # inside the library around the single launch and stays a mean)
        import statistics
        return {"tile_us": float(us.value), "launches": int(cnt.value),
                "photo_us": statistics.median(e[0].elapsed_time(e[1]) for e in ev) * 1e3,
                "path_us": statistics.median(e[0].elapsed_time(e[3]) for e in ev) * 1e3}

    if seg_step is not None and not tr.time_tile_kernel:
        # Roofline leg of the replayed step: inside a graph the tile kernel cannot be bracketed by events, so the loss path is evaluated
        # again directly behind the timed region, host-issued and alone on the stream, same kernels, same process, same batch of frames:
        #   "synthetic": on SURVEY.md 8(d)'s synthetic network outputs (low-frequency fields) -- what `roofline.frac` is quoted on;
        #   "white_noise": the same distributions drawn per pixel (every tap of the warp scattered);
        #   "identity": on the LAST timed step's network outputs, still in the graphs' static buffers -- random-init networks publish
        #    near-zero flow and a constant disparity, their warps are the identity plus the ego-motion and a wave's taps are contiguous:
        #    the kindest input, kept as `frac_identity_warps`.
        probe_n = max(a.steps, 10)
        legs["identity"] = loss_leg(seg_step.loss_outputs, probe_n)
        for name, low in (("synthetic", True), ("white_noise", False)):
            legs[name] = loss_leg(synthetic_net_outputs(a.batch, opt.height, opt.width, opt.scales, cmp_on, mot_on, tr.device, low_frequency=low), probe_n)
    events = FL.PROFILE_EVENTS if FL.PROFILE_EVENTS else warm_events[1:]
    FL.PROFILE_EVENTS = None
    kern_ms = [ev[0].elapsed_time(ev[1]) for ev in events if ev[2]]
    path_ms = [ev[0].elapsed_time(ev[3]) for ev in events if ev[2]]       # photometric + regularisers + assembly, launch to launch
    timed_in = "timed region" if mode == "eager" else "eager warm-up steps"
    replay_note = {}
    if mode == "eager":
        HL.check(hip.dd_photo_timing_read(C.byref(tile_us), C.byref(tile_n), 0), "dd_photo_timing_read")   # the timed region's launches
        HL.check(hip.dd_photo_timing(0), "dd_photo_timing")
    elif seg_step is not None:
        # The replayed step: every timed step recorded an event pair around its loss graph (stream time inside the step, beside the
        # other streams' kernels); the kernel times come from the host-issued evaluations behind the timed region (or, with
        # DD_BENCH_SPLIT_LOSS=1, from the tile kernel launched by the host between two graphs of the loss inside the timed region).
        graph_ms = [e0.elapsed_time(e1) for e0, e1 in seg_step.loss_events]
        seg_step.loss_events = None
        if not probe_n:
            HL.check(hip.dd_photo_timing_read(C.byref(tile_us), C.byref(tile_n), 0), "dd_photo_timing_read")
        HL.check(hip.dd_photo_timing(0), "dd_photo_timing")
        timed_in = ("{} host-issued evaluations of the loss path (forward AND gradients) on SURVEY 8(d) synthetic network outputs "
                    "(disp 0.05+0.9U, axis-angle 0.01 N, translation 0.1 N, flow 0.05 N, prob N(0,1); drawn on an 8x coarser grid and bilinearly "
                    "up-sampled) and this step's batch of frames, directly behind the timed region, alone on the stream (the timed step itself is "
                    "train.py's: the loss is one graph)".format(probe_n) if probe_n else
                    "timed region (DD_BENCH_SPLIT_LOSS=1: the tile kernel is launched by the host between two graphs of the loss; the step's own network outputs)")
        if graph_ms:
            replay_note["loss_path_replayed_us"] = round(sum(graph_ms) / len(graph_ms) * 1e3, 1)
            replay_note["frac_loss_path_replayed"] = None       # filled below
            replay_note["loss_path_timed_in"] = "loss_path_us: the host-issued evaluations; loss_path_replayed_us: event pair around the loss graph of every timed step (the step's own network outputs, beside the other streams' kernels)"
    conv_bytes, single_bytes = algorithmic_bytes(a.batch, opt.height, opt.width, opt.scales, motion)

    def frac_of(us):
        return round(conv_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    main_leg = legs.get("synthetic")
    if main_leg is None and kern_ms and tile_n.value > 0:        # eager mode / split-loss instrumentation: the step's own launches
        main_leg = {"tile_us": tile_us.value, "launches": tile_n.value, "photo_us": sum(kern_ms) / len(kern_ms) * 1e3,
                    "path_us": sum(path_ms) / len(path_ms) * 1e3}
    roof = None
    if main_leg is not None:
        avg_us = main_leg["tile_us"]
        gbs = conv_bytes / (avg_us * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": "dd::photo_tile_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "avg_launch_us": round(avg_us, 1), "launches_timed": main_leg["launches"],
                "dd_photo_loss_us": round(main_leg["photo_us"], 1),          # events around the tile kernel's part of the C-ABI call
                "algorithmic_bytes_per_launch": conv_bytes, "single_pass_bytes_per_launch": single_bytes,
                "achieved_single_pass": round(single_bytes / (avg_us * 1e-6) / 1e9, 1),
                # the WHOLE fused loss the north star states its target on (warp + SSIM + smoothness + motion regularisers + ground
                # term + assembly: every launch of dd_fused_loss, HIP events from the first to behind the last)
                "loss_path_us": round(main_leg["path_us"], 1), "frac_loss_path": frac_of(main_leg["path_us"]),
                "workload": "SURVEY 8(d) synthetic network outputs, low-frequency fields" if "synthetic" in legs else "the step's own network outputs",
                "timed_in": timed_in}
        for name, tag in (("identity", "identity_warps"), ("white_noise", "white_noise_fields")):
            leg = legs.get(name)
            if leg:
                roof["avg_launch_us_" + tag], roof["frac_" + tag] = round(leg["tile_us"], 1), frac_of(leg["tile_us"])
                roof["loss_path_us_" + tag], roof["frac_loss_path_" + tag] = round(leg["path_us"], 1), frac_of(leg["path_us"])
        if "loss_path_replayed_us" in replay_note:
            replay_note["frac_loss_path_replayed"] = frac_of(replay_note["loss_path_replayed_us"])
        roof.update(replay_note)
        roof.update(pmc_traffic(a, opt, motion))

    if rank == 0:
        imgs = a.batch * world * a.steps
        line = {
            "metric": "training images/sec ({}x{} triplets)".format(opt.height, opt.width), "value": round(imgs / elapsed, 2), "unit": "img/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": ("f32" if a.amp == "none" else a.amp + " networks / f32 loss (NOT the headline precision)") +
            ("" if _mfma_products() == 6 else " -- 3x3 convolutions from %d bf16 partial product(s) per multiply-add, matmul precision '%s' (NOT the headline precision)"
             % (_mfma_products(), torch.get_float32_matmul_precision())), "data": "synthetic",
            "config": {"workload": "{} {} {}x{} batch={}/GPU phase={} (all loss terms of the phase), random-init weights".format(
                a.dataset, a.depth_model, opt.height, opt.width, a.batch, a.phase),
                "global_batch": a.batch * world, "parallelism": "dp{}".format(world), "mode": mode, "mode_requested": a.mode, "auto_probe": auto_note, "miopen_find": bool(opt.miopen_find), "channels_last": bool(a.channels_last),
                "loss_path": "operators" if a.no_fused_loss else "fused HIP", "final_loss": round(loss_val, 6),
                "distinct_hw_queues_found": _queues_found(), "rccl_ranks": dist.get_world_size() if dist_on else 0, "dist_backend": backend if dist_on else None, "capture_fallback": capture_fallback,
                "reduce_mode": (getattr(seg_step, "reduce_mode_chosen", None) if seg_step is not None else ("flat buffer, one all-reduce behind backward()" if dist_on else None)),
                "reduce_probe_ms": getattr(seg_step, "reduce_probe_ms", None) if seg_step is not None else None,
                "allreduce_budget": allreduce_budget(seg_step, world),
                "library_gemms": "TunableOp " + __import__("gemm_env").STATE["status"],
                "photo_source_layout": ("pixel-interleaved copies of the two source frames (dd_pack_rgb, inside every timed step; {} loss evaluations recorded)".format(FL.PACKED_CALLS[0])
                                        if FL.PACKED_CALLS[0] > 0 else "planar (B,3,H,W) tensors"),
                # observed, not inferred from flags: launches of dd_conv_small_fwd in this process (eager warm-up steps + graph captures)
                "motion_decoder_full_res_convs": "dd_conv_small ({} forward launches recorded)".format(_small_conv_calls()) if _small_conv_calls() > 0 else "MIOpen (dd_conv_small never ran)",
                "conv3x3_stride1": ("dd_conv3x3_mfma: fp32 operands as three bf16 pieces, " + {6: "six", 3: "THREE (bf16x3)", 1: "ONE (bf16 operands)"}[_mfma_products()] + " MFMA partial products, fp32 accumulation ({} forward launches recorded, {} of them dd_conv3x3_mfma_flat on the small images)".format(_mfma_conv_calls(), __import__("hipops.functions", fromlist=["x"])._FLAT_CONV_CALLS[0])
                                    if _mfma_conv_calls() > 0 else "MIOpen fp32 (dd_conv3x3_mfma never ran)"),
                "conv3x3_stride1_half_precision": (("dd_conv3x3_half: half-precision operands, one MFMA per operand pair, fp32 accumulation; forward and data gradient "
                                                    "({} forward launches recorded), weight gradient on the library".format(_half_conv_calls()))
                                                   if _half_conv_calls() > 0 else ("the library (MIOpen / CK)" if a.amp != "none" else None)),
                "litemono_mlp": ("dd_pw_gemm: pwconv1 / pwconv2 and their data gradients on the bf16 matrix pipe (three bf16 pieces per fp32 operand), exact GELU in the second Linear's prologue ({} block passes recorded)".format(_mlp_calls())
                                 if _mlp_calls() > 0 else ("training passes: dd_mlp_fwd forward (hidden tile on chip, nothing but the input kept), the pre-activation rebuilt by one library GEMM in the backward ({} block forwards recorded)".format(_mlp_recompute_calls())
                                                            if _mlp_recompute_calls() > 0 else "training passes: hipBLASLt fp32 + ATen GELU") + "; statistics-only side batch: " +
                                 ("dd_mlp_fwd, the whole block in one kernel with the hidden tile on chip ({} block forwards recorded)".format(_mlp_fused_calls())
                                  if _mlp_fused_calls() > 0 else "the same (dd_mlp_fwd never ran)")),
                "optimizer_update": ("dd_adam_multi (one launch)" if getattr(seg_step, "one_launch_adam", None) is not None else "torch multi-tensor Adam ({})".format(getattr(seg_step, "adam_fallback", None))) if seg_step is not None else "torch multi-tensor Adam (eager step)",
                "side_frames": "depth encoder only (--stats_only_side_frames, NOT the headline)" if a.stats_only_side_frames else "full depth net, as the reference",
                "host_enqueue_ms_per_step": round(t_enqueued / a.steps * 1e3, 3), "host_enqueue_ms_per_rank": enqueue_per_rank},
            "roofline": roof,
            "roofline_conv3x3": _conv_mfma_roofline() if _mfma_conv_calls() > 0 else None,
        }
        if world == 1 and not a.no_cpu_baseline:
            note("timed region done; running the CPU baseline (bounded sample)")
            line["cpu_baseline"] = cpu_baseline_guarded([x for x in opt_args if x not in ("--no_hip_graph", "--nchw", "--single_stream", "--no_miopen_find", "--miopen_find")], a.phase, sample_batch=2)
            # the unmodified reference itself cannot travel to the GPU box; its timing in the build container is on record
            line["cpu_baseline"]["reference_in_build_container"] = reference_cpu_record()
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
