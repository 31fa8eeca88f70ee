"""Training runtime of Dynamo-Depth on MI355X -- same surface as the reference Trainer (Trainer.py:19-756):
four phases (disp_init, motion_init, mask_init, fine_tune), per-phase Adam + StepLR over a sub-set of the
networks, process_batch(inputs) -> (outputs, losses), per-module checkpoints.

What differs underneath:
  * the loss path of a step is ONE autograd node over fused HIP kernels (hipops.fused_loss) instead of
    ~2.5k-5.1k ATen launches with three host synchronisations; the operator-by-operator path
    (generate_images_pred + compute_losses over the tools.py HIP operators) is kept for eval scripts,
    log steps and --no_fused_loss;
  * one process per GPU over RCCL (`nccl` backend of PyTorch-ROCm), launched by torchrun or the reference's
    torch.distributed.launch line; gradients are averaged by DDP's bucketed all-reduce overlapped with backward;
  * optional whole-step hipGraph capture (--hip_graph) once the ramped loss weights are constant;
  * the target pyramid is built on the device, synthetic triplets can stand in for a dataset (--synthetic).
"""
import json
import os
import os.path as osp
import random
import time

import numpy as np
# before the HIP runtime initialises (first device call): hipGraph launches without the runtime's packet-capture path, which
# corrupts replayed steps when several launches are in flight (miopen_env.py has the story; train.py / bench.py set it there too)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

import datasets
import networks
from tools import BackprojectDepth, DepthMetrics, GroundPlane, Project3D, SSIM, compute_smooth_loss, depth_to_disp, disp_to_depth
from utils import interp, join_dir, make_ind_map, cart2polar, hsv_to_rgb, readlines, sec_to_hm_str

try:                                    # observability only; absent on the MI355X image
    import wandb
except Exception:                       # pragma: no cover
    wandb = None

PHASES = ("disp_init", "motion_init", "mask_init", "fine_tune")
# phase -> (bool_CmpFlow, bool_MotMask, optimised networks, lr factor)      reference Trainer.py:466-490
PHASE_TABLE = {
    "disp_init": (False, False, ["Depth", "Pose"], 1.0),
    "motion_init": (True, False, ["CmpFlow"], 1.0),
    "mask_init": (True, True, ["Pose", "CmpFlow", "MotMask"], 1.0),
    "fine_tune": (True, True, ["Depth", "Pose", "CmpFlow", "MotMask"], 0.5),
}


class _BicubicAA:
    """Tensor Resize((h,w), BICUBIC, antialias=True) of the reference (Trainer.py:80) without torchvision."""

    def __init__(self, size):
        self.size = tuple(size)

    def __call__(self, img):
        return F.interpolate(img, self.size, mode="bicubic", align_corners=False, antialias=True)


def _join_streams_then_allreduce(state, bucket):
    """DDP communication hook of the GPU path.  With the multi-stream forward autograd runs every backward node -- and
    the AccumulateGrad + reducer hook behind it, which copies the gradient into its bucket -- on the stream of the node's
    forward, so one bucket collects gradients from several streams, while the reducer orders the collective only behind the
    stream that is current when the bucket's LAST gradient arrives.  A gradient still in flight on another stream then
    either misses the collective or, worse, lands in the bucket after the reduced values (seen as ranks drifting apart in
    the depth stem's weights -- the last gradients of a backward -- tests/ddp_worker_gpu.py).  The hook runs at the moment
    the bucket is complete on the host: everything the other streams have enqueued so far includes their gradients of this
    bucket, so the current stream waits for all of them first; then the stock averaged all-reduce.  gloo (CPU collectives on
    GPU tensors: the two-ranks-on-one-device test) completes its future when the copy back to the GPU is merely enqueued on
    a private stream; there the collective runs blocking, which orders that copy on the current stream."""
    import torch.distributed as dist
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    model = state["model"]
    cur = torch.cuda.current_stream()
    for st in list(getattr(model, "_streams", None) or ()) + [getattr(model, "_main_stream", None)]:
        if st is not None and st != cur:
            cur.wait_stream(st)
    if dist.get_backend(state["group"]) == "gloo":
        buf = bucket.buffer()
        buf.div_(dist.get_world_size(state["group"]))
        dist.all_reduce(buf, group=state["group"])
        fut = torch.futures.Future()
        fut.set_result(buf)
        return fut
    return default_hooks.allreduce_hook(state["group"], bucket)


class FlatGradients:
    """The gradients of a phase's trainable parameters as views into ONE contiguous buffer, averaged over the ranks by ONE
    all-reduce behind backward() -- the eager steps' counterpart of segments.SegmentedStep's flat buffer.
    torch's DDP reducer copies every gradient into a bucket on the stream of its backward node and joins its bucket stream with
    the caller's once per bucket: with the branches of the backward on four streams that serialised them (the eager step lost
    16 % as soon as a process group existed: 221.5 against 262.8 img/s with ONE rank).  Here autograd accumulates straight into the
    views (each with its parameter's own strides: channels-last weights stay channels-last), the caller joins the branch streams
    once and issues one collective of the whole buffer; what the reference's DDP (Trainer.py:44, train.py:6-10) computes -- the
    mean of the ranks' gradients -- is what comes out.

    Optimizer state: every trainable parameter keeps a (zero-filled) `.grad` attached, also one that received no gradient in a
    step; torch's DDP wrapper and the single-GPU path leave such a grad None and Adam skips the parameter.  With a zero gradient Adam
    applies a zero update but still advances that parameter's step count and decays its moments -- so the flat mode is not
    state-identical to the `ddp` mode for parameters that go without a gradient in SOME steps of a phase.  In the four phases of
    the schedule every parameter of an optimised network is reached by the loss in every step (the phase's parameter list is
    exactly the reached networks: Trainer.setup_phase), so the two coincide; the replayed step has the same property by
    construction (its Adam graph updates the whole flat buffer)."""

    def __init__(self, params, device):
        self.params = [p for p in params if p.requires_grad]
        total = sum((p.numel() + 3) & ~3 for p in self.params)
        self.flat = torch.zeros(max(total, 1), dtype=torch.float32, device=device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].as_strided(p.size(), p.stride()))
            off += (p.numel() + 3) & ~3

    def attach(self):
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def zero(self):
        self.flat.zero_()
        self.attach()

    def all_reduce(self, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)        # RCCL: the average in the collective itself
        else:
            self.flat.div_(world)
            dist.all_reduce(self.flat, group=group)


class EpochSubsetSampler(torch.utils.data.Sampler):
    """The sample order of one epoch over a PERSISTENT dataset of the whole split: `set_subset(indices)` names the epoch's files
    (drawn by the caller exactly as the reference draws its file subset), __iter__ shuffles them the way a fresh loader over a
    dataset of just those files would -- single process: RandomSampler (a seed drawn from torch's global generator per pass);
    under --ddp: DistributedSampler(shuffle=True, seed 0, epoch 0 -- the reference never calls set_epoch), padded to a multiple
    of the world size and strided by rank."""

    def __init__(self, n_total, world=1, rank=0):
        self.n_total, self.world, self.rank = int(n_total), int(world), int(rank)
        self.subset = None

    def set_subset(self, indices):
        self.subset = None if indices is None else [int(i) for i in indices]

    def _count(self):
        return self.n_total if self.subset is None else len(self.subset)

    def __len__(self):
        n = self._count()
        return n if self.world == 1 else (n + self.world - 1) // self.world

    def __iter__(self):
        n = self._count()
        if self.world == 1:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            order = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
        else:
            order = torch.randperm(n, generator=torch.Generator().manual_seed(0)).tolist()
            total = len(self) * self.world
            pad = total - len(order)
            if pad > 0:
                order += (order * ((pad + len(order) - 1) // max(len(order), 1)))[:pad]
            order = order[self.rank:total:self.world]
        for i in order:
            yield i if self.subset is None else self.subset[i]


def mirror_fresh_loader_draw(loader):
    """A fresh DataLoader iterator draws its `_base_seed` from torch's global generator BEFORE the sampler draws its shuffle seed
    (torch/utils/data/dataloader.py, _BaseDataLoaderIter.__init__); the reference builds a fresh loader every epoch
    (Trainer.py:526-531).  A loader with persistent workers re-uses its iterator through `_reset`, which draws nothing -- from the
    second epoch on the sampler would receive the value the base seed should have consumed and the sample ORDER would leave the
    reference's.  Draw and discard that value when the iterator is about to be re-used.  (The workers' own augmentation
    streams are seeded once, at start-up: those do differ from a loader whose workers are re-seeded every epoch.)"""
    if getattr(loader, "persistent_workers", False) and loader.num_workers > 0 and getattr(loader, "_iterator", None) is not None:
        torch.empty((), dtype=torch.int64).random_(generator=loader.generator)


class Trainer:
    def __init__(self, options):
        self.opt = opt = options
        assert opt.height % 32 == 0, "height(={}) must be a multiple of 32".format(opt.height)
        assert opt.width % 32 == 0, "width(={}) must be a multiple of 32".format(opt.width)
        assert opt.frame_ids[0] == 0, "frame_ids(={}) must start with 0".format(opt.frame_ids)
        assert len(opt.epoch_schedules) == 4 and all(e >= 0 for e in opt.epoch_schedules), \
            "epoch_schedules(={}) must be length=4 and non-negative".format(opt.epoch_schedules)
        for name, default in (("fused_loss", True), ("hip_graph", None), ("synthetic", False), ("amp", "none"), ("multi_stream", None),
                              ("channels_last", None), ("miopen_find", None), ("skip_unused_depth_frames", False), ("local_world_size", 1), ("resume", "")):
            if not hasattr(opt, name):
                setattr(opt, name, default)
        # the fast configuration is the default on a GPU (flags left at None): channels-last networks, multi-stream forward,
        # per-network hipGraphs, MIOpen Find -- `python train.py -d kitti` runs what bench.py measures
        on_gpu = torch.cuda.is_available()
        for name in ("hip_graph", "multi_stream", "channels_last"):
            if getattr(opt, name) is None:
                setattr(opt, name, on_gpu)
        if opt.miopen_find is None:
            opt.miopen_find = not self._find_db_covers(opt)
        if on_gpu and opt.miopen_find and os.environ.get("DD_MIOPEN_FIND", "1") != "0":
            # (DD_MIOPEN_FIND=0: the test-suite's switch -- Find on dozens of one-off shapes takes minutes per test)
            torch.backends.cudnn.benchmark = True
        if getattr(opt, "matmul_precision", None):
            # PyTorch's own switch for fp32 contractions; hipops.functions.mfma_products() reads it at every dd_conv3x3_mfma call
            torch.set_float32_matmul_precision(opt.matmul_precision)
        if on_gpu:
            import gemm_env
            gemm_env.enable()           # recorded solution choices for the library GEMMs (LiteMono's Linears): gemm_db/, TunableOp with tuning off

        self.local_rank = opt.local_rank
        self.cuda_id = opt.cuda_ids[self.local_rank]
        if torch.cuda.is_available():
            assert self.cuda_id < torch.cuda.device_count(), "cuda_ids[local_rank](={}) must be visible".format(self.cuda_id)
            self.device = torch.device("cuda:{}".format(self.cuda_id))
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device("cpu")
        self.print("\n=============== Trainer Initialization ===============")

        self.base_model = networks.Model(opt)
        self._resume = None
        if opt.resume != "":
            # continue a run: the folder's per-module weights are the checkpoint to load; optimizer / scheduler / counters /
            # random-number streams follow in train() once the phase's optimizer exists
            opt.load_ckpt = opt.resume
            with open(osp.join(osp.expanduser(opt.resume), "resume.json")) as fh:
                self._resume = json.load(fh)
        if opt.load_ckpt != "":
            self.load_model()
        self.base_model.to(self.device)
        if opt.channels_last:
            self.base_model.to(memory_format=torch.channels_last)
        # DDP wraps per phase (setup_phase -> wrap_for_phase): the set of parameters that receive gradients differs from phase
        # to phase, and a wrapper built for exactly that set needs neither find_unused_parameters nor its per-step graph walk
        self.model = self.base_model

        self.num_scales = len(opt.scales)
        self.B, self.H, self.W = opt.batch_size, opt.height, opt.width
        self.log_path = osp.join(opt.log_dir, opt.model_name)
        table = {"kitti": datasets.KITTIDataset, "waymo": datasets.WaymoDataset, "nuscenes": datasets.nuScenesDataset}
        self.dataset = datasets.SyntheticTriplets if opt.synthetic else table[opt.dataset]

        self.depth_metrics = DepthMetrics(opt.eval_img_bound, opt.eval_min_depth, opt.eval_max_depth)
        self.gplane = GroundPlane(num_points_per_it=opt.gp_np_per_it, max_it=opt.gp_max_it, tol=opt.gp_tol, g_prior=opt.gp_prior)
        self.ssim = SSIM().to(self.device)
        self.bce = nn.BCEWithLogitsLoss()
        self.I = torch.eye(4).reshape(1, 4, 4).repeat(self.B, 1, 1).to(self.device)
        self.resize, self.backproject_depth, self.project_3d, self.prob_target = {}, {}, {}, {}
        for s in opt.scales:
            h, w = self.H // (2 ** s), self.W // (2 ** s)
            self.resize[s] = _BicubicAA((h, w))
            self.backproject_depth[s] = BackprojectDepth(self.B, h, w).to(self.device)
            self.project_3d[s] = Project3D(self.B, h, w).to(self.device)
            self.prob_target[s] = torch.zeros(self.B, 1, h, w).to(self.device)
        self.bool_automask = True
        self.step, self.epoch, self.g_step = 0, 0, 0
        self.num_steps_per_epoch = max(int(getattr(opt, "epoch_size", 1)), 1)
        self.noise_override = None          # tests: {scale: (B,2,H,W)} replacing the tie-break randn of Trainer.py:339
        self.rand_idx_override = None       # tests: {scale: (B, max_it*np)} replacing the RANSAC draws of tools.py:125-127
        self.materialise = False            # set on log steps: the fused path then also writes the image-sized outputs
        self._graph = None
        self.save_opt()
        self.print("=============== Trainer Initialization ===============\n")

    @staticmethod
    def _find_db_covers(opt):
        """Are this run's convolution problems among the shipped find-db records (miopen_db/recorded.json)?  Then MIOpen Find has
        nothing to add: immediate mode returns the recorded solver of every problem -- measured the same step time (263.6 against
        262.4 img/s on the headline workload) with the first step after 1.5 s instead of 50 s (torch's benchmark mode walks MIOpen's
        Find for every problem even when the record exists)."""
        path = osp.join(osp.dirname(osp.abspath(__file__)), "miopen_db", "recorded.json")
        try:
            with open(path) as fh:
                rows = json.load(fh)["recorded"]
        except (OSError, ValueError, KeyError):
            return False
        if "MIOPEN_USER_DB_PATH" not in os.environ and "MIOPEN_SYSTEM_DB_PATH" not in os.environ:
            return False                 # miopen_env.setup() was not called: the records are not in use
        if not Trainer._find_db_matches_device(osp.dirname(path)):
            return False                 # records of another GPU / MIOpen build: immediate mode would find nothing, keep Find on
        key = {"depth_model": opt.depth_model, "height": opt.height, "width": opt.width, "batch_size": opt.batch_size, "amp": getattr(opt, "amp", "none")}
        return bool(getattr(opt, "channels_last", True) is not False) and any(all(r.get(k) == v for k, v in key.items()) for r in rows)

    @staticmethod
    def _find_db_matches_device(db_dir, arch=None, cus=None, miopen_version=None):
        """Do the shipped find-db files (`<arch><CU count in hex>.HIP.<major>_<minor>_<patch>_*.ufdb.txt`, MIOpen's own naming) belong
        to THIS device and MIOpen build?  MIOpen looks records up under exactly that name: on another GPU or version immediate mode
        finds none and falls back to heuristic solvers -- silently slower with Find off.  arch / cus / miopen_version:
        overrides for the CPU test; by default read from the device and the runtime."""
        import re
        try:
            names = [n for n in os.listdir(db_dir) if n.endswith(".ufdb.txt")]
        except OSError:
            return False
        if arch is None:
            if not torch.cuda.is_available():
                return False
            prop = torch.cuda.get_device_properties(torch.cuda.current_device())
            arch, cus = str(getattr(prop, "gcnArchName", "")).split(":")[0], int(prop.multi_processor_count)
            v = torch.backends.cudnn.version()           # MIOpen's version on ROCm: major * 1e6 + minor * 1e3 + patch
            miopen_version = None if not v else (v // 1000000, (v // 1000) % 1000, v % 1000)
        for n in names:
            m = re.match(r"^(gfx[0-9a-f]+?)([0-9a-f]{2,3})\.HIP\.(\d+)_(\d+)_(\d+)_", n)
            if not m:
                continue
            # gfx950 + "100" (256 CUs): the arch name is the prefix of the device's, the rest is the CU count in hex
            if not (n.startswith(arch) and n[len(arch):].split(".")[0] == "{:x}".format(cus)):
                continue
            if miopen_version is not None and tuple(int(x) for x in m.groups()[2:5]) != tuple(miopen_version):
                continue
            return True
        return False

    # ===================================================================================================
    # schedule
    # ===================================================================================================
    def train(self):
        self.setup_wandb()
        self.g_step = 0
        self.init_loaders()
        resume = self._resume
        for i, phase in enumerate(PHASES):
            epochs = self.opt.epoch_schedules[i]
            if resume is not None and PHASES.index(resume["phase"]) > i:
                self.print("======== {} - finished before the resumed checkpoint ========".format(phase.upper()))
                continue
            self.print("======== {} - Num Epochs={} ========".format(phase.upper(), epochs))
            if epochs > 0:
                self.run_phase(phase, epochs, resume=resume if (resume is not None and resume["phase"] == phase) else None)
            self.print("======== {} - Num Epochs={} ========\n".format(phase.upper(), epochs))

    def run_phase(self, phase_name, num_epoch, resume=None):
        self.setup_phase(phase_name)
        self.step, self.epoch = 0, 0
        first_epoch = 0
        if resume is not None:
            first_epoch = self.restore_training_state(resume)
        self.bool_automask = phase_name == "disp_init"          # Trainer.py:117
        self.num_total_steps = self.num_steps_per_epoch * num_epoch
        self.start_time = time.time()
        for self.epoch in range(first_epoch, num_epoch):
            self.print()
            self.run_epoch()
            if (self.epoch + 1) % self.opt.save_frequency == 0 or self.epoch == num_epoch - 1:
                self.save_model(phase_name)

    def run_epoch(self):
        self.setup_train_loader()
        self.set_train()
        gpu_time = data_time = 0.0
        tic = time.time()
        self.zero_grads()
        loader = self.train_loader
        if self.device.type == "cuda" and getattr(self.opt, "prefetch", True):
            from hipops.inputs import DevicePrefetcher
            loader = DevicePrefetcher(self.train_loader, self.process_inputs, self.device)      # batch k+1 uploads / prepares under step k
        window_start, window_steps = time.time(), 0           # steps since the last log line (validation excluded)
        for batch_idx, inputs in enumerate(loader):
            data_time += time.time() - tic
            tic = time.time()
            early = batch_idx % self.opt.log_frequency == 0 and self.step < 10 * self.opt.log_frequency
            late = self.step % (10 * self.opt.log_frequency) == 0
            self.materialise = (early or late) and not self.opt.no_train_vis
            outputs, losses = self.train_step(inputs)
            took = time.time() - tic
            gpu_time += took
            window_steps += 1
            if early or late:
                loss = losses["loss"].detach().cpu()           # waits for the step: the window's wall time is complete
                # examples/s over the steps since the last log line: a step returns once it is ENQUEUED (a replayed step in a
                # third of its run time), so the duration of one call (what the reference prints, Trainer.py:153-160) says nothing
                self.log_time(batch_idx, (time.time() - window_start) / window_steps, loss, data_time, gpu_time)
                if not self._all_ranks_finite(loss) and not getattr(self.opt, "keep_going_on_nan", False):
                    terms = {k: float(v) for k, v in losses.items() if torch.is_tensor(v) and v.numel() == 1}
                    bad = [n for n, p in self.base_model.named_parameters() if not bool(torch.isfinite(p).all())]
                    probe = self._graph.probe_report() if hasattr(self._graph, "probe_report") else ""
                    raise FloatingPointError("non-finite loss at step {} (batch {} of epoch {}): {}; {} parameters non-finite{}\n{}".format(
                        self.step, batch_idx, self.epoch, terms, len(bad), ", first: " + bad[0] if bad else "", probe))
                gpu_time = data_time = 0.0
                self.log("train", inputs, outputs, losses)
                self.val(batch_idx)
                window_start, window_steps = time.time(), 0
            del outputs
            self.g_step += 1
            self.step += 1
            tic = time.time()
        self.step_lr_scheduler()

    def _all_ranks_finite(self, loss):
        """True when the logged loss is finite on EVERY rank (one MAX all-reduce of a flag on log steps): a rank that stopped on
        its own local loss would leave the others waiting in val()'s buffer broadcast until the collective times out."""
        bad = 0.0 if bool(torch.isfinite(loss)) else 1.0
        if self.opt.ddp and torch.distributed.is_available() and torch.distributed.is_initialized():
            flag = torch.tensor([bad], device=self.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            bad = float(flag)
        return bad == 0.0

    def step_lr_scheduler(self):
        """StepLR at the end of an epoch.  A captured step has the learning rate baked into its fused-Adam launch: drop the
        graph when the rate changes so that the next step re-captures with the new one."""
        before = [g["lr"] for g in self.optim["optimizer"].param_groups]
        self.optim["lr_scheduler"].step()
        if [g["lr"] for g in self.optim["optimizer"].param_groups] != before and isinstance(self._graph, dict):
            self._graph = None               # (the per-network graphs re-capture their optimizer graph alone: segments.SegmentedStep.run)

    def train_step(self, inputs):
        """process_batch + backward + optimizer step (the timed `compute` region of Trainer.py:145-153)."""
        # (fp16 networks: the per-network graphs carry the dynamic loss scaler on the device -- segments.py; the whole-step graph does not)
        if (self.opt.hip_graph and self.device.type == "cuda" and not self.materialise and self._weights_constant()
                and (self._graph_mode() == "segments" or (not self.opt.ddp and self.opt.amp != "fp16"))):
            return self._graph_step(inputs)
        if self._graph is not None:
            # The captured graph ends after optimizer.step(): p.grad still references the last replay's gradients (graph-pool
            # memory).  An eager step must start from empty gradients or AccumulateGrad adds onto them; replays keep writing
            # to their own fixed pool addresses, so dropping the references is safe.
            self.zero_grads(set_to_none=True)
        elif getattr(self, "_flat_grads", None) is not None:
            self._flat_grads.attach()            # (a caller may have dropped the views: zero_grad(set_to_none=True) is torch's default)
        outputs, losses = self.process_batch(inputs)
        scaler = self._grad_scaler()
        if scaler is None:
            losses["loss"].backward()
            self.reduce_eager_grads()
            self.optim["optimizer"].step()
        else:
            # fp16 networks (--amp fp16, config 5 of BASELINE.json): dynamic loss scaling keeps the small gradients of the
            # half-precision convolutions representable; the fp32 loss path and the fp32 master weights are untouched
            scaler.scale(losses["loss"]).backward()
            self.reduce_eager_grads()            # (the scaled gradients: averaging commutes with the unscale)
            scaler.step(self.optim["optimizer"])
            scaler.update()
        self.zero_grads()
        return outputs, losses

    def _grad_scaler(self):
        if self.opt.amp != "fp16" or self.device.type != "cuda":
            return None
        if getattr(self, "_scaler", None) is None:
            self._scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        return self._scaler

    def val(self, batch_idx):
        self.sync_buffers()          # evaluate with rank 0's running statistics on every rank (what the reference's DDP does)
        self.set_eval()
        try:
            inputs = next(self.val_iter)
        except StopIteration:
            self.val_iter = iter(self.val_loader)
            inputs = next(self.val_iter)
        with torch.no_grad():
            outputs, losses = self.process_batch(inputs)
            if self.val_loader.dataset.load_depth:
                if ("disp_scaled", 0, 0) not in outputs:
                    outputs[("disp_scaled", 0, 0)] = disp_to_depth(outputs[("disp", 0, 0)], self.opt.min_depth, self.opt.max_depth)[0]
                losses.update(self.depth_metrics(inputs, outputs))
            self.log("val", inputs, outputs, losses)
        self.set_train()

    # ===================================================================================================
    # the step
    # ===================================================================================================
    def process_batch(self, inputs):
        self.process_inputs(inputs)
        return self.forward_and_losses(inputs)

    def forward_and_losses(self, inputs):
        outputs = self.run_networks(inputs)
        if self.opt.fused_loss and self.device.type == "cuda":
            losses = self.fused_losses(inputs, outputs)
        else:
            self.generate_images_pred(inputs, outputs)
            losses = self.compute_losses(inputs, outputs)
        return outputs, losses

    def run_networks(self, inputs):
        if self.opt.amp != "none" and self.device.type == "cuda":
            dtype = torch.bfloat16 if self.opt.amp == "bf16" else torch.float16
            # autocast caches the half-precision copy of every weight for the length of the context -- made on the stream that
            # uses the weight first, handed to every later user without a dependency.  With the branches of the forward on
            # separate streams (the depth net runs on two of them at once) a branch could read a copy that is still being
            # written: the run-dependent `final_loss=nan` of the fp16 bench rows (one of three in round 2, one of one in round 3;
            # never with per-module host syncs, scripts/probe_amp_overflow.py).  No cache then: every use casts for itself.
            cache = not (getattr(self.opt, "multi_stream", False) and not torch.cuda.is_current_stream_capturing())
            if os.environ.get("DD_AMP_CACHE") in ("0", "1"):         # (A/B switch of scripts/probe_amp_nan.py)
                cache = os.environ["DD_AMP_CACHE"] == "1"
            with torch.autocast("cuda", dtype=dtype, cache_enabled=cache):
                outputs = self.model(inputs)
            # the loss path is fp32 (the reference has no AMP): promote what it reads -- once per tensor: the two frames share
            # their flow-field / mask tensors and the fused loss recognises that by identity
            once = {}
            for k, v in list(outputs.items()):
                if torch.is_tensor(v) and v.dtype != torch.float32 and k[0] in ("disp", "cam_T_cam", "complete_flow", "complete_flow_field", "motion_prob",
                                                                                "motion_mask", "axisangle", "translation"):
                    if id(v) not in once:
                        once[id(v)] = v.float()
                    outputs[k] = once[id(v)]
            return outputs
        return self.model(inputs)

    def loss_coefficients(self):
        """losses['loss_coef/*']: g_* with the linear ramp over the first 1/ramp_red of the phase's first epoch (Trainer.py:299-310)."""
        names = [k[2:] for k in self.opt.__dict__.keys() if k[:2] == "g_"]
        coefs = {}
        for n in names:
            v = getattr(self.opt, "g_" + n)
            if "g_" + n in self.opt.weight_ramp:
                v = v * float(np.clip(self.opt.ramp_red * self.step / self.num_steps_per_epoch, 0.0, 1.0))
            coefs[n] = v
        return names, coefs

    def _weights_constant(self):
        return self.opt.ramp_red * self.step / self.num_steps_per_epoch >= 1.0 or not any(
            n in ("CmpFlow", "MotMask") for n in self.optim["network_names"])

    def fused_losses(self, inputs, outputs):
        from hipops.fused_loss import LossPlan, fused_loss
        _, coefs = self.loss_coefficients()
        o = self.opt
        plan = LossPlan(height=self.H, width=self.W, scales=o.scales, min_depth=o.min_depth, max_depth=o.max_depth,
                        ssim_weight=o.ssim_weight, mask_disp_thrd=o.mask_disp_thrd, gp_prior=o.gp_prior, gp_tol=o.gp_tol,
                        gp_max_it=o.gp_max_it, gp_np_per_it=o.gp_np_per_it, cmpflow=self.base_model.bool_CmpFlow,
                        motmask=self.base_model.bool_MotMask, automask=self.bool_automask,
                        optimised=self.optim["network_names"], coefs=coefs)
        return fused_loss(plan, inputs, outputs, frame_ids=o.frame_ids, noise=self.noise_override, rand_idx=self.rand_idx_override,
                          materialise=self.materialise)

    # ---- operator-by-operator path (same structure of results as the reference's two methods) --------------
    def generate_images_pred(self, inputs, outputs):
        """Warps every source frame into the target view at every scale (reference Trainer.py:215-287) with the
        tools.py HIP operators; fills the same `outputs` keys."""
        o, H, W = self.opt, self.H, self.W
        cmpflow, motmask = self.base_model.bool_CmpFlow, self.base_model.bool_MotMask
        K, inv_K = inputs[("K", 0)], inputs[("inv_K", 0)]
        lift, drop = self.backproject_depth[0], self.project_3d[0]
        for s in o.scales:
            disp_s = outputs[("disp", 0, s)]
            B, _, h, w = disp_s.shape
            scaled, depth = disp_to_depth(interp(disp_s, (H, W)), o.min_depth, o.max_depth)
            outputs[("depth", 0, s)], outputs[("disp_scaled", 0, s)] = depth, scaled
            for f in o.frame_ids[1:]:
                T = outputs[("cam_T_cam", 0, f)]
                pts = lift(depth, inv_K)
                outputs[("cam_points", 0, s)] = pts
                if motmask:
                    mask_full = interp(outputs[("motion_mask", f, s)], (H, W))
                else:
                    outputs[("motion_mask", f, s)] = torch.ones(B, 1, h, w, device=self.device)
                    mask_full = torch.ones(B, 1, H, W, device=self.device)
                outputs[("motion_mask_r", f, s)] = mask_full
                if cmpflow:
                    grid_ego, ego = drop(pts, K, T)
                    complete = interp(outputs[("complete_flow", f, s)], (H, W)).view(B, 3, -1) * inputs[("ts", f)].view(B, 1, 1)
                    residual = complete - ego
                    independ = residual * mask_full.view(B, 1, -1)
                    outputs[("sample_ego", f, s)] = grid_ego.detach()
                    shifted = torch.cat([pts.detach()[:, :3] + complete, pts.detach()[:, 3:]], 1)
                    outputs[("sample_complete", f, s)] = drop(shifted, K, None)[0].detach()
                    if motmask:
                        moved = torch.cat([pts[:, :3] + independ, pts[:, 3:]], 1)
                        grid, _ = drop(moved, K, T)
                    else:
                        moved = torch.cat([pts[:, :3] + complete, pts[:, 3:]], 1)
                        grid, _ = drop(moved, K, None)
                else:
                    grid, ego = drop(pts, K, T)
                    residual = torch.zeros_like(ego)
                    independ = torch.zeros_like(ego)
                outputs[("sample", f, s)] = grid
                outputs[("color", f, s)] = F.grid_sample(inputs[("color", f, 0)], grid, padding_mode="border", align_corners=True)
                outputs[("ego_flow", f, s)] = ego
                outputs[("independ_flow", f, s)] = independ.reshape(B, 3, H, W)
                outputs[("residual_flow", f, s)] = interp(residual.reshape(B, 3, H, W), (h, w))
                if self.bool_automask:
                    outputs[("color_identity", f, s)] = inputs[("color", f, 0)]

    def compute_reprojection_loss(self, pred, target):
        """ssim_weight * mean_c SSIM + (1 - ssim_weight) * mean_c |target - pred| (reference Trainer.py:413-423)."""
        a = self.opt.ssim_weight
        return a * self.ssim(pred, target).mean(1, True) + (1 - a) * torch.abs(target - pred).mean(1, True)

    def compute_losses(self, inputs, outputs):
        """All loss terms from the materialised outputs (reference Trainer.py:289-411)."""
        o = self.opt
        trained = self.optim["network_names"]
        names, coefs = self.loss_coefficients()
        losses = {"loss": 0}
        for t in names + list(o.scales):
            losses["loss_term/{}".format(t)] = 0
        for t in names:
            losses["loss_coef/{}".format(t)] = coefs[t]
        src = o.frame_ids[1:]
        target = inputs[("color", 0, 0)]
        cmpflow, motmask = self.base_model.bool_CmpFlow, self.base_model.bool_MotMask
        for s in o.scales:
            term = {t: 0 for t in names}
            color = inputs[("color", 0, s)]
            warped = torch.cat([self.compute_reprojection_loss(outputs[("color", f, s)], target) for f in src], 1)
            if self.bool_automask:
                ident = torch.cat([self.compute_reprojection_loss(inputs[("color", f, 0)], target) for f in src], 1)
                noise = self.noise_override[s].to(self.device) if self.noise_override is not None else torch.randn(ident.shape, device=self.device)
                ident = ident + noise * 0.00001      # tie breaker
                stack = torch.cat((ident, warped), 1)
            else:
                stack = warped
            if stack.shape[1] == 1:
                best = stack
            else:
                best, which = torch.min(stack, dim=1)
            if self.bool_automask:
                outputs["identity_selection/{}".format(s)] = (which > ident.shape[1] - 1).float()
            term["p_photo"] = best.mean()
            disp = outputs[("disp", 0, s)]
            if "Depth" in trained:
                if coefs["d_smooth"] > 0:
                    term["d_smooth"] = compute_smooth_loss(disp / (disp.mean(2, True).mean(3, True) + 1e-7), color) / (2 ** s)
                if coefs["d_ground"] > 0 and motmask:
                    _, diff, _ = self.process_ground(inputs, outputs, scale=s)
                    term["d_ground"] = -1 * torch.where(diff > 0, torch.zeros_like(diff), diff).mean() / (2 ** s)
            for f in src:
                mask = outputs[("motion_mask", f, s)]
                h, w = mask.shape[-2:]
                if "CmpFlow" in trained and cmpflow:
                    if coefs["c_smooth"] > 0:
                        term["c_smooth"] = term["c_smooth"] + compute_smooth_loss(outputs[("complete_flow", f, s)], color) / (2 ** s) / len(src)
                    if motmask and coefs["c_consistency"] > 0:
                        static_w = (disp > o.mask_disp_thrd).detach() * (1 - mask.detach())
                        term["c_consistency"] = term["c_consistency"] + torch.mean(static_w * torch.abs(outputs[("residual_flow", f, s)])) / (2 ** s) / len(src)
                if "MotMask" in trained and motmask:
                    if coefs["m_sparsity"] > 0:
                        e = interp(outputs[("sample_ego", f, s)].permute(0, 3, 1, 2), (h, w))
                        k = interp(outputs[("sample_complete", f, s)].permute(0, 3, 1, 2), (h, w))
                        mag = torch.sum((e - k) ** 2, 1)
                        static = (mag < mag.mean()).unsqueeze(1)
                        if torch.all(torch.sum(static, (1, 2, 3)) > 0):
                            prob = outputs[("motion_prob", f, s)]
                            term["m_sparsity"] = term["m_sparsity"] + self.bce(prob[static], self.prob_target[s][:prob.shape[0]][static]) / (2 ** s) / len(src)
                    if coefs["m_smooth"] > 0:
                        term["m_smooth"] = term["m_smooth"] + compute_smooth_loss(mask, color) / (2 ** s) / len(src)
            for t in names:
                losses["loss_term/{}".format(s)] = losses["loss_term/{}".format(s)] + term[t] * coefs[t]
                losses["loss_term/{}".format(t)] = losses["loss_term/{}".format(t)] + term[t]
            losses["loss"] = losses["loss"] + losses["loss_term/{}".format(s)] / self.num_scales
        return losses

    def process_ground(self, inputs, outputs, scale=0):
        """Ground plane from the scale's own back-projection; disparity excess over the plane (reference Trainer.py:425-444)."""
        o = self.opt
        disp = outputs[("disp", 0, scale)]
        _, depth = disp_to_depth(disp, o.min_depth, o.max_depth)
        inv_K = inputs[("inv_K", scale)]
        h, w = self.H // (2 ** scale), self.W // (2 ** scale)
        pts = self.backproject_depth[scale](depth, inv_K)
        ridx = None if self.rand_idx_override is None else self.rand_idx_override[scale]
        dist, plane = self.gplane(pts[:, :3].reshape(-1, 3, h, w), rand_idx=ridx)
        g_mask = (torch.abs(dist) < o.gp_tol).float()
        shifted = plane.clone()
        shifted[:, 2] += o.gp_tol
        ground_disp, ground_depth = self.get_ground_depth(shifted, inv_K, scale)
        diff = disp - ground_disp
        diff = torch.where(ground_depth == o.max_depth, torch.zeros_like(diff), diff)
        return dist, diff, g_mask

    def get_ground_depth(self, plane_param, inv_K, scale=0):
        """Depth at which every pixel ray meets the plane, clipped to max_depth (reference Trainer.py:446-461)."""
        h, w = self.H // (2 ** scale), self.W // (2 ** scale)
        B = inv_K.size(0)
        rays = torch.matmul(inv_K[:, :3, :3], self.backproject_depth[scale].pix_coords[:B])
        w1, w2, w3 = plane_param[:, 0:1], plane_param[:, 1:2], plane_param[:, 2:3]
        depth = (w3 / (rays[:, 1:2] - rays[:, 0:1] * w1 - rays[:, 2:3] * w2)).reshape(B, 1, h, w)
        depth = torch.where((depth < 0) | (depth > self.opt.max_depth), torch.full_like(depth, self.opt.max_depth), depth)
        return depth_to_disp(depth, self.opt.min_depth, self.opt.max_depth), depth

    # ===================================================================================================
    # phases / optimiser
    # ===================================================================================================
    def setup_phase(self, phase_name):
        if phase_name not in PHASE_TABLE:
            raise Exception("Phase name {} not recognized.".format(phase_name))
        cmpflow, motmask, nets, lr_factor = PHASE_TABLE[phase_name]
        self.base_model.bool_CmpFlow, self.base_model.bool_MotMask = cmpflow, motmask
        self.drop_graphs()
        self.optim = self.get_optim(list(nets), lr_factor=lr_factor)
        self.phase_name = phase_name
        self.wrap_for_phase(list(nets))

    def wrap_for_phase(self, network_names):
        """Marks exactly the phase's optimised parameters as trainable and (under --ddp) wraps the model for them.

        The reference wraps once with find_unused_parameters=True (Trainer.py:44) because out-of-phase networks and the
        torchvision `fc` heads never receive gradients; DDP then walks the autograd graph after every forward to find them.
        Here the parameters outside the phase's optimizer are frozen instead (requires_grad=False: backward skips their
        weight gradients -- they were computed and thrown away -- and DDP registers no hook for them), the parameters that
        can never be reached (`encoder.fc`) likewise, and the wrapper is built per phase with static_graph=True.
        broadcast_buffers=False: the per-step broadcast of rank 0's BatchNorm running statistics does not touch training-mode
        arithmetic and rank 0 writes the checkpoints from its own statistics either way -- one collective per step less.
        Gradients live inside the all-reduce buckets (gradient_as_bucket_view); 48 MB buckets: three to five collectives per
        step, each long enough to run at xGMI ring bandwidth while backward continues.
        With --multi_stream the gradients of a bucket come from several streams: see _join_streams_then_allreduce."""
        trainable = set(id(p) for p in self.base_model.parameters_by_names(network_names))
        for name, p in self.base_model.named_parameters():
            p.requires_grad_(id(p) in trainable and ".fc." not in name)
        self._flat_grads = None
        if self.opt.ddp and self._eager_reduce_mode() == "flat":
            # no wrapper: one flat gradient buffer, one collective behind backward() (FlatGradients); what DDP's constructor does
            # once -- every rank starts from rank 0's parameters and buffers -- is done here
            import torch.distributed as dist
            self.model = self.base_model
            with torch.no_grad():
                for t in list(self.base_model.parameters()) + list(self.base_model.buffers()):
                    dist.broadcast(t, src=0)
            self._flat_grads = FlatGradients(self.base_model.parameters(), self.device)
            self._flat_grads.attach()
        elif self.opt.ddp:
            ids = [self.cuda_id] if self.device.type == "cuda" else None
            # DD_DDP_BUCKET_MB (default 48): the size of the all-reduce buckets of the eager steps
            bucket = int(os.environ.get("DD_DDP_BUCKET_MB", "48"))
            self.model = DDP(self.base_model, device_ids=ids, static_graph=True, gradient_as_bucket_view=True, bucket_cap_mb=bucket,
                             broadcast_buffers=False)
            if getattr(self.opt, "multi_stream", False) and self.device.type == "cuda":
                self.model.register_comm_hook({"model": self.base_model, "group": None}, _join_streams_then_allreduce)
        else:
            self.model = self.base_model

    def _eager_reduce_mode(self):
        """How eager steps average gradients under --ddp: 'flat' (FlatGradients; the default on a GPU) or 'ddp' (torch's reducer with the
        stream-joining communication hook; the default on the CPU, where tests/test_ddp_gloo.py pins the wrapper's contract).
        DD_EAGER_REDUCE overrides."""
        mode = os.environ.get("DD_EAGER_REDUCE", "")
        if mode in ("flat", "ddp"):
            return mode
        return "flat" if self.device.type == "cuda" else "ddp"

    def reduce_eager_grads(self):
        """Behind backward() of an eager step under --ddp in 'flat' mode: the branch streams of the multi-stream backward are joined
        into the caller's stream, then ONE all-reduce averages the whole gradient buffer.  (No-op otherwise: DDP's hooks did it.)"""
        fg = getattr(self, "_flat_grads", None)
        if fg is None:
            return
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream()
            for st in list(getattr(self.base_model, "_streams", None) or ()) + [getattr(self.base_model, "_main_stream", None)]:
                if st is not None and st != cur:
                    cur.wait_stream(st)
        fg.all_reduce()

    def zero_grads(self, set_to_none=False):
        """optimizer.zero_grad() of an eager step; in 'flat' mode the gradient views stay attached and the buffer is zero-filled."""
        fg = getattr(self, "_flat_grads", None)
        if fg is not None:
            fg.zero()
        else:
            self.optim["optimizer"].zero_grad(set_to_none=set_to_none) if set_to_none else self.optim["optimizer"].zero_grad()

    def get_optim(self, network_names, optm=optim.Adam, lr_factor=1):
        kw = {}
        if optm is optim.Adam and self.opt.hip_graph and self.device.type == "cuda":
            kw["capturable"] = True
        if optm is optim.Adam and self.device.type == "cuda" and os.environ.get("DD_STOCK_ADAM", "0") != "1":
            kw["fused"] = True                   # same update rule, one multi-tensor kernel per chunk instead of ~45 launches
        opt_ = optm(self.base_model.parameters_by_names(network_names), self.opt.learning_rate * lr_factor, **kw)
        sched = optim.lr_scheduler.StepLR(opt_, self.opt.scheduler_step_size, 0.5)
        return {"optimizer": opt_, "lr_scheduler": sched, "network_names": network_names}

    # ---- hipGraph steps --------------------------------------------------------------------------------------------
    def _graph_mode(self):
        """'segments' (default): one forward and one backward graph per sub-network, replayed on their own streams
        (segments.SegmentedStep).  'whole': the single-stream whole-step capture of round 1 (DD_GRAPH=whole; single GPU only)."""
        mode = os.environ.get("DD_GRAPH", "segments")
        if mode == "segments" and not self.opt.fused_loss:
            mode = "whole"            # the per-network graphs are built around the fused loss (it leaves d loss / d outputs in place)
        return mode

    def drop_graphs(self):
        g = self._graph
        if g is not None and hasattr(g, "flush_counters"):
            g.flush_counters()
        self._graph = None

    def _graph_step(self, inputs):
        """Replays the captured step (capturing it on first use).  Only used while the loss weights are constant: they are
        launch-time scalars of the fused kernels."""
        if self._graph_mode() == "segments":
            self.upload_inputs(inputs)
            if self._graph is None:
                from segments import SegmentedStep
                self._graph = SegmentedStep(self, inputs)
            return self._graph.run(inputs)
        return self._whole_graph_step(inputs)

    def _whole_graph_step(self, inputs):
        """process_batch + backward + Adam as ONE hipGraph (static input buffers), single stream."""
        g = self._graph
        if g is None:
            self.process_inputs(inputs)
            static = {k: v.clone() for k, v in inputs.items() if torch.is_tensor(v)}
            optimizer = self.optim["optimizer"]
            # The warm-up iterations (allocator, MIOpen solver selection) must not count as training steps: snapshot, then
            # restore IN PLACE.  The optimizer state has to exist before the capture: Adam creates `step` / `exp_avg` /
            # `exp_avg_sq` lazily, and zero-fills captured inside the graph would be replayed on every step (the update would
            # degenerate to lr * sign(g)).  So the tensors the warm-up created stay, and are reset to the snapshot's values --
            # zeros when the phase's optimizer was fresh.
            model_state = {k: v.detach().clone() for k, v in self.base_model.state_dict().items()}
            optim_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                           for p, st in optimizer.state.items()}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    optimizer.zero_grad(set_to_none=True)
                    _, l = self.forward_and_losses(dict(static))
                    l["loss"].backward()
                    optimizer.step()
            torch.cuda.current_stream().wait_stream(side)
            with torch.no_grad():
                for k, v in self.base_model.state_dict().items():
                    v.copy_(model_state[k])
                for p, st in optimizer.state.items():
                    saved = optim_state.get(id(p))
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if saved is not None and k in saved:
                                v.copy_(saved[k])
                            else:
                                v.zero_()
            graph = torch.cuda.CUDAGraph()
            optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                outputs, losses = self.forward_and_losses(dict(static))
                losses["loss"].backward()
                optimizer.step()
            self._graph = g = {"graph": graph, "static": static, "outputs": outputs, "losses": losses}
            graph.replay()                       # the step for this batch
            return outputs, losses
        self.process_inputs(inputs)
        for k, v in g["static"].items():
            v.copy_(inputs[k], non_blocking=True)
        g["graph"].replay()
        return g["outputs"], g["losses"]

    # ===================================================================================================
    # data
    # ===================================================================================================
    def init_loaders(self):
        self.setup_train_loader(verbose=True)
        self.setup_val_loader()
        self.num_steps_per_epoch = len(self.train_loader)
        self.val_iter = iter(self.val_loader)
        self.print("Number of training items / batches:    {} / {}".format(len(self.train_dataset), len(self.train_loader)))
        self.print("Number of validation items / batches:  {} / {}\n".format(len(self.val_dataset), len(self.val_loader)))

    def _split_file(self, name):
        """<splits root>/<split>/<name>.  The split lists are data files of the reference checkout (splits/eigen_zhou/..., 4 MB,
        not shipped here): DYNAMO_SPLITS points at that directory, default `splits/` next to this file."""
        root = os.environ.get("DYNAMO_SPLITS", osp.join(osp.dirname(osp.abspath(__file__)), "splits"))
        path = osp.join(root, self.opt.split, name)
        if name == "train_files.txt" and not osp.exists(path):
            raise FileNotFoundError("{} not found: copy or link the reference's splits/ directory next to Trainer.py or set DYNAMO_SPLITS "
                                    "(--synthetic trains on generated frames of the dataset's shape instead)".format(path))
        return path

    def _world(self):
        return self.opt.local_world_size if self.opt.ddp else 1

    def setup_train_loader(self, verbose=False):
        """The epoch's loader.  The reference builds a new dataset over the epoch's random file subset and a new DataLoader every
        epoch (Trainer.py:519-531) -- on a GPU box that is 2-3 s of worker start-up per epoch even from the fork server.  Here
        ONE dataset over the whole split and ONE DataLoader with persistent workers live for the run; the epoch's subset is a
        sampler over it (EpochSubsetSampler) that draws from the same random streams in the same order as the reference's
        construction -- np.random.choice for the subset, the base seed of a fresh loader iterator (mirror_fresh_loader_draw), then
        the shuffle of a fresh RandomSampler / DistributedSampler -- so a run sees the SAMPLES the reference's construction would
        show it, in the same order, with or without workers (tests/test_loaders.py).  What differs with workers: persistent workers
        keep their augmentation streams across epochs where the reference's new workers are re-seeded every epoch -- statistically
        equivalent draws, not the same ones.  --fresh_loader_per_epoch restores the reference's construction."""
        o = self.opt
        if o.synthetic:
            count = o.batch_size * self._world() * (o.epoch_size if o.epoch_size > 0 else 64)
            files = ["synthetic {}".format(i) for i in range(count)]
            want = None
        else:
            files = getattr(self, "_train_files", None)
            if files is None:
                files = self._train_files = readlines(self._split_file("train_files.txt"))
            if verbose:
                self.print("Total number of available training examples: {}".format(len(files)))
            want = o.batch_size * self._world() * o.epoch_size if o.epoch_size > 0 else None
        fresh = getattr(o, "fresh_loader_per_epoch", False)
        if fresh and want is not None:
            files = np.random.choice(files, want, replace=(want > len(files)))
        if fresh or getattr(self, "_train_loader_key", None) != (len(files), o.ddp, self.B, o.num_workers):
            self.train_dataset = self.get_dataset(files, is_train=True, load_depth=False, load_mask=False)
            if fresh:
                sampler = DistributedSampler(self.train_dataset) if o.ddp else None
            else:
                sampler = EpochSubsetSampler(len(files), world=self._world() if o.ddp else 1, rank=self._rank())
            self.train_sampler = sampler
            self.train_loader = DataLoader(self.train_dataset, batch_size=self.B, shuffle=sampler is None, num_workers=o.num_workers,
                                           pin_memory=self.device.type == "cuda", drop_last=True, sampler=sampler,
                                           persistent_workers=(not fresh) and o.num_workers > 0,
                                           collate_fn=getattr(self.train_dataset, "collate", None), **self._worker_start())
            self._train_loader_key = None if fresh else (len(files), o.ddp, self.B, o.num_workers)
        if not fresh:
            n = len(files)
            mirror_fresh_loader_draw(self.train_loader)
            self.train_sampler.set_subset(None if want is None else np.random.choice(n, want, replace=(want > n)))

    def _worker_start(self):
        """How DataLoader workers are started.  On a GPU: from a fork SERVER -- forking THIS process, which maps the device's
        address ranges, took ~5 s per worker on an MI355X box (40 s for the 8 workers of every epoch's new loader, and of every
        re-created validation iterator); the server is a small process started once, workers import their modules in parallel
        (~2 s).  `--loader_start fork` restores the stock behaviour."""
        how = getattr(self.opt, "loader_start", None)
        if how is None:
            how = "forkserver" if self.device.type == "cuda" else "fork"
        if self.opt.num_workers == 0 or how == "fork":
            return {}
        return {"multiprocessing_context": how}

    def setup_val_loader(self):
        o = self.opt
        if o.synthetic:
            files = ["synthetic {}".format(i) for i in range(4 * max(o.batch_size * self._world(), 8))]
        else:
            val_path = self._split_file("val_files.txt")
            files = readlines(val_path if osp.exists(val_path) else self._split_file("train_files.txt"))
        self.val_dataset = self.get_dataset(files, is_train=False, load_depth=True, load_mask=False)
        sampler = DistributedSampler(self.val_dataset) if o.ddp else None
        # its own generator: re-creating the validation iterator must not draw from the stream that orders the training data
        # (a resumed run re-creates it at a different step than the run that wrote the checkpoint)
        self.val_loader = DataLoader(self.val_dataset, batch_size=self.B, shuffle=sampler is None, num_workers=o.num_workers,
                                     pin_memory=self.device.type == "cuda", drop_last=True, sampler=sampler,
                                     generator=torch.Generator().manual_seed(0), persistent_workers=o.num_workers > 0,
                                     collate_fn=getattr(self.val_dataset, "collate", None),
                                     **self._worker_start())          # (persistent: a new pass over the set re-uses the workers)

    def get_dataset(self, filenames, is_train=False, load_depth=False, load_mask=False, **kwargs):
        o = self.opt
        if not o.synthetic and self.device.type == "cuda" and getattr(o, "device_preprocess", True):
            kwargs.setdefault("device_preprocess", True)
            kwargs.setdefault("device_decode", getattr(o, "device_decode", True))
            kwargs.setdefault("device_resize", getattr(o, "device_resize", True))
        return self.dataset(data_path=o.data_path, filenames=filenames, height=o.height, width=o.width, cam_name=o.cam_name,
                            img_type=o.train_img_type, frame_idxs=o.frame_ids, num_scales=len(o.scales), is_train=is_train,
                            img_ext=o.img_ext, load_depth=load_depth, load_mask=load_mask, **kwargs)

    def process_inputs(self, inputs):
        """Upload, then finish the samples on the device: ToTensor / flip / per-frame ColorJitter of the uint8 frames when the
        loader hands those over (--device_preprocess; reference datasets/base_dataset.py:83-95 does it per sample on the host),
        and the target pyramid (the reference resizes on the host first, Trainer.py:722-734)."""
        self.upload_inputs(inputs)
        self.derive_inputs(inputs)

    def derive_inputs(self, inputs):
        """What a step derives from the uploaded batch on the device: the target pyramid, and the source frames' pixel-interleaved copies."""
        self.apply_img_resize(inputs)
        self.pack_sources(inputs)

    def pack_sources(self, inputs):
        """inputs[('color_packed', f)]: (B,H,W,3) copies of the source frames for the photometric kernel's gather (hipops.inputs.pack_rgb,
        DDPhotoArgs.source_packed): one 12-byte pixel per bilinear tap instead of three planes, at every scale -- one HBM-bound launch
        per frame and step (12 us for both at the KITTI batch) against 17-27 us less in the photometric kernel
        (profiles/r05_photo_gather_ablation.txt).  The reference has no counterpart (F.grid_sample reads the planar tensor)."""
        if self.device.type != "cuda" or not self.opt.fused_loss or os.environ.get("DD_PACK_SOURCES", "1") != "1" or (self.H * self.W) % 4:
            return
        from hipops.inputs import pack_rgb
        for f in self.opt.frame_ids[1:]:
            img = inputs.get(("color", f, 0))
            if ("color_packed", f) in inputs or img is None or not img.is_cuda or img.dtype != torch.float32 or tuple(img.shape[1:]) != (3, self.H, self.W):
                continue
            inputs[("color_packed", f)] = pack_rgb(img)

    def upload_inputs(self, inputs):
        groups = None
        if "jpeg_hdr" in inputs and not inputs["jpeg_hdr"].is_cuda:
            # the frames arrive compressed: read every frame's geometry from its header record while the records are on the host
            # and group the frames of equal geometry (one decode per group; KITTI's originals come in four sizes)
            rec = inputs["jpeg_hdr"].reshape(-1, inputs["jpeg_hdr"].shape[-1]).numpy().view(np.int32)      # DDJpegHeader: width, height at words 2, 3; ncomp 5; h 6..8; v 9..11
            groups = {}
            for i in range(rec.shape[0]):
                key = (int(rec[i, 3]), int(rec[i, 2]), int(rec[i, 5]), tuple(int(x) for x in rec[i, 6:9]), tuple(int(x) for x in rec[i, 9:12]))
                groups.setdefault(key, []).append(i)
        for key, value in inputs.items():
            if torch.is_tensor(value) and value.device != self.device:
                inputs[key] = value.to(self.device, non_blocking=True)
        if groups is not None:
            from hipops.jpeg import decode_batch        # Huffman + IDCT + chroma up-sampling + colour conversion on the device
            data, hdr = inputs.pop("jpeg_bytes"), inputs.pop("jpeg_hdr")
            lead = data.shape[:-1]
            data, hdr = data.reshape(-1, data.shape[-1]), hdr.reshape(-1, hdr.shape[-1])
            if len(groups) == 1 and next(iter(groups))[:2] == (self.H, self.W):
                inputs["frames_u8"] = decode_batch(data, hdr, *next(iter(groups))).view(*lead, self.H, self.W, 3)
            else:
                # frames of other sizes: Pillow's bicubic resize on the device (reference: transforms.Resize(BICUBIC) on the PIL
                # frames, datasets/base_dataset.py:80,147), each group written into its slots of the batch
                from hipops.resize import resize_batch
                frames = torch.empty((data.shape[0], self.H, self.W, 3), dtype=torch.uint8, device=self.device)
                for geom, idx in groups.items():
                    sel = torch.as_tensor(idx, device=self.device)
                    resize_batch(decode_batch(data[sel], hdr[sel], *geom), self.H, self.W, out=frames, slots=idx)
                inputs["frames_u8"] = frames.view(*lead, self.H, self.W, 3)
        if "frames_u8" in inputs:
            from hipops.inputs import prepare_frames
            color, aug = prepare_frames(inputs.pop("frames_u8"), inputs.pop("jitter"), inputs.pop("flip"))
            for i, f in enumerate(self.opt.frame_ids):
                inputs[("color", f, 0)], inputs[("color_aug", f, 0)] = color[i], aug[i]

    def apply_img_resize(self, inputs):
        for s in self.opt.scales:
            if s != 0 and ("color", 0, s) not in inputs:
                prev = inputs[("color", 0, s - 1)]
                if prev.is_cuda and prev.shape[-2] == 2 * (self.H >> s) and prev.shape[-1] == 2 * (self.W >> s):
                    from hipops.inputs import pyramid_down2
                    inputs[("color", 0, s)] = pyramid_down2(prev)              # one HIP launch: resize + clamp
                else:
                    inputs[("color", 0, s)] = torch.clamp(self.resize[s](prev), 0, 1)

    # ===================================================================================================
    # logging / checkpoints
    # ===================================================================================================
    def vis_motion(self, depth, K, inv_K, motion_map=None, camTcam=None, scale=0):
        """Optical-flow visualisation of a 3-D motion field and/or an ego-motion (reference Trainer.py:574-605)."""
        assert motion_map is not None or camTcam is not None, "At least one form of motion is supplied"
        b, _, h, w = depth.shape
        ident = make_ind_map(h, w).to(self.device)
        pts = self.backproject_depth[scale](depth, inv_K)
        err = self.project_3d[scale](pts, K, None)[0] - ident
        if motion_map is not None:
            pts = torch.cat([pts[:, :3] + motion_map.reshape(b, 3, h * w), pts[:, 3:]], 1)
        raw = self.project_3d[scale](pts, K, camTcam)[0] - ident - err
        mag, theta = cart2polar(raw)
        max_mag = mag.max().item() + 1e-8
        hsv = torch.ones(b, 3, h, w).to(self.device)
        hsv[:, 0] = (theta - torch.pi / 4) % (2 * torch.pi) / (2 * torch.pi)
        hsv[:, 2] = mag / max_mag
        return 1 - hsv_to_rgb(hsv), hsv, max_mag

    def reduce_losses(self, losses):
        """Scalar loss values as floats, averaged over the ranks under --ddp (ONE all-reduce of the stacked vector): the curves
        of a multi-GPU run then describe the global batch.  The reference logs rank 0's local values (Trainer.py:610, every
        rank calls wandb)."""
        keys = [k for k, v in losses.items() if torch.is_tensor(v) and v.numel() == 1]
        out = {k: v for k, v in losses.items() if k not in keys}
        if not keys:
            return out
        vec = torch.stack([losses[k].detach().float().reshape(()) for k in keys])
        if self.opt.ddp and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(vec, op=torch.distributed.ReduceOp.SUM)
            vec = vec / torch.distributed.get_world_size()
        out.update({k: float(x) for k, x in zip(keys, vec.cpu())})
        return out

    def vis_rows(self, inputs, outputs, frame_id=-1, s=0):
        """The three image rows of the reference's log() (Trainer.py:615-651) as one (B,3,3H,3W) tensor in [0,1]:
        rgb | reconstruction | scaled L1;  disparity | motion mask | depth / max;  ego | independent | total flow (HSV wheel).
        Works on what either loss path leaves in `outputs`: the fused path does not keep the full-resolution
        ('independ_flow', f, 0) -- at scale 0 it is residual_flow x motion_mask (Trainer.py:252-253,284 with interp = identity)."""
        color, recon = inputs[("color", 0, 0)], outputs[("color", frame_id, 0)]
        l1 = torch.abs(color - recon).mean(1, keepdim=True)
        l1 = l1 / (l1.max() + 1e-6)
        disp = outputs[("disp", 0, s)].detach()
        B, _, h, w = disp.shape
        mask = outputs.get(("motion_mask", frame_id, 0))
        mask = torch.ones_like(disp) if mask is None else mask.detach()
        _, depth = disp_to_depth(disp, self.opt.min_depth, self.opt.max_depth)
        motion = outputs.get(("independ_flow", frame_id, s))
        if motion is None:
            resid = outputs.get(("residual_flow", frame_id, s))
            motion = torch.zeros(B, 3, h, w, device=disp.device) if resid is None else resid.detach() * mask
        K, inv_K, T = inputs[("K", s)], inputs[("inv_K", s)], outputs[("cam_T_cam", 0, frame_id)].detach()
        _, ego_hsv, ego_mag = self.vis_motion(depth, K, inv_K, motion_map=None, camTcam=T, scale=s)
        _, ind_hsv, ind_mag = self.vis_motion(depth, K, inv_K, motion_map=motion.detach(), camTcam=None, scale=s)
        _, tot_hsv, tot_mag = self.vis_motion(depth, K, inv_K, motion_map=motion.detach(), camTcam=T, scale=s)
        top = max(ind_mag, ego_mag, tot_mag)
        flows = []
        for hsv, mag in ((ego_hsv, ego_mag), (ind_hsv, ind_mag), (tot_hsv, tot_mag)):
            hsv[:, 2] = torch.clamp(hsv[:, 2] * mag / top, 0, 1)
            flows.append(1 - hsv_to_rgb(hsv))
        three = lambda x: x.repeat(1, 3, 1, 1)      # noqa: E731
        row1 = torch.cat((color, recon.detach(), three(l1.detach())), 3)
        row2 = torch.cat((three(disp), three(mask), three(depth / depth.flatten(1).max(1)[0].view(B, 1, 1, 1))), 3)
        row3 = torch.cat(flows, 3)
        return torch.cat((row1, row2, row3), 2)

    def log(self, mode, inputs, outputs, losses):
        package = {"{}_{}".format(mode, k): v for k, v in self.reduce_losses(losses).items()}
        if not self.opt.no_train_vis and wandb is not None and self.is_main() and ("color", -1, 0) in outputs:
            rows = self.vis_rows(inputs, outputs)
            for j in range(min(self.B, rows.shape[0])):
                package["vis/{}_{}".format(mode, j)] = wandb.Image(rows[j])
        if self.is_main():
            self.wandb_log(package)
        return package

    def wandb_log(self, package):
        if wandb is None:
            return
        try:
            wandb.log(package, step=self.g_step)
        except Exception:
            pass

    def log_time(self, batch_idx, duration, loss, data_time, gpu_time):
        if not self.is_main():
            return
        sofar = time.time() - self.start_time
        left = (self.num_total_steps / self.step - 1.0) * sofar if self.step > 0 else 0
        print("epoch {:>3} | batch {:>6} | examples/s: {:5.1f} | loss: {:.5f} | time elapsed: {} | time left: {} | CPU/GPU time: {:0.1f}s/{:0.1f}s".format(
            self.epoch, batch_idx, self.B / duration, float(loss), sec_to_hm_str(sofar), sec_to_hm_str(left), data_time, gpu_time))

    def save_opt(self):
        if not self.is_main():
            return
        models_dir = join_dir(self.log_path, "models")
        dump = {k: v for k, v in self.opt.__dict__.items()}
        if self.opt.print_opt:
            for k, v in dump.items():
                print("{:30}{}".format(k + ":", v))
        with open(osp.join(models_dir, "opt.json"), "w") as fh:
            json.dump(dump, fh, indent=2)

    def save_model(self, save_name="weights"):
        """Per-module .pth + adam.pth (reference layout, Trainer.py:697-707) plus what a bit-exact continuation needs and the
        reference does not keep: scheduler, phase / epoch / step counters (resume.json) and the random-number streams of
        this rank (rng.pth: torch CPU + device generators, NumPy, Python -- the epoch's file draw, DropPath masks, auto-mask
        noise and RANSAC draws all come from them)."""
        folder = osp.join(self.log_path, "models", "{}_{:02}".format(save_name, self.epoch))
        if self._graph is not None and hasattr(self._graph, "flush_counters"):
            self._graph.flush_counters()          # num_batches_tracked of the replayed steps
        self.sync_buffers()          # under --ddp: rank 0's BatchNorm running statistics are the ones saved and evaluated
        # every rank keeps its OWN random-number streams (data order, DropPath masks, auto-mask noise, RANSAC draws differ per
        # rank): rng.pth is rank 0's (the single-GPU name), rng_rank<r>.pth the others'
        rng = {"torch": torch.get_rng_state(), "numpy": np.random.get_state(), "python": random.getstate()}
        if self.device.type == "cuda":
            rng["device"] = torch.cuda.get_rng_state(self.device)
        rank = self._rank()
        ddp = self.opt.ddp and torch.distributed.is_available() and torch.distributed.is_initialized()
        if self.is_main():
            folder = join_dir(self.log_path, "models", "{}_{:02}".format(save_name, self.epoch))
        if ddp:
            torch.distributed.barrier()           # the folder exists on every rank's side of this (no polling, no time-out)
        if not self.is_main():
            if osp.isdir(folder) or self._wait_for(folder):       # (_wait_for: a shared file system that lags behind the barrier)
                torch.save(rng, osp.join(folder, "rng_rank{}.pth".format(rank)))
            else:
                raise RuntimeError("rank {}: checkpoint folder {} did not appear; rng_rank{}.pth not written".format(rank, folder, rank))
            return
        self.base_model.save(folder)
        torch.save(self.optim["optimizer"].state_dict(), osp.join(folder, "adam.pth"))
        torch.save(rng, osp.join(folder, "rng.pth"))
        if getattr(self, "_scaler", None) is not None:
            torch.save(self._scaler.state_dict(), osp.join(folder, "scaler.pth"))      # fp16: loss scale and its growth tracker
        with open(osp.join(folder, "resume.json"), "w") as fh:
            json.dump({"phase": save_name, "epoch": self.epoch, "step": self.step, "g_step": self.g_step,
                       "scheduler": self.optim["lr_scheduler"].state_dict(), "num_steps_per_epoch": self.num_steps_per_epoch}, fh)

    def restore_training_state(self, record):
        """Second half of --resume (the weights were loaded in __init__): optimizer moments, scheduler, counters and random
        streams of the checkpoint written at the END of epoch record['epoch'] of the current phase.  Returns the epoch to
        continue with."""
        folder = osp.expanduser(self.opt.resume)
        state = torch.load(osp.join(folder, "adam.pth"), map_location=self.device)
        self.optim["optimizer"].load_state_dict(state)
        if "scheduler" in record:
            self.optim["lr_scheduler"].load_state_dict(record["scheduler"])
        self.step, self.g_step = int(record["step"]), int(record["g_step"])
        scaler_path = osp.join(folder, "scaler.pth")
        if osp.exists(scaler_path) and self._grad_scaler() is not None:
            self._scaler.load_state_dict(torch.load(scaler_path, map_location="cpu"))
        # this rank's own streams; a checkpoint without the per-rank file (written by fewer ranks) falls back to rank 0's with
        # the rank folded into the seeds, so that the ranks do not draw the same numbers
        rank = self._rank()
        rng_path = osp.join(folder, "rng.pth" if rank == 0 else "rng_rank{}.pth".format(rank))
        if rank != 0 and not osp.exists(rng_path):
            torch.manual_seed(torch.initial_seed() + 7919 * rank)
            np.random.seed((int(np.random.get_state()[1][0]) + 7919 * rank) % (2 ** 32))
            random.seed(7919 * rank)
        if osp.exists(rng_path):
            rng = torch.load(rng_path, map_location="cpu", weights_only=False)
            torch.set_rng_state(rng["torch"])
            np.random.set_state(rng["numpy"])
            random.setstate(rng["python"])
            if "device" in rng and self.device.type == "cuda":
                torch.cuda.set_rng_state(rng["device"], self.device)
        self.drop_graphs()
        self.print("resumed {} after epoch {} (step {}, lr {})".format(record["phase"], record["epoch"], self.step,
                                                                        self.optim["optimizer"].param_groups[0]["lr"]))
        return int(record["epoch"]) + 1

    def _rank(self):
        if self.opt.ddp and torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank()
        return 0

    @staticmethod
    def _wait_for(path, seconds=30.0):
        t0 = time.time()
        while not osp.isdir(path) and time.time() - t0 < seconds:
            time.sleep(0.05)
        return osp.isdir(path)

    def sync_buffers(self):
        """Under --ddp: every rank takes rank 0's module buffers (the BatchNorm running statistics).  The wrapper runs with
        broadcast_buffers=False -- training-mode arithmetic never reads them, so the reference's per-step broadcast
        (Trainer.py:44, DDP default) is one collective per step for nothing -- but the statistics that are EVALUATED (val())
        and SAVED are then rank 0's, exactly as in the reference, where every forward starts from rank 0's copy."""
        if not (self.opt.ddp and torch.distributed.is_available() and torch.distributed.is_initialized()):
            return
        for m in self.base_model.modules():
            if hasattr(m, "flush_counter"):
                m.flush_counter()
        bufs = [b for b in self.base_model.buffers() if b.numel() > 0]
        if not bufs:
            return
        by_dtype = {}
        for b in bufs:
            by_dtype.setdefault(b.dtype, []).append(b)
        for group in by_dtype.values():
            flat = torch.cat([b.detach().reshape(-1) for b in group])
            torch.distributed.broadcast(flat, src=0)
            torch._foreach_copy_([b.detach() for b in group], [c.view_as(b) for b, c in zip(group, flat.split([b.numel() for b in group]))])

    def load_model(self):
        self.base_model.load(verbose=self.is_main())

    def setup_wandb(self):
        if wandb is not None and self.is_main():
            wandb.init(project="Dynamo", name=self.opt.model_name, notes=self.opt.comment, config=vars(self.opt))

    def is_main(self):
        return self.local_rank == 0

    def print(self, s=""):
        if self.is_main():
            print(s)

    def set_train(self):
        self.base_model.set_train()

    def set_eval(self):
        self.base_model.set_eval()
