"""A training step as a handful of hipGraphs, one per sub-network, replayed side by side on their own HIP streams.

Why: the eager step is bound by the host (round 2: 50.05 ms to enqueue 50.69 ms of step: ~2 600 launches, ~1 100 Python
autograd.Function + ctypes crossings), and ONE whole-step graph gives the host back but serialises the branches of the
forward (58 ms of kernels back to back) -- parallel branches inside one captured graph are not dependable on this ROCm
stack.  Here every branch is its own single-stream capture (the dependable kind):

    inputs   target pyramid                                                        (current stream)
    depth    depth net on the target frame, forward | backward                     (current stream)
    side     statistics-only depth passes on frames -1/+1 as one batch, no tape    (stream 1)
    pose     both pose passes as one batch + pose-vector -> 4x4, forward | backward   (stream 2)
    menc     motion encoder, forward | backward                                    (stream 3, starts with the step)
    motion   flow decoder + mask decoder, forward | backward                       (stream 4, after pose and menc forward)
    fold     deferred BatchNorm statistics of the side batch into the running buffers (stream 1, after depth forward)
    loss     fused view-synthesis loss AND d loss / d network outputs               (current stream)
    optim    fused Adam over the phase's parameters                                (current stream)

and a step is ~14 graph launches with stream waits between them.  There is no autograd engine at replay time: the loss graph
leaves d loss / d (network outputs) in fixed buffers, the backward graphs were captured with exactly those buffers as their
grad_outputs, and the parameter gradients land in one flat buffer per segment (p.grad are views of it).

What long runs taught: ROCm 7.2's graph "packet capture" launch path must be off
(DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, set by miopen_env / Trainer before the first device call) or replayed training turns
non-finite within a few hundred steps; every graph is captured behind a cleared rocBLAS / hipBLASLt workspace cache (graphs recorded
on one stream and replayed on several would share one workspace); the host stays at most DD_SEG_RUN_AHEAD (2) steps ahead of the
GPU; the step's cross-stream events live until the GPU has passed them.  Debugging switches: DD_SEG_CHECK=1 (finiteness of every
buffer after each replay, with a host sync), DD_SEG_PROBE=1 (the same as device-side flags, no sync; probe_report()),
DD_SEG_SERIAL=all|name,.. (those segments on the caller's stream), DD_SEG_CHECK_AT=k, DD_SEG_TIMING=1 (timeline()),
DD_SEG_SIDE_LATE=0, DD_SEG_EXEC_GUARD=1.

Multi-GPU: the parameter gradients of all segments live in ONE flat buffer (a contiguous slice per segment), averaged over the ranks
(RCCL) by one all-reduce behind the last backward graph, in front of Adam's graph (DD_SEG_REDUCE=overlap: round 3's scheme, one
collective per segment issued behind that segment's backward graph -- measured slower: RCCL's stream is a fifth hardware queue
beside the four the branches occupy).  (The eager steps of a run -- ramp-up, log steps -- go through the DDP
wrapper; both average the same gradients.)

Reference: this replaces the body of the training loop, Trainer.py:145-151 (process_batch, backward, optimizer step).
"""
import os
import time

import torch
import torch.distributed as dist


class _empty_graph_watch:
    """Context manager around a graph capture: `.empty` says whether torch reported "The CUDA Graph is empty" (zero nodes) when the
    capture ended.  That warning is swallowed, every other warning is passed on."""

    def __enter__(self):
        import warnings
        self.empty = False
        self._cm = warnings.catch_warnings(record=True)
        self._log = self._cm.__enter__()
        warnings.simplefilter("always")
        return self

    def __exit__(self, *exc):
        import warnings
        self._cm.__exit__(*exc)
        for w in self._log:
            if "Graph is empty" in str(w.message):
                self.empty = True
            else:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        return False


def _is_float(t):
    return torch.is_tensor(t) and t.is_floating_point()


class _Segment:
    def __init__(self, name, stream):
        self.name, self.stream = name, stream
        self.pool = torch.cuda.graph_pool_handle()
        self.fwd = self.bwd = None
        self.views = ()             # views of `flat`, one per parameter, with the parameter's own strides
        self.outs = ()              # forward outputs that carry a tape (static buffers)
        self.params = []
        self.flat = None            # flat gradient buffer; p.grad are views of it
        self.work = None


class SegmentedStep:
    """Captures on construction (after eager warm-up steps whose effect on weights / optimizer state is undone), then
    run(inputs) copies the batch into the static input buffers and replays."""

    AUTO_PROBE = 6          # replays per placement of the DD_SEG_REDUCE=auto probe (the first of each is not counted)

    WARMUP = 3

    def __init__(self, trainer, inputs):
        if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "") != "0":
            import warnings
            warnings.warn("DEBUG_CLR_GRAPH_PACKET_CAPTURE is not 0: on ROCm 7.2 replayed training steps turn non-finite within a few hundred "
                          "steps with the runtime's graph packet capture on (see miopen_env.py); set it before the first device call")
        self.tr = tr = trainer
        self.model = model = tr.base_model
        self.opt = tr.opt
        self.replays = 0            # since the last flush_counters() (BatchNorm bookkeeping)
        self.total_replays = 0      # monotonic: what the DD_SEG_REDUCE=auto probe counts -- a checkpoint inside the probe (save_model ->
        self._bn_delta = []         # flush_counters) must neither restart it nor let one rank decide at another step than the others
        self.main = torch.cuda.Stream()             # capture stream of the segments that replay on the caller's stream (the
        if os.environ.get("DD_STREAM_REPICK", "0") == "1":         # (experiment: measure the streams' queues again at capture time)
            from hipops import queues
            queues._cache.clear()
            model._streams = None
        self.side_streams = list(model.side_streams())      # legacy default stream cannot capture)
        self.side_late = os.environ.get("DD_SEG_SIDE_LATE", "1") != "0"
        # WHEN the statistics-only side batch runs (it feeds nothing but the BatchNorm buffers, so any time between the inputs and the
        # optimizer is correct): "start" = beside the forward passes, "loss" = beside the backward passes (issued behind the loss),
        # "pose" = behind the pose backward on the pose stream (the backward's short branch)
        self.side_at = os.environ.get("DD_SEG_SIDE_AT", "start") if self.side_late else "start"
        self.run_ahead = int(os.environ.get("DD_SEG_RUN_AHEAD", "2"))       # steps the host may be ahead of the GPU (0 = unbounded)
        self._ends = []
        self._events = []
        self.host_wait_s = 0.0
        self.exec_guard = os.environ.get("DD_SEG_EXEC_GUARD", "0") == "1"      # (debugging: one launch of a graph exec in flight at a time)
        self._last_launch = {}
        self.serial = set(x for x in os.environ.get("DD_SEG_SERIAL", "").split(",") if x)
        self.check_at = int(os.environ.get("DD_SEG_CHECK_AT", "-1"))
        self.probe = os.environ.get("DD_SEG_PROBE", "0") == "1"
        self._ring = None
        self.check = os.environ.get("DD_SEG_CHECK", "0") == "1"            # finiteness of every buffer after each replay (debugging)
        self.timing = os.environ.get("DD_SEG_TIMING", "0") == "1"       # events around every replay (timeline(); bench.py prints it)
        self.marks = []
        self.loss_events = None
        self.hp_stream = torch.cuda.Stream(priority=-1) if os.environ.get("DD_SEG_PRIORITY", "0") == "1" else None
        self.time_tile_kernel = bool(getattr(trainer, "time_tile_kernel", False))
        self.static = {k: v.clone() for k, v in inputs.items() if torch.is_tensor(v) and not self._is_pyramid_key(k)}
        self.ddp = bool(self.opt.ddp and dist.is_available() and dist.is_initialized())
        self.world = dist.get_world_size() if self.ddp else 1
        # How the gradients are averaged under --ddp.  "end" (default): ONE all-reduce of the whole gradient buffer behind the last
        # backward graph.  "overlap": one collective per segment, issued behind that segment's backward graph (round 3's design).
        # Measured with a one-rank RCCL group (round 4): RCCL's stream is one more hardware queue next to the
        # four the step's branches occupy, and while collectives are in flight beside the backward graphs every branch is
        # time-sliced -- 52.98 ms per step against 45.74 without a process group, with NOTHING to exchange.
        # "auto" (the default with more than one rank): the first replays time BOTH placements on this node -- AUTO_PROBE steps of
        # "end", then as many of "overlap" (both are correct at every step) -- the per-step times are MAX-reduced over the ranks and
        # the faster one stays; `reduce_mode_chosen` / `reduce_probe_ms` tell bench.py which and by how much.  With one rank there is
        # nothing to hide: "end".
        self.reduce_mode = os.environ.get("DD_SEG_REDUCE", "auto" if self.world > 1 else "end")
        self.reduce_probe_ms = None
        self._probe_marks = {"end": [], "overlap": []}
        if self.reduce_mode == "auto" and not self.ddp:
            self.reduce_mode = "end"
        self.reduce_mode_chosen = None if self.reduce_mode == "auto" else self.reduce_mode
        # fp16 networks: the dynamic loss scaler lives ON THE DEVICE inside the graphs -- the loss graph multiplies d loss / d outputs
        # by the scale tensor, the optimizer graph holds the non-finite check of the flat gradient buffers, the fused Adam kernel
        # with its skip-on-overflow predicate (found_inf) and in-kernel unscaling, and the scale update (_amp_update_scale_): all
        # device-side, no host decision per step (round 3 kept the fp16 step eager for the scaler: host-bound at 29 of 31 ms)
        self.scaler = trainer._grad_scaler()
        self._scaler_state = {}
        # with a process group alive, its watchdog thread polls events while this thread captures: only THIS thread's calls may
        # invalidate a capture (the default, "global", lets any thread's event query do so)
        self.capture_mode = "thread_local" if self.ddp else "global"
        self._warm_up()
        self._capture()

    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _is_pyramid_key(k):
        # derived on the device by the inputs graph (Trainer.derive_inputs): the target pyramid, the packed copies of the source frames
        return isinstance(k, tuple) and ((len(k) == 3 and k[0] == "color" and k[1] == 0 and k[2] != 0) or k[0] == "color_packed")

    def _warm_up(self):
        """Eager steps through the same code (allocator, MIOpen solver selection, lazily created Adam state), after which
        weights, buffers and optimizer state are put back IN PLACE: the warm-up is not training.  The optimizer state has to
        exist before the capture -- zero-fills captured inside the optimizer graph would be replayed on every step."""
        tr, optimizer = self.tr, self.tr.optim["optimizer"]
        model_state = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        optim_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                       for p, st in optimizer.state.items()}
        pending = [(m, m._pending_batches) for m in self.model.modules() if hasattr(m, "_pending_batches")]
        rng = torch.cuda.get_rng_state(tr.device)
        scaler = self.scaler
        for _ in range(self.WARMUP):
            optimizer.zero_grad(set_to_none=True)
            batch = dict(self.static)
            tr.derive_inputs(batch)
            _, losses = tr.forward_and_losses(batch)
            if scaler is None:
                losses["loss"].backward()
                optimizer.step()
            else:
                if scaler._scale is not None and "scale" not in self._scaler_state:
                    self._scaler_state = {"scale": scaler._scale.clone(), "tracker": scaler._growth_tracker.clone()}
                scaler.scale(losses["loss"]).backward()          # (creates the scale / growth tracker tensors on first use)
                if "scale" not in self._scaler_state:
                    self._scaler_state = {"scale": scaler._scale.clone().fill_(scaler._init_scale), "tracker": scaler._growth_tracker.clone().zero_()}
                scaler.step(optimizer)
                scaler.update()
        torch.cuda.synchronize()
        if scaler is not None:                  # the warm-up is not training: the loss scale and its growth tracker as they were
            scaler._scale.copy_(self._scaler_state["scale"])
            scaler._growth_tracker.copy_(self._scaler_state["tracker"])
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(model_state[k])
            for p, st in optimizer.state.items():
                saved = optim_state.get(id(p))
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if saved is not None and k in saved:
                            v.copy_(saved[k])
                        else:
                            v.zero_()
        for m, n in pending:
            m.__dict__["_pending_batches"] = n
        torch.cuda.set_rng_state(rng, tr.device)
        optimizer.zero_grad(set_to_none=True)

    # ------------------------------------------------------------------------------------------------------------------
    def _capture(self):
        tr, model, o = self.tr, self.model, self.opt
        main = self.main
        s_side, s_dec, s_pose, s_mot = self.side_streams
        if os.environ.get("DD_SEG_DEC_STREAM", "shared") != "own":
            # The decoders follow their encoder anyway: one stream for both leaves caller | side batch | pose | motion = four
            # streams for the four hardware queues (networks.Model.side_streams picks them on distinct queues).  Round 3 gave
            # the decoders a fifth stream, and pose shared a queue with the motion encoder: it ran behind it in both directions.
            s_dec = s_mot
        frames = list(o.frame_ids)
        target, sources = frames[0], frames[1:]
        motions = bool(model.bool_CmpFlow or model.bool_MotMask)
        stat = self.static
        outputs = {}
        bn_before = [(m, m._pending_batches) for m in model.modules() if hasattr(m, "_pending_batches")]
        # Fresh autograd leaves for the trainable parameters.  Autograd keeps ONE AccumulateGrad node per leaf, bound to the
        # stream that was current when the node was created, and it lives as long as any graph (or DDP's reducer) refers to
        # it.  A node left over from an eager step is bound to another stream than the capture stream; the backward then
        # hands the gradient over with an event that the OTHER stream waits on -- which pulls that stream into the capture,
        # never to be joined again (hipStreamEndCapture crashed on exactly this).  Aliases of the parameters (same storage,
        # new leaves, first used inside the capture) cannot have a history: the networks are captured on them.
        import gc
        from torch.nn.utils.stateless import _reparametrize_module
        gc.collect()
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        with torch.cuda.stream(main):
            alias = {n: p.detach().requires_grad_() for n, p in named}
        alias_of = {id(p): alias[n] for n, p in named}
        import contextlib
        amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(getattr(o, "amp", "none"))

        @contextlib.contextmanager
        def swap():
            """The networks on the parameter aliases (a module registered under two names -- PoseDecoder.net -- is swapped once),
            under autocast when the run uses reduced-precision networks (Trainer.run_networks)."""
            with _reparametrize_module(model, alias):
                if amp_dtype is None:
                    yield
                else:
                    with torch.autocast("cuda", dtype=amp_dtype):
                        yield
        torch.cuda.synchronize()

        dbg = os.environ.get("DD_SEG_DEBUG", "")

        def capture(seg, fn, stream, what="fwd"):
            """Every graph is captured on ONE stream (self.main), whatever stream it is replayed on later: autograd ties
            each node to the stream of its forward and synchronises streams with events when gradients cross from one to
            another -- under capture such an event drags a second stream into the capture (the backward capture of a segment
            whose forward had been captured on another stream than the loss graph crashed inside hipStreamEndCapture)."""
            if dbg:
                print("[segments] capturing {} {}".format(seg.name, what), flush=True)
            g = torch.cuda.CUDAGraph()
            stream = self.main if "m" not in dbg else stream
            self._private_blas_workspace()
            with _empty_graph_watch() as watch:
                with torch.cuda.graph(g, pool=seg.pool, stream=stream, capture_error_mode=self.capture_mode):
                    res = fn()
            if dbg:
                print("[segments] captured  {} {}{}".format(seg.name, what, " (empty: not replayed)" if watch.empty else ""), flush=True)
            # a segment the phase gives nothing to do (the statistics-only batch of an eval-mode run, ...) is not replayed at all:
            # no launch, and no "graph is empty" warning that would be indistinguishable from an empty OPTIMIZER graph (an error below)
            return (None if watch.empty else g), res

        # ---- inputs: the target pyramid (Trainer.apply_img_resize, reference Trainer.py:729-734) ------------------------
        self.inputs_seg = seg = _Segment("inputs", main)
        batch = dict(stat)

        def f_inputs():
            tr.derive_inputs(batch)
        seg.fwd, _ = capture(seg, f_inputs, main)
        self.batch = batch

        # ---- forward graphs -----------------------------------------------------------------------------------------------
        from networks.layers import DeferredStats, BatchNorm2d
        skip_side = bool(getattr(o, "skip_unused_depth_frames", False))
        self.segs = []

        depth = _Segment("depth", main)

        def f_depth():
            with swap():
                model.predict_depths(batch, outputs, frames=[target])
        depth.fwd, _ = capture(depth, f_depth, main)
        self.segs.append(depth)

        self.collectors = []
        side = None
        if not skip_side:
            side = _Segment("side", s_side)
            bn_floats = sum(2 * m.num_features for mod in (model.depth_enc, model.depth_dec) for m in mod.modules() if isinstance(m, BatchNorm2d))

            def f_side():
                cols = [DeferredStats(tr.device, max(bn_floats, 1)) for _ in sources]
                self.collectors = cols
                with swap():
                    model.predict_depths(batch, outputs, frames=list(sources), collectors=cols)
            side.fwd, _ = capture(side, f_side, s_side)
            self.segs.append(side)

        pose = _Segment("pose", s_pose)

        def f_pose():
            with swap():
                model.predict_poses(batch, outputs)
        pose.fwd, _ = capture(pose, f_pose, s_pose)
        self.segs.append(pose)

        # The motion encoder does not need the pose networks -- only the decoders read the (detached) pose vectors -- so it is a
        # segment of its own that starts with the step; the decoders follow on another stream once pose and encoder are done.
        # Their backward hands the encoder's feature gradients over in fixed buffers (feat_leaves below).
        menc = motion = None
        feat_leaves = []
        if motions:
            menc = _Segment("menc", s_mot)

            def f_menc():
                with swap():
                    model.predict_motion_feat(batch, outputs)
            menc.fwd, _ = capture(menc, f_menc, s_mot)
            self.segs.append(menc)
            motion = _Segment("motion", s_dec)
            dec_view = dict(outputs)
            for k, lst in list(outputs.items()):
                if isinstance(k, tuple) and k[0] == "motion_feats":
                    stand_ins = []
                    for t in lst:
                        if torch.is_tensor(t) and t.requires_grad:
                            leaf = t.detach().requires_grad_()
                            feat_leaves.append((t, leaf))
                            stand_ins.append(leaf)
                        else:
                            stand_ins.append(t)
                    dec_view[k] = stand_ins

            def f_motion():
                with swap():
                    model.predict_motions(batch, dec_view, feats_done=True)
            motion.fwd, _ = capture(motion, f_motion, s_dec)
            for k, v in dec_view.items():
                if k not in outputs:
                    outputs[k] = v
            self.segs.append(motion)
        self.depth, self.side, self.pose, self.menc, self.motion = depth, side, pose, menc, motion

        # which taped output belongs to which segment
        def taped(prefixes):
            found = []
            for k, v in outputs.items():
                if isinstance(k, tuple) and k[0] in prefixes and torch.is_tensor(v) and v.requires_grad:
                    if not any(v is t for t in found):
                        found.append(v)
            return found
        depth.outs = [v for k, v in outputs.items() if isinstance(k, tuple) and k[0] == "disp" and k[1] == target and v.requires_grad]
        pose.outs = taped(("cam_T_cam",))
        if motion is not None:
            motion.outs = taped(("complete_flow_field", "motion_prob", "motion_mask"))
            if not any(k[0] == "complete_flow_field" for k in outputs if isinstance(k, tuple)):
                motion.outs += taped(("complete_flow",))

        # ---- loss graph: values AND d loss / d (network outputs) ----------------------------------------------------------
        self.loss_seg = lseg = _Segment("loss", main)
        leaves = {}          # id(tensor with tape) -> detached leaf standing in for it in the loss
        loss_outputs = {}
        for k, v in outputs.items():
            if torch.is_tensor(v) and v.requires_grad:
                leaf = leaves.get(id(v))
                if leaf is None:
                    leaf = leaves[id(v)] = v.detach().requires_grad_()
                loss_outputs[k] = leaf
            else:
                loss_outputs[k] = v
        self.loss_outputs = loss_outputs
        holder = {}

        # The statistics-only passes feed nothing but the BatchNorm running buffers, which no training-mode kernel reads: their
        # fold (layers.DeferredStats, after the target-frame pass, in frame order) is a graph of its own, so that the side batch
        # -- twice the target pass's work -- need not finish before the loss; it runs on under the backward graphs and joins
        # the step in front of the optimizer (DD_SEG_SIDE_LATE=0: joined in front of the loss as before).
        self.fold_seg = None
        if self.collectors and self.side_late:
            self.fold_seg = _Segment("fold", s_side)

            def f_fold():
                for col in self.collectors:
                    col.apply()
            self.fold_seg.fwd, _ = capture(self.fold_seg, f_fold, s_side)

        def f_loss():
            if self.fold_seg is None:
                for col in self.collectors:
                    col.apply()
            # the loss path is fp32: half-precision network outputs are promoted here, inside the graph (Trainer.run_networks)
            once = {}                # one promotion per tensor: the two frames share their flow / mask tensors, and the loss checks identity

            def up(v):
                if not (torch.is_tensor(v) and v.is_floating_point() and v.dtype != torch.float32):
                    return v
                if id(v) not in once:
                    once[id(v)] = v.float()
                return once[id(v)]
            promoted = {k: up(v) for k, v in loss_outputs.items()}
            losses = tr.fused_losses(batch, promoted)
            wanted = [leaves[id(t)] for seg in self.segs for t in seg.outs]
            top = losses["loss"] if self.scaler is None else self.scaler.scale(losses["loss"])       # (x the device-side loss scale)
            grads = torch.autograd.grad(top, wanted, allow_unused=True) if wanted else ()
            holder["losses"], holder["grads"], holder["wanted"] = losses, grads, wanted
        self.tile_launch = None
        if self.time_tile_kernel:
            # bench.py's roofline leg: the loss as graph | the photometric tile kernel, launched by the host | graph.  The kernel's
            # arguments are addresses inside the loss segment's pool (fixed for good), so the same launch is valid at every replay;
            # as an ordinary launch on the stream it can be bracketed by HIP events (dd_photo_timing) like the host-issued step's.
            from hipops import fused_loss as FL
            graphs = [torch.cuda.CUDAGraph()]
            torch.cuda.synchronize()
            self._private_blas_workspace()
            with torch.cuda.stream(self.main):
                graphs[0].capture_begin(lseg.pool, capture_error_mode=self.capture_mode)

                def cut(launch):
                    assert self.tile_launch is None, "one gradient-carrying photometric launch per step"
                    self.tile_launch = launch
                    graphs[-1].capture_end()
                    graphs.append(torch.cuda.CUDAGraph())
                    graphs[-1].capture_begin(lseg.pool, capture_error_mode=self.capture_mode)
                FL.TILE_CUT = cut
                try:
                    f_loss()
                finally:
                    FL.TILE_CUT = None
                    graphs[-1].capture_end()
            assert len(graphs) == 2 and self.tile_launch is not None, len(graphs)
            self.loss_graphs = graphs
            lseg.fwd = None
        else:
            lseg.fwd, _ = capture(lseg, f_loss, main)
            self.loss_graphs = [lseg.fwd]
        self.losses = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in holder["losses"].items()}
        self._loss_grads = [(w, g) for w, g in zip(holder["wanted"], holder["grads"]) if g is not None]      # DD_SEG_CHECK
        grad_of = {id(w): g for w, g in zip(holder["wanted"], holder["grads"])}

        # ---- backward graphs, parameter gradients into one flat buffer per segment -----------------------------------------
        feat_grads = {}              # id(encoder feature) -> its gradient, left behind by the decoders' backward graph
        order = [g for g in self.segs if g.name != "menc"] + [g for g in self.segs if g.name == "menc"]      # the encoder after its decoders
        # ONE gradient buffer for the step, a contiguous slice per segment: per-segment collectives (DD_SEG_REDUCE=overlap) reduce
        # their slice, the default reduces the whole buffer with one collective behind the last backward graph
        seg_names = {"depth": ["depth_enc", "depth_dec"], "pose": ["pose_enc", "pose_dec"], "menc": ["motion_enc"],
                     "motion": ["motion_dec", "motion_mask"], "side": []}
        sizes = {seg.name: sum((p.numel() + 3) & ~3 for n in seg_names[seg.name] for p in getattr(model, n).parameters() if p.requires_grad) for seg in order}
        self.flat_all = torch.zeros(max(sum(sizes.values()), 1), dtype=torch.float32, device=tr.device)
        flat_off = 0
        for seg in order:
            if seg.name == "menc":
                pairs = [(t, feat_grads.get(id(t))) for t, _ in feat_leaves]
            else:
                pairs = [(t, grad_of.get(id(leaves[id(t)]))) for t in seg.outs]
            pairs = [(t, g) for t, g in pairs if g is not None]
            names = {"depth": ["depth_enc", "depth_dec"], "pose": ["pose_enc", "pose_dec"], "menc": ["motion_enc"],
                     "motion": ["motion_dec", "motion_mask"], "side": []}[seg.name]
            seg.params = [p for n in names for p in getattr(model, n).parameters() if p.requires_grad]
            if not pairs or not seg.params:
                seg.params = []
                continue
            total = sum((p.numel() + 3) & ~3 for p in seg.params)
            seg.flat = self.flat_all[flat_off:flat_off + total]
            flat_off += total
            views, off = [], 0
            for p in seg.params:
                # the parameter's own strides (channels-last conv weights are dense but permuted): the fused Adam kernel wants
                # gradient and parameter laid out alike; every view starts on a 16-byte boundary (dd_adam_multi's vector accesses)
                views.append(seg.flat[off:off + p.numel()].as_strided(p.size(), p.stride()))
                off += (p.numel() + 3) & ~3

            def f_bwd(seg=seg, pairs=pairs, views=views):
                leaves_ = [alias_of[id(p)] for p in seg.params]
                extra = [leaf for _, leaf in feat_leaves] if seg.name == "motion" else []
                grads = torch.autograd.grad([t for t, _ in pairs], leaves_ + extra, grad_outputs=[g for _, g in pairs], allow_unused=True)
                for (t, _), g in zip(feat_leaves if extra else [], grads[len(leaves_):]):
                    if g is not None:
                        feat_grads[id(t)] = g
                grads = grads[:len(leaves_)]
                dst = [v for v, g in zip(views, grads) if g is not None]
                src = [g for g in grads if g is not None]
                if "c" in dbg:
                    for d_, s_ in zip(dst, src):
                        d_.copy_(s_)
                elif "n" not in dbg:
                    torch._foreach_copy_(dst, src)
                else:
                    seg._keep = src
            seg.bwd, _ = capture(seg, f_bwd, seg.stream, "bwd")
            seg.views = views
            for p, v in zip(seg.params, views):
                p.grad = v
        # the tapes are spent: what callers see are plain tensors
        self.outputs = {k: ([t.detach() if torch.is_tensor(t) else t for t in v] if isinstance(v, list) else (v.detach() if torch.is_tensor(v) else v))
                        for k, v in outputs.items()}

        # ---- optimizer graph ------------------------------------------------------------------------------------------------
        self.optim_seg = _Segment("optim", main)
        self._capture_optimizer()

        bn_after = {id(m): m._pending_batches for m, _ in bn_before}
        self._bn_delta = [(m, bn_after[id(m)] - n) for m, n in bn_before if bn_after[id(m)] != n]
        for m, n in bn_before:                  # the capture itself ran nothing
            m.__dict__["_pending_batches"] = n
        torch.cuda.synchronize()

    @staticmethod
    def _private_blas_workspace():
        """PyTorch keeps ONE rocBLAS / hipBLASLt workspace per (handle, stream) and bakes its address into every GEMM it records.
        All graphs here are recorded on one stream and replayed side by side on several: the GEMMs of the depth pass, of the
        statistics-only batch and of the depth backward would share one split-K workspace (the eager step's streams have one
        each).  Dropping the cached workspaces in front of a capture makes the next GEMM allocate a fresh one inside THAT
        graph's private pool.  (A precaution: the non-finite runs of round 3 were the runtime's graph packet capture, not this.)"""
        torch._C._cuda_clearCublasWorkspaces()

    def _capture_optimizer(self):
        """The fused-Adam graph.  Adam skips parameters whose .grad is None, and an eager step in between (log steps, the epoch's
        zero_grad) leaves every .grad at None: the views into the flat buffers are put back first, or a re-capture after a
        learning-rate change would record an EMPTY graph and the replayed steps would stop updating the weights."""
        optimizer = self.tr.optim["optimizer"]
        seg = self.optim_seg
        for s_ in self.segs:
            for p, v in zip(s_.params, getattr(s_, "views", ())):
                p.grad = v
        # torch's multi-tensor Adam needs a dozen launches for the ~400 parameter tensors and moves them at ~1.4 TB/s; the update runs
        # alone at the end of the step.  hipops.adam covers plain fp32 Adam (what the reference configures) with one launch; anything
        # else -- and DD_STOCK_ADAM=1 -- stays with torch.  (fp16 networks: the loss scaler drives it, see below.)
        self.one_launch_adam = None
        self.adam_fallback = "DD_STOCK_ADAM=1"
        stock = os.environ.get("DD_STOCK_ADAM", "0")          # "1": torch's kernels everywhere; "fp16": under a loss scaler only (A/B switch)
        if stock == "fp16" and self.scaler is not None:
            self.adam_fallback = "DD_STOCK_ADAM=fp16"
        elif stock != "1":
            from hipops import adam as HA
            if type(optimizer) is torch.optim.Adam:
                HA.ensure_state(optimizer)
            self.adam_fallback = HA.unsupported_reason(optimizer)          # (bench.py reports it)
            if self.adam_fallback is None:
                self.one_launch_adam = HA.MultiTensorAdam(optimizer)
        g = torch.cuda.CUDAGraph()
        self._private_blas_workspace()
        with _empty_graph_watch() as watch, torch.cuda.graph(g, pool=seg.pool, stream=self.main, capture_error_mode=self.capture_mode):
            if self.scaler is None and self.one_launch_adam is not None:
                self.one_launch_adam.step()             # dd_adam_multi: every parameter tensor of the step in one launch
            elif self.scaler is None:
                optimizer.step()
            else:
                # GradScaler.step on an optimizer that takes grad_scale / found_inf (fused Adam): _amp_foreach_non_finite_check over
                # the gradients, then the step with both tensors -- no host read; update(): _amp_update_scale_ on the device.
                # With dd_adam_multi the scaler still does all of that; only what `optimizer.step` LAUNCHES is replaced: the scaler
                # parks its two device scalars on the optimizer object (optimizer.grad_scale / .found_inf) around the call.
                mt = self.one_launch_adam
                had, saved = "step" in optimizer.__dict__, optimizer.__dict__.get("step")       # (StepLR wraps step on the instance)
                if mt is not None and getattr(optimizer, "_step_supports_amp_scaling", False):
                    optimizer.step = lambda *a, **k: mt.step(getattr(optimizer, "grad_scale", None), getattr(optimizer, "found_inf", None))
                else:
                    self.one_launch_adam = None
                    self.adam_fallback = self.adam_fallback or "the optimizer does not take GradScaler's device scalars"
                try:
                    self.scaler.step(optimizer)
                finally:
                    if had:
                        optimizer.__dict__["step"] = saved
                    else:
                        optimizer.__dict__.pop("step", None)
                self.scaler.update()
        if watch.empty:
            raise RuntimeError("the optimizer graph recorded no launch: no parameter of the phase carries a gradient view "
                               "(replayed steps would not train)")
        seg.fwd = g
        self._lrs = [grp["lr"] for grp in optimizer.param_groups]

    # ------------------------------------------------------------------------------------------------------------------
    def flush_counters(self):
        """BatchNorm's `num_batches_tracked` bookkeeping (host side, layers.BatchNorm2d) for the replays since the last call."""
        if self.replays:
            for m, d in self._bn_delta:
                m.__dict__["_pending_batches"] += d * self.replays
        self.replays = 0

    def run(self, inputs):
        """One training step.  `inputs`: the batch on the device (after Trainer.upload_inputs).
        DD_SEG_PRIORITY=1 (experiment): the caller's role -- inputs, depth net, loss, Adam: the critical path of the step -- is
        replayed on a high-priority stream of its own, so that its kernels are dispatched ahead of the side branches'."""
        if self.hp_stream is None:
            return self._run(inputs)
        caller = torch.cuda.current_stream()
        self.hp_stream.wait_stream(caller)
        with torch.cuda.stream(self.hp_stream):
            out = self._run(inputs)
        caller.wait_stream(self.hp_stream)
        return out

    def _probe_reduce_mode(self):
        """DD_SEG_REDUCE=auto: which placement this replay uses, and -- once both have been timed -- the decision (one host sync and one
        small collective, once per captured step)."""
        k, n = self.total_replays, self.AUTO_PROBE
        if k < n:
            return "end"
        if k < 2 * n:
            return "overlap"
        torch.cuda.synchronize()
        ms = {}
        for mode, marks in self._probe_marks.items():
            vals = [a.elapsed_time(b) for a, b in marks[1:]]
            ms[mode] = sum(vals) / max(len(vals), 1)
        t = torch.tensor([ms["end"], ms["overlap"]], dtype=torch.float64, device=self.tr.device)
        if dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)            # the step is as slow as its slowest rank
        self.reduce_probe_ms = {"end": round(float(t[0]), 3), "overlap": round(float(t[1]), 3)}
        self.reduce_mode = self.reduce_mode_chosen = "overlap" if float(t[1]) < 0.99 * float(t[0]) else "end"
        self._probe_marks = None
        return self.reduce_mode

    def _run(self, inputs):
        tr = self.tr
        optimizer = tr.optim["optimizer"]
        if [grp["lr"] for grp in optimizer.param_groups] != self._lrs:
            self._capture_optimizer()           # the learning rate is a launch constant of the fused Adam kernel
        main = torch.cuda.current_stream()

        def S(seg):
            """The stream a segment is replayed on (DD_SEG_SERIAL=all | name,name: those on the caller's stream -- debugging)."""
            return main if ("all" in self.serial or seg.name in self.serial) else seg.stream
        # the batch into the static buffers: one multi-tensor copy
        dst, src = [], []
        for k, v in self.static.items():
            w = inputs[k]
            if w.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(w if w.dtype == v.dtype else w.to(v.dtype))
        if dst:
            torch._foreach_copy_(dst, src)
        # The host needs a third of a step's run time to enqueue it; unchecked it runs ahead of the GPU by as many steps as fit
        # between two host syncs (log steps).  Bound that to `run_ahead` steps: wait for the end of step k - run_ahead before
        # enqueuing step k -- free while the GPU is the bottleneck, and it bounds the kernel-argument / event memory in flight.
        if self.run_ahead > 0:
            if len(self._ends) >= self.run_ahead:
                t = time.perf_counter()
                self._ends.pop(0).synchronize()
                self.host_wait_s += time.perf_counter() - t           # (idle, not work: bench.py reports the host's enqueue time without it)
        self._events.append([])                 # this step's cross-stream events; dropped once the GPU is certainly past them
        if len(self._events) > (self.run_ahead + 2 if self.run_ahead > 0 else 256):
            self._events.pop(0)
        marks = self.marks = []
        reduce_mode = self._probe_reduce_mode() if self.reduce_mode == "auto" else self.reduce_mode
        probing = self.reduce_mode == "auto"
        if probing:
            p0 = torch.cuda.Event(enable_timing=True)
            p0.record(main)

        def replay(seg, graph, what):
            """graph.replay() on the current stream; with DD_SEG_TIMING=1 between two timing events."""
            if graph is None:             # captured empty: nothing to launch
                return
            guard = self.exec_guard
            if guard:
                # one launch of a graph exec in flight at a time: wait (host) until the GPU has finished the previous step's
                # launch of THIS graph before launching it again
                last = self._last_launch.get(id(graph))
                if last is not None:
                    t = time.perf_counter()
                    last.synchronize()
                    self.host_wait_s += time.perf_counter() - t
            cur = torch.cuda.current_stream()
            if not self.timing:
                graph.replay()
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                graph.replay()
                e1.record(cur)
                marks.append((seg.name + " " + what, e0, e1))
            if guard:
                done = torch.cuda.Event()
                done.record(cur)
                self._last_launch[id(graph)] = done

        if self.timing:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record(main)
        replay(self.inputs_seg, self.inputs_seg.fwd, "fwd")
        side, pose, menc, motion, depth = self.side, self.pose, self.menc, self.motion, self.depth
        for seg in (side, pose, menc, motion):
            if seg is not None:
                self._wait(S(seg), main)
        if menc is not None:                                 # the longest chain first: encoder -> decoders
            with torch.cuda.stream(S(menc)):
                replay(menc, menc.fwd, "fwd")
        with torch.cuda.stream(S(pose)):
            replay(pose, pose.fwd, "fwd")
        if motion is not None:
            self._wait(S(motion), S(pose))          # the decoders read the (detached) pose vectors ...
            self._wait(S(motion), S(menc))          # ... and the encoder's features
            with torch.cuda.stream(S(motion)):
                replay(motion, motion.fwd, "fwd")
        late = side is not None and self.fold_seg is not None
        side_at = self.side_at if late else "start"

        def run_side(behind):
            """the side batch and the fold of its statistics, on the side batch's stream, behind `behind`"""
            self._wait(S(side), behind)               # (the fold follows the target-frame pass's own update of the buffers)
            with torch.cuda.stream(S(side)):
                replay(side, side.fwd, "fwd")
                replay(self.fold_seg, self.fold_seg.fwd, "fwd")
        if side is not None and side_at == "start":
            with torch.cuda.stream(S(side)):
                replay(side, side.fwd, "fwd")
        replay(depth, depth.fwd, "fwd")
        if late and side_at == "start":
            self._wait(S(side), main)                   # the fold follows the target-frame pass's own update of the buffers
            with torch.cuda.stream(S(side)):
                replay(self.fold_seg, self.fold_seg.fwd, "fwd")
        for seg in (None if late else side, pose, menc, motion):
            if seg is not None:
                self._wait(main, S(seg))
        if self.loss_events is not None:             # bench.py: HIP events around the whole loss, on the stream it runs on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
        for i, g in enumerate(self.loss_graphs):
            if i == 1:
                from hipops import lib as HL
                self.tile_launch(HL.current_stream())        # the tile kernel itself (time_tile_kernel)
            replay(self.loss_seg, g, "fwd")
        if self.loss_events is not None:
            e1.record(main)
            self.loss_events.append((e0, e1))
        # backward: the longest chain first (decoders, then the encoder behind them)
        works = []
        ran = []
        if late and side_at == "loss":
            run_side(main)
        if motion is not None and motion.bwd is not None:
            self._wait(S(motion), main)
            with torch.cuda.stream(S(motion)):
                replay(motion, motion.bwd, "bwd")
                if self.ddp and reduce_mode == "overlap":
                    works.append(self._all_reduce(motion))
            ran.append(motion)
        if menc is not None and menc.bwd is not None:
            self._wait(S(menc), main)
            self._wait(S(menc), S(motion))
            with torch.cuda.stream(S(menc)):
                replay(menc, menc.bwd, "bwd")
                if self.ddp and reduce_mode == "overlap":
                    works.append(self._all_reduce(menc))
            ran.append(menc)
        if pose.bwd is not None:
            self._wait(S(pose), main)
            with torch.cuda.stream(S(pose)):
                replay(pose, pose.bwd, "bwd")
                if self.ddp and reduce_mode == "overlap":
                    works.append(self._all_reduce(pose))
            ran.append(pose)
        if late and side_at == "pose":
            run_side(S(pose))
        if depth.bwd is not None:
            replay(depth, depth.bwd, "bwd")
            if self.ddp and reduce_mode == "overlap":
                works.append(self._all_reduce(depth))
        for seg in ran + ([side] if late else []):
            self._wait(main, S(seg))
        if self.ddp and reduce_mode not in ("overlap", "none"):        # ("none": experiments with a one-rank group only)
            works.append(self._all_reduce(None))          # every backward graph has been joined: the whole buffer, one collective
        for w in works:
            if w is not None:
                w.wait()                         # orders the collective before the optimizer on the current stream
        if self.check or self.replays == self.check_at:
            self._check_finite("before the optimizer")
        if self.probe:
            self._probe()
        replay(self.optim_seg, self.optim_seg.fwd, "optim")
        if probing:
            p1 = torch.cuda.Event(enable_timing=True)
            p1.record(main)
            self._probe_marks[reduce_mode].append((p0, p1))
        if self.run_ahead > 0:
            end = torch.cuda.Event()
            end.record(main)
            self._ends.append(end)
        self.replays += 1
        self.total_replays += 1
        if self.check:
            self._check_finite("after the optimizer", params=True)
        return self.outputs, self.losses

    def _probe(self):
        """DD_SEG_PROBE=1 (debugging, no host sync): per step, in front of the optimizer, one finiteness flag per buffer of the
        replayed step into a ring on the device; probe_report() reads the ring when something has gone wrong."""
        items = [("loss", self.losses["loss"])]
        for k, v in self.outputs.items():
            for i, t in enumerate(v if isinstance(v, list) else [v]):
                if torch.is_tensor(t) and t.is_floating_point():
                    items.append(("output {}{}".format(k, "[%d]" % i if isinstance(v, list) else ""), t))
        items += [("d loss / d output {}".format(tuple(g.shape)), g) for _, g in self._loss_grads]
        items += [("gradients of " + seg.name, seg.flat) for seg in self.segs if seg.flat is not None]
        items += [("weights of " + n, torch.cat([p.detach().reshape(-1)[:1] for p in getattr(self.model, n).parameters()])) for n in self.model.module_names]
        if self._ring is None:
            self._ring_names = [n for n, _ in items]
            self._ring = torch.ones(1024, len(items), dtype=torch.bool, device=self.tr.device)
            self._ring_step = torch.full((1024,), -1, dtype=torch.int64, device=self.tr.device)
        row = self.replays % 1024
        self._ring[row] = torch.stack([torch.isfinite(t).all() for _, t in items])
        self._ring_step[row] = self.replays

    def probe_report(self):
        if self._ring is None:
            return ""
        ring, steps = self._ring.cpu(), self._ring_step.cpu()
        rows = sorted((int(steps[i]), i) for i in range(len(steps)) if int(steps[i]) >= 0)
        for step, i in rows:
            bad = [self._ring_names[j] for j in range(ring.shape[1]) if not bool(ring[i, j])]
            if bad:
                return "first replay with a non-finite buffer in front of the optimizer: {} -> {}".format(step, bad[:20])
        return "no non-finite buffer in the probed replays ({}..{})".format(rows[0][0], rows[-1][0]) if rows else ""

    def _check_finite(self, when, params=False):
        """DD_SEG_CHECK=1 (debugging; one device sync per call): the first non-finite buffer of the replayed step, in data-flow order."""
        torch.cuda.synchronize()

        def bad(t):
            return torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())
        found = []
        for k, v in self.static.items():
            if bad(v):
                found.append("input {}".format(k))
        for k, v in self.outputs.items():
            for i, t in enumerate(v if isinstance(v, list) else [v]):
                if bad(t):
                    found.append("network output {}{}".format(k, "[%d]" % i if isinstance(v, list) else ""))
        for k, v in self.losses.items():
            if bad(v):
                found.append("loss {}".format(k))
        for w, g in self._loss_grads:
            if bad(g):
                found.append("d loss / d output of shape {}".format(tuple(g.shape)))
        for seg in self.segs:
            if seg.flat is not None and bad(seg.flat):
                names = [n for n, p in self.model.named_parameters() if p.grad is not None and bad(p.grad)]
                found.append("gradients of segment {} ({} tensors, first {})".format(seg.name, len(names), names[:2]))
        if params:
            names = [n for n, p in self.model.named_parameters() if bad(p)]
            if names:
                found.append("{} parameters, first {}".format(len(names), names[:3]))
            names = [n for n, b in self.model.named_buffers() if bad(b)]
            if names:
                found.append("{} buffers, first {}".format(len(names), names[:3]))
        if not params:
            self._trail = (getattr(self, "_trail", []) + [(self.replays, float(self.losses["loss"]),
                                                           max(float(seg.flat.abs().max()) for seg in self.segs if seg.flat is not None))])[-12:]
        if found:
            raise FloatingPointError("replay {} {}: non-finite {}\n(replay, loss, largest |gradient|) of the last steps: {}".format(
                self.replays, when, "; ".join(found[:12]), ["%d %.4g %.3g" % t for t in getattr(self, "_trail", [])]))

    def timeline(self):
        """[(segment, start ms, end ms)] of the last run() relative to its start (DD_SEG_TIMING=1; synchronises)."""
        torch.cuda.synchronize()
        return [(name, self.t0.elapsed_time(e0), self.t0.elapsed_time(e1)) for name, e0, e1 in self.marks]

    def _wait(self, waiter, waited):
        """waiter.wait_stream(waited) with an event that stays alive until the GPU has passed it.  Stream.wait_stream() creates an
        event, enqueues the wait and drops the event at once; with the host one or two replayed steps ahead of the GPU the
        runtime is then asked to destroy an event that a stream has yet to wait on.  Legal, and not what broke the long runs of
        round 3 -- but one assumption less about the runtime, for a list of events per step."""
        ev = torch.cuda.Event()
        ev.record(waited)
        waiter.wait_event(ev)
        self._events[-1].append(ev)

    def _all_reduce(self, seg):
        """Average of a gradient buffer over the ranks: the segment's slice, issued behind its backward graph on its stream
        (DD_SEG_REDUCE=overlap), or -- seg None -- the whole buffer behind the last backward graph (the collective runs on RCCL's
        stream; the caller waits for it in front of the optimizer graph)."""
        buf = self.flat_all if seg is None else seg.flat
        if dist.get_backend() == "gloo":             # CPU collectives on GPU tensors (the two-ranks-on-one-device tests): blocking
            buf.div_(self.world)
            dist.all_reduce(buf)
            return None
        return dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True)
