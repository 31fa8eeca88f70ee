"""Container of the seven Dynamo-Depth sub-networks; same surface as the reference's networks.Model
(networks/model.py:15-230): forward(inputs) -> outputs dict, phase flags bool_CmpFlow / bool_MotMask,
parameters_by_names, per-module save / load with the reference's file layout, set_train / set_eval.

Depth D (3 frames), pose P (2 ordered pairs), complete flow C and motion mask M (one 3-frame stack, shared
encoder).  The conv GEMMs are MIOpen / hipBLASLt through PyTorch-ROCm; what this tree adds natively sits
behind the outputs (hipops.fused_loss) and in the pose-vector -> matrix kernel.
"""
import os
import os.path as osp

import torch
import torch.nn as nn

from .layers import transformation_from_parameters
from .resnet_encoder import ResnetEncoder
from .depth_encoder import LiteMono
from .depth_decoder import DepthDecoder, LiteDepthDecoder
from .pose_decoder import PoseDecoder
from .motion_decoder import MotionDecoder

# network name -> sub-modules (reference networks/model.py:38-43); C and M share the motion encoder
NETWORK_MODULES = {
    "Depth": ["depth_enc", "depth_dec"],
    "Pose": ["pose_enc", "pose_dec"],
    "CmpFlow": ["motion_enc", "motion_dec"],
    "MotMask": ["motion_enc", "motion_mask"],
}

MODEL_ZOO = ("ckpt/K_Dynamo-Depth_MD2", "ckpt/K_Dynamo-Depth", "ckpt/N_Dynamo-Depth_MD2", "ckpt/N_Dynamo-Depth",
             "ckpt/W_Dynamo-Depth_MD2", "ckpt/W_Dynamo-Depth")


class Model(nn.Module):
    def __init__(self, options):
        super().__init__()
        self.opt = opt = options
        pre = opt.weights_init == "pretrained"
        if opt.depth_model == "monodepthv2":
            self.depth_enc = ResnetEncoder(opt.encoder_num_layers, pre)
            self.depth_dec = DepthDecoder(self.depth_enc.num_ch_enc, opt.scales)
        elif opt.depth_model == "litemono":
            self.depth_enc = LiteMono(model="lite-mono-8m", drop_path_rate=0.4, pretrained=pre)
            self.depth_dec = LiteDepthDecoder(self.depth_enc.num_ch_enc, opt.scales)
        else:
            raise Exception("Model Name {} not recognized.".format(opt.depth_model))
        self.pose_enc = ResnetEncoder(opt.encoder_num_layers, pre, num_input_images=2, inp_disp=False)
        self.pose_dec = PoseDecoder(self.pose_enc.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
        self.motion_enc = ResnetEncoder(opt.encoder_num_layers, pre, num_input_images=3, inp_disp=False)
        self.motion_dec = MotionDecoder(self.pose_enc.num_ch_enc, opt.scales, num_input_images=3, inp_disp=False, out_dim=3)
        self.motion_mask = MotionDecoder(self.pose_enc.num_ch_enc, opt.scales, num_input_images=3, inp_disp=False, out_dim=1)
        # every network packs the weights of its dd_conv3x3_mfma layers in one launch at the top of its forward pass (hipops.functions.PackSet)
        from hipops.functions import pack_weights_once_per_forward
        for net in (self.depth_enc, self.depth_dec, self.pose_enc, self.pose_dec, self.motion_enc, self.motion_dec, self.motion_mask):
            pack_weights_once_per_forward(net)
        self.network2modules = {k: list(v) for k, v in NETWORK_MODULES.items()}
        self.module_names = list(set(m for mods in self.network2modules.values() for m in mods))
        self.bool_CmpFlow = True
        self.bool_MotMask = True
        self.model_zoo = {k: None for k in MODEL_ZOO}

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, inputs):
        outputs = {}
        if getattr(self.opt, "multi_stream", False) and inputs["color_aug", 0, 0].is_cuda:
            # Eager execution runs the branches side by side.  Under hipGraph capture the plain single-stream forward is
            # recorded unless DD_MS_CAPTURE names a placement: parallel graph branches are faster (KITTI shape, B=12: 216-222
            # img/s against 192 for the single-stream graph) but not dependable on this ROCm stack -- with motion encoder and
            # decoders both on one side stream the replayed step turns non-finite within two updates, a fifth stream for
            # the decoders crashes the capture inside the HIP runtime, and the placement that replays correctly at 192x640
            # ("e": encoder on the capturing stream) hangs in replay at the Waymo shape 320x480 (scripts/debug_graph_ms.py).
            if not torch.cuda.is_current_stream_capturing() or os.environ.get("DD_MS_CAPTURE"):
                return self.forward_streams(inputs, outputs)
        self.predict_depths(inputs, outputs)
        self.predict_poses(inputs, outputs)
        self.predict_motions(inputs, outputs)
        return outputs

    def side_streams(self):
        """The HIP streams of the multi-stream forward (created on first use, on the current device)."""
        if getattr(self, "_streams", None) is None:
            # [statistics-only batch, (spare), pose, motion]: the three that carry work sit on hardware queues of their own,
            # measured (hipops.queues) -- HIP folds its streams onto four queues and two branches on one queue run one after
            # the other (round 3: the pose branch behind the motion encoder, ~4 ms of the step)
            from hipops import queues
            s_prev, s_pose, s_mot, s_next = queues.pick(4)
            self._streams = [s_prev, s_next, s_pose, s_mot]
        return self._streams

    def forward_streams(self, inputs, outputs):
        """The same forward with its independent branches on separate HIP streams: depth net on the target frame (current
        stream), the statistics-only depth passes (one batch), the pose passes (one batch), the motion networks.  Many kernels of these networks
        launch fewer workgroups than the chip has CUs (LiteMono's 1/16-resolution stage: ~160 for a convolution); side by side
        they fill it (eager, KITTI shape, B=12: 254 against 216 img/s).  Autograd runs every backward node on its forward
        stream, so the backward is spread the same way."""
        import torch.cuda as tc
        cur = tc.current_stream()
        self._main_stream = cur
        s_prev, s_next, s_pose, s_mot = self.side_streams()
        dbg = os.environ.get("DD_MS_DEBUG", "")            # debugging: letters d / p / m keep that branch on the current stream
        if tc.is_current_stream_capturing():
            dbg += os.environ.get("DD_MS_CAPTURE", "")      # opt-in placements under capture (see forward)
        if "d" in dbg:
            s_prev = s_next = cur
        if "p" in dbg:
            s_pose = cur
        if "m" in dbg:
            s_mot = cur
        frames = list(self.opt.frame_ids)
        side = {}
        if not (getattr(self.opt, "skip_unused_depth_frames", False) and self.training):
            side = dict(zip(frames[1:], (s_prev, s_next)))
        # stochastic-depth factors of the depth passes, drawn in the frame order of the single-stream forward
        predrawn = {}
        if self.training and hasattr(self.depth_enc, "draw_drop_masks") and os.environ.get("DD_STOCK_DROP_PATH", "0") != "1":
            for f in frames[:1] + list(side):
                predrawn[f] = self.depth_enc.draw_drop_masks(inputs["color_aug", f, 0], install=False)
        for st in self._streams:
            st.wait_stream(cur)
        # the statistics-only passes run beside the target-frame pass; their BatchNorm running-statistics updates are kept
        # aside and folded in afterwards in the reference's order (frame 0, -1, +1) -- see layers.DeferredStats
        from networks.layers import BatchNorm2d, DeferredStats
        if getattr(self, "_bn_floats", None) is None:
            self._bn_floats = sum(2 * m.num_features for mod in (self.depth_enc, self.depth_dec) for m in mod.modules() if isinstance(m, BatchNorm2d))
        deferred = []
        if side:
            # both statistics-only frames go through the net as one batch on one side stream (predict_depths)
            with tc.stream(s_prev):
                for f in side:
                    if predrawn.get(f) is not None:
                        predrawn[f].record_stream(s_prev)
                cols = None
                if self.training:
                    cols = [DeferredStats(inputs["color_aug", f, 0].device, max(self._bn_floats, 1)) for f in side]
                    deferred = [(s_prev, col) for col in cols]
                self._predrawn_masks = predrawn
                try:
                    self.predict_depths(inputs, outputs, frames=list(side), collectors=cols)
                finally:
                    self._predrawn_masks = None
        with tc.stream(s_pose):
            self.predict_poses(inputs, outputs)
        motions = self.bool_CmpFlow or self.bool_MotMask
        s_enc = cur if "e" in dbg else s_mot             # motion encoder and motion decoders are placed separately, see below
        s_dec = cur if "c" in dbg else s_mot
        if motions:
            with tc.stream(s_enc):
                self.predict_motion_feat(inputs, outputs)
        if predrawn.get(frames[0]) is not None:
            self.depth_enc.install_drop_masks(predrawn[frames[0]])
        self.predict_depths(inputs, outputs, frames=frames[:1])
        if motions:
            s_dec.wait_stream(s_pose)                    # the decoders read the (detached) pose vectors
            if s_dec is not s_enc:
                s_dec.wait_stream(s_enc)
            with tc.stream(s_dec):
                self.predict_motions(inputs, outputs, feats_done=True)
        for st in self._streams:
            cur.wait_stream(st)
        for _, col in deferred:              # after the join, on the current stream: frame -1's update, then frame +1's
            col.apply()
        for v in outputs.values():
            for t in (v if isinstance(v, list) else [v]):
                if torch.is_tensor(t):
                    t.record_stream(cur)
        return outputs

    def predict_depths(self, inputs, outputs, frames=None, collectors=None):
        # all frames go through the depth net although only frame 0 feeds the loss: the extra passes update
        # the BatchNorm running statistics exactly as the reference does (networks/model.py:69-74).  Nothing
        # differentiates through them, so they run without an autograd tape (same arithmetic, same random draws;
        # no activations kept for a backward that never comes) -- and, in training mode, as ONE batch of 2B samples whose
        # BatchNorm layers keep the two frames apart (layers.batch_groups): half the launches for the same statistics.
        # `collectors`: one DeferredStats per tape-free frame (the multi-stream forward), None = update the buffers in order.
        if frames is None:
            frames = self.opt.frame_ids
            if getattr(self.opt, "skip_unused_depth_frames", False) and self.training:
                frames = frames[:1]           # opt-in: changes the BatchNorm running statistics w.r.t. the reference
        frames = list(frames)
        target = self.opt.frame_ids[0]
        taped = [f for f in frames if f == target and torch.is_grad_enabled()]
        free = [f for f in frames if f not in taped]
        for f in taped:
            for (name, s), v in self.depth_dec(self.depth_enc(inputs["color_aug", f, 0])).items():
                outputs[(name, f, s)] = v
        if not free:
            return
        batched = (self.training and len(free) > 1 and os.environ.get("DD_STOCK_SIDE_PASSES", "0") != "1"
                   and inputs["color_aug", free[0], 0].is_cuda)
        # opt-in (--stats_only_side_frames): a statistics-only pass stops after the encoder -- the decoders hold no BatchNorm
        # and nothing in a training step reads the disparities of frames -1/+1 (reference Trainer.py:222-230,357,371,428)
        decode = self.depth_dec if not (self.training and getattr(self.opt, "stats_only_side_frames", False)) else (lambda feats: {})
        with torch.no_grad():
            if not batched:
                from networks.layers import defer_running_stats
                for i, f in enumerate(free):
                    if collectors is not None:
                        with defer_running_stats(collectors[i]):
                            out = decode(self.depth_enc(inputs["color_aug", f, 0]))
                    else:
                        out = decode(self.depth_enc(inputs["color_aug", f, 0]))
                    for (name, s), v in out.items():
                        outputs[(name, f, s)] = v
                return
            from networks.layers import batch_groups
            per = inputs["color_aug", free[0], 0].shape[0]
            masks = getattr(self, "_predrawn_masks", None)
            if hasattr(self.depth_enc, "draw_drop_masks") and os.environ.get("DD_STOCK_DROP_PATH", "0") != "1":
                # per-frame stochastic-depth factors in frame order (drawn here unless the caller drew them up front)
                rows = [masks[f] if masks and masks.get(f) is not None else self.depth_enc.draw_drop_masks(inputs["color_aug", f, 0], install=False)
                        for f in free]
                if all(r is not None for r in rows):
                    self.depth_enc.install_drop_masks(torch.cat(rows, 1))
            x = torch.cat([inputs["color_aug", f, 0] for f in free])
            with batch_groups(len(free), collectors):
                out = decode(self.depth_enc(x))
            for (name, s), v in out.items():
                for i, f in enumerate(free):
                    outputs[(name, f, s)] = v[i * per:(i + 1) * per]

    def predict_poses(self, inputs, outputs, frames=None):
        frames = list(self.opt.frame_ids[1:] if frames is None else frames)
        pairs = [torch.cat([inputs["color_aug", f, 0], inputs["color_aug", 0, 0]], 1) for f in frames]   # target frame last
        if (self.training and len(frames) > 1 and pairs[0].is_cuda and os.environ.get("DD_STOCK_POSE_PASSES", "0") != "1"):
            # both pose passes as one batch of 2B through the shared networks; BatchNorm keeps the two apart and updates its
            # running statistics pass by pass (layers.batch_groups) -- the same numbers from half the launches
            from networks.layers import batch_groups
            per = pairs[0].shape[0]
            with batch_groups(len(frames)):
                feats = self.pose_enc(torch.cat(pairs))
                axisangle, translation = self.pose_dec([feats])
            for i, f in enumerate(frames):
                sl = slice(i * per, (i + 1) * per)
                self._publish_pose(outputs, f, pairs[i], [t[sl] for t in feats], axisangle[sl, 0], translation[sl, 0])
            return
        for f, pair in zip(frames, pairs):
            feats = self.pose_enc(pair)
            axisangle, translation = self.pose_dec([feats])
            self._publish_pose(outputs, f, pair, feats, axisangle[:, 0], translation[:, 0])

    def _publish_pose(self, outputs, f, pair, feats, axisangle, translation):
        outputs[("pose_feats", 0, f)] = [pair] + list(feats)
        outputs[("axisangle", 0, f)] = axisangle
        outputs[("translation", 0, f)] = translation
        outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(axisangle, translation, invert=True)

    def predict_motion_feat(self, inputs, outputs):
        for gap in set(abs(f) for f in self.opt.frame_ids[1:]):
            stack = torch.cat([inputs["color_aug", -gap, 0], inputs["color_aug", 0, 0], inputs["color_aug", gap, 0]], 1)
            outputs[("motion_feats", 0, gap)] = [stack] + self.motion_enc(stack)

    def predict_motions(self, inputs, outputs, feats_done=False):
        if not (self.bool_CmpFlow or self.bool_MotMask):
            return
        if not feats_done:
            self.predict_motion_feat(inputs, outputs)
        for gap in set(abs(f) for f in self.opt.frame_ids[1:]):
            prev, nxt = -gap, gap
            feats = outputs[("motion_feats", 0, gap)]
            ego_t = (outputs[("translation", 0, prev)].detach() - outputs[("translation", 0, nxt)].detach()) / 2
            ego_a = (outputs[("axisangle", 0, prev)].detach() - outputs[("axisangle", 0, nxt)].detach()) / 2
            ego = torch.cat((ego_t, ego_a), -1).permute(0, 2, 1).unsqueeze(3)           # (B,6,1,1)
            if self.bool_CmpFlow:
                for (name, s), v in self.motion_dec(feats, ego).items():
                    outputs[(name, prev, s)] = -1 * v          # the field points forward in time
                    outputs[(name, nxt, s)] = 1 * v
                    # the un-negated field itself (extra key, not in the reference): lets the fused loss read ONE
                    # tensor for both frames (sign folded into the time step) instead of two copies
                    outputs[(name + "_field", gap, s)] = v
            if self.bool_MotMask:
                for (name, s), v in self.motion_mask(feats, ego).items():
                    outputs[(name, prev, s)] = v               # shared by both frames (same tensor object)
                    outputs[(name, nxt, s)] = v

    # ---- helpers -----------------------------------------------------------------------------------
    def parameters_by_names(self, network_names):
        # sorted: the optimizer state in adam.pth is keyed by position in this list, and a set of strings iterates in a
        # different order in every process (hash randomisation) -- the reference's list(set(...)) makes adam.pth unloadable
        mods = sorted(set(m for n in network_names for m in self.network2modules[n]))
        params = []
        for m in mods:
            params += list(getattr(self, m).parameters())
        return params

    def modules_by_names(self, network_names):
        return sorted(set(m for n in network_names for m in self.network2modules[n]))

    def save(self, save_folder):
        """One <module>.pth per sub-module; encoders also record the training resolution (networks/model.py:163-172)."""
        for name in self.module_names:
            state = getattr(self, name).state_dict()
            if "enc" in name:
                state["height"], state["width"] = self.opt.height, self.opt.width
            torch.save(state, osp.join(save_folder, "{}.pth".format(name)))

    def load(self, dev="cpu", verbose=True):
        """Tolerant per-module load (missing files skipped, mismatching keys ignored) -- networks/model.py:174-208."""
        self.opt.load_ckpt = osp.expanduser(self.opt.load_ckpt)
        self.check_load_ckpt(self.opt.load_ckpt)
        if verbose:
            print("loading model from folder {}".format(self.opt.load_ckpt))
        for name in self.module_names:
            path = osp.join(self.opt.load_ckpt, "{}.pth".format(name))
            if not osp.exists(path):
                if verbose:
                    print("|- Loading {} weights... FAILED :: Path {} not found".format(name, path))
                continue
            ckpt = torch.load(path, map_location=dev)
            if "height" in ckpt:
                if verbose and (ckpt["height"], ckpt["width"]) != (self.opt.height, self.opt.width):
                    print("|- === WARNING: self.opt ({},{}) != loaded ({},{})".format(self.opt.height, self.opt.width, ckpt["height"], ckpt["width"]))
                ckpt.pop("height")
                ckpt.pop("width")
            module = getattr(self, name)
            if verbose:
                print("|- Loading {} weights...".format(name))
            try:
                module.load_state_dict(ckpt)
            except Exception:
                if verbose:
                    print("|- Loading {} weights... FAILED :: load_state_dict() mismatch - Loading Matched Parameters.".format(name))
                own = module.state_dict()
                own.update({k: v for k, v in ckpt.items() if k in own and own[k].shape == v.shape})
                module.load_state_dict(own)

    def check_load_ckpt(self, load_ckpt):
        if osp.isdir(load_ckpt):
            return
        if load_ckpt in self.model_zoo:
            raise Exception("Checkpoint {} is not present locally and this build has no network access to download it".format(load_ckpt))
        raise Exception("Cannot find folder {}".format(load_ckpt))

    def set_train(self):
        # the container's own flag follows (the batched statistics-only / pose passes are gated on it)
        self.training = True
        for name in self.module_names:
            getattr(self, name).train()

    def set_eval(self):
        self.training = False
        for name in self.module_names:
            getattr(self, name).eval()
