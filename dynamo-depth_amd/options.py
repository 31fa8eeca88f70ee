"""Command-line surface of Dynamo-Depth, flag-for-flag compatible with the reference's options.py
(options.py:4-303): same names, short aliases, types, defaults and dataset-dependent fall-backs, so
existing launch lines and `opt.json` files keep working.  Built from a table instead of ~50 add_argument
blocks.  `Trainer.compute_losses` discovers the loss terms from the `g_*` attributes in declaration
order (reference Trainer.py:299), so that order is part of the contract.

Additions: --fused_loss / --no_fused_loss, --synthetic, --amp, --matmul_precision, --skip_unused_depth_frames, --stats_only_side_frames, --dist_backend, --resume,
--no_device_preprocess, --no_device_decode, --no_prefetch.  The fast configuration is the default on a GPU and every part of it has an off switch:
--nchw (channels-last networks), --single_stream (multi-stream forward), --no_miopen_find (MIOpen Find), --no_hip_graph
(per-network hipGraphs); Trainer resolves the `None` defaults by device (all off on a CPU).
"""
import argparse

# (flags, kwargs) in the reference's order
_SPEC = [
    # EXPERIMENT
    (("--model_name", "-n"), dict(type=str, default="--", help="the name of the folder to save the model in")),
    (("--log_dir",), dict(type=str, default="./logs", help="log directory")),
    (("--eval_dir",), dict(type=str, default="./outputs", help="evalutation directory")),
    # SYSTEM
    (("--cuda_ids",), dict(nargs="+", type=int, default=[0], help="device ids to use - ddp if len() > 1, cuda_ids[0] for eval/vis")),
    (("--local_rank", "--local-rank"), dict(type=int, default=0, help="local rank of the trainer")),
    (("--ddp",), dict(type=bool, default=False, help="boolean: always false, only set True by train.py")),
    (("--num_workers",), dict(type=int, default=2, help="number of dataloader workers")),
    # DATASET
    (("--dataset", "-d"), dict(type=str, default="waymo", choices=["kitti", "waymo", "nuscenes"], help="dataset to train on")),
    (("--data_path",), dict(type=str, default=None, help="path to the training data")),
    (("--split",), dict(type=str, default=None, help="which train/val split to use")),
    (("--height",), dict(type=int, default=None, help="input image height")),
    (("--width",), dict(type=int, default=None, help="input image width")),
    (("--img_ext",), dict(type=str, default=".jpg", choices=[".png", ".jpg"], help="extension of images to be loaded")),
    (("--cam_name",), dict(type=str, default=None, help="which camera to use")),
    # LOSS weights -- `g_<term>`; order defines the loss-term order
    (("--g_p_photo",), dict(type=float, default=1.0, help="photometric (SSIM+L1) weight")),
    (("--g_d_smooth",), dict(type=float, default=1e-3, help="disparity smoothness weight")),
    (("--g_d_ground",), dict(type=float, default=0.1, help="disparity above-ground weight")),
    (("--g_c_smooth",), dict(type=float, default=1e-3, help="complete 3d flow smoothness weight")),
    (("--g_c_consistency",), dict(type=float, default=5.0, help="complete/ego flow consistency at static regions weight")),
    (("--g_m_sparsity",), dict(type=float, default=0.04, help="motion mask sparsity weight")),
    (("--g_m_smooth",), dict(type=float, default=0.1, help="motion mask smoothness weight")),
    (("--weight_ramp",), dict(nargs="+", type=str, default=["g_c_smooth", "g_c_consistency", "g_m_sparsity", "g_m_smooth"],
                              help="loss coefficients that require a weight ramp")),
    (("--ramp_red",), dict(type=float, default=3, help="factor by which the weight ramp is shortened")),
    (("--ssim_weight",), dict(type=float, default=0.85, help="SSIM share of the photometric loss")),
    (("--mask_disp_thrd",), dict(type=float, default=0.03, help="disparity threshold below which the consistency loss is ignored")),
    # TRAINING hyper-parameters
    (("--epoch_schedules",), dict(nargs="+", type=int, default=[1, 1, 5, 20], help="[disp_init, motion_init, mask_init, fine_tune] epochs")),
    (("--epoch-size",), dict(type=int, default=8000, help="manual epoch size (will match dataset size if 0)")),
    (("--batch_size", "-b"), dict(type=int, default=3, help="batch size")),
    (("--learning_rate",), dict(type=float, default=1e-4, help="learning rate")),
    (("--scheduler_step_size",), dict(type=int, default=10, help="step size of the scheduler")),
    # MODEL
    (("--depth_model",), dict(type=str, default="litemono", choices=["monodepthv2", "litemono"], help="depth model to use")),
    (("--encoder_num_layers",), dict(type=int, default=18, choices=[18, 34, 50, 101, 152], help="number of resnet layers")),
    (("--weights_init",), dict(type=str, default="pretrained", choices=["pretrained", "scratch"], help="pretrained or scratch")),
    (("--scales",), dict(nargs="+", type=int, default=None, help="scales of reconstruction used in the loss")),
    # TRAINING options
    (("--frame_ids",), dict(nargs="+", type=int, default=[0, -1, 1], help="frames to load")),
    (("--min_depth",), dict(type=float, default=0.1, help="minimum depth")),
    (("--max_depth",), dict(type=float, default=100.0, help="maximum depth")),
    (("--train_img_type",), dict(type=str, default=None, choices=["original", "downsample"], help="type of images to be loaded")),
    # ground plane RANSAC
    (("--gp_prior",), dict(type=float, default=0.4, help="ground prior (bottom fraction of the image)")),
    (("--gp_tol",), dict(type=float, default=0.005, help="RANSAC tolerance")),
    (("--gp_max_it",), dict(type=int, default=100, help="RANSAC iterations")),
    (("--gp_np_per_it",), dict(type=int, default=5, help="points per RANSAC iteration")),
    # LOADING / LOGGING
    (("--load_ckpt", "-l"), dict(type=str, default="", help="name of model to load")),
    (("--log_frequency",), dict(type=int, default=100, help="number of batches between each log")),
    (("--no_train_vis",), dict(action="store_true", help="if set, train without image visualisation")),
    (("--save_frequency",), dict(type=int, default=1, help="number of epochs between each save")),
    (("--comment", "-c"), dict(type=str, default="", help="additional comment wrt experiment")),
    (("--print_opt",), dict(type=bool, default=True, help="boolean: print the list of opt in command line")),
    # EVAL
    (("--eval_min_depth",), dict(type=float, default=1e-3, help="minimum depth used for depth evaluation")),
    (("--eval_max_depth",), dict(type=float, default=None, help="maximum depth used for depth evaluation")),
    (("--eval_img_bound",), dict(nargs="+", type=int, default=None, help="top, bottom, left, right image bound fractions")),
    (("--eval_img_ext",), dict(type=str, default=None, choices=[".png", ".jpg"], help="extension of evaluation images")),
    (("--eval_img_type",), dict(type=str, default=None, choices=["original", "downsample"], help="type of evaluation images")),
]

# MI355X build additions
_EXTRA = [
    (("--fused_loss",), dict(dest="fused_loss", action="store_true", default=True, help="single-pass fused HIP loss (default)")),
    (("--no_fused_loss",), dict(dest="fused_loss", action="store_false", help="operator-by-operator loss path (tools.py modules)")),
    (("--hip_graph",), dict(dest="hip_graph", action="store_true", default=None,
                            help="replay the step from hipGraphs (one forward / backward graph per sub-network on its own stream, segments.py) once the "
                                 "loss weights are constant; default: on for a GPU run"),),
    (("--no_hip_graph",), dict(dest="hip_graph", action="store_false", help="issue every step eagerly")),
    (("--synthetic",), dict(action="store_true", help="train on synthetic triplets of the configured shape (no dataset on disk needed)")),
    (("--amp",), dict(type=str, default="none", choices=["none", "bf16", "fp16"], help="autocast dtype for the networks (the loss stays fp32)")),
    (("--matmul_precision",), dict(type=str, default=None, choices=["highest", "high", "medium"],
                                   help="torch.set_float32_matmul_precision for the run: 'highest' (PyTorch's default, and this option's when absent) "
                                        "keeps the fp32 3x3 convolutions of csrc/dd_conv_mfma.hip at fp32 accuracy (six bf16 partial products); 'high' = "
                                        "bf16x3 (three partial products, 2^-16 per product); 'medium' = bf16 operands, fp32 accumulation")),
    (("--channels_last",), dict(dest="channels_last", action="store_true", default=None,
                                help="NHWC memory format for the conv nets (default on a GPU: MIOpen's fp32 implicit-GEMM kernels are NHWC)")),
    (("--nchw",), dict(dest="channels_last", action="store_false", help="keep the networks in PyTorch's default NCHW layout")),
    (("--no_miopen_find",), dict(dest="miopen_find", action="store_false", default=None,
                                 help="do not let MIOpen Find time the solvers of each convolution (torch.backends.cudnn.benchmark).  Default on a GPU: Find "
                                      "on, EXCEPT for the workloads whose problems have shipped find-db records (miopen_db/recorded.json): immediate mode "
                                      "then returns the recorded solvers -- same speed, ~50 s less warm-up")),
    (("--miopen_find",), dict(dest="miopen_find", action="store_true", help="MIOpen Find on whatever the shipped records cover")),
    (("--skip_unused_depth_frames",), dict(action="store_true", help="run the depth net on frame 0 only (changes BatchNorm statistics; off = reference behaviour)")),
    (("--stats_only_side_frames",), dict(action="store_true", help="frames -1/+1 go through the depth ENCODER only: their disparities are never read by a "
                                                                    "training step and the decoders hold no BatchNorm, so every weight, statistic and loss is "
                                                                    "unchanged; outputs[('disp', +-1, s)] are not produced (off = the reference's work)")),
    (("--keep_going_on_nan",), dict(action="store_true", help="do not stop when a logged training loss is non-finite (the reference keeps going; default here: raise with the loss terms)")),
    (("--loader_start",), dict(type=str, default=None, choices=["fork", "forkserver", "spawn"],
                               help="start method of the DataLoader workers (default: forkserver on a GPU -- forking a process that maps a GPU is slow --, fork on a CPU)")),
    (("--fresh_loader_per_epoch",), dict(action="store_true", help="build a new dataset over the epoch's file subset and a new DataLoader every epoch like the "
                                                                     "reference (default: one persistent dataset + loader, the epoch's subset is a sampler)")),
    (("--dist_backend",), dict(type=str, default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL on ROCm)")),
    (("--no_device_preprocess",), dict(dest="device_preprocess", action="store_false", default=True,
                                       help="prepare the samples (ToTensor, flip, ColorJitter) in the DataLoader workers like the reference instead of on the GPU")),
    (("--no_device_decode",), dict(dest="device_decode", action="store_false", default=True,
                                   help="decode the JPEG frames with PIL in the DataLoader workers like the reference instead of on the GPU")),
    (("--no_device_resize",), dict(dest="device_resize", action="store_false", default=True,
                                   help="frames that are not stored at the training resolution are decoded and resized with PIL in the DataLoader workers like the reference")),
    (("--no_prefetch",), dict(dest="prefetch", action="store_false", default=True, help="no double-buffered upload of the next batch")),
    (("--multi_stream",), dict(dest="multi_stream", action="store_true", default=None,
                               help="run the independent network branches of a forward on separate HIP streams (default on a GPU)")),
    (("--single_stream",), dict(dest="multi_stream", action="store_false", help="issue the whole forward on one stream")),
    (("--resume",), dict(type=str, default="", help="checkpoint folder (<log_dir>/<model_name>/models/<phase>_<epoch>) to continue from: "
                                                  "weights, optimizer, scheduler, phase / epoch / step counters and random-number streams")),
]

# dataset-dependent defaults for flags left at None (reference options.py:274-286)
_DATASET_DEFAULTS = {
    "split": {"waymo": "waymo", "nuscenes": "nuscenes", "kitti": "eigen_zhou"},
    "height": {"waymo": 320, "nuscenes": 288, "kitti": 192},
    "width": {"waymo": 480, "nuscenes": 512, "kitti": 640},
    "cam_name": {"waymo": "FRONT", "nuscenes": "FRONT", "kitti": "image_02"},
    "train_img_type": {"waymo": "downsample", "nuscenes": "downsample", "kitti": "downsample"},
    "eval_max_depth": {"waymo": 75, "nuscenes": 75, "kitti": 80},
    "eval_img_bound": {"waymo": [0, 1, 0, 1], "nuscenes": [0, 1, 0, 1],
                       "kitti": [0.40810811, 0.99189189, 0.03594771, 0.96405229]},   # Garg/Eigen crop
    "eval_img_ext": {"waymo": ".jpg", "nuscenes": ".jpg", "kitti": ".png"},
    "eval_img_type": {"waymo": "downsample", "nuscenes": "downsample", "kitti": "original"},
}

_DEFAULT_SCALES = {"monodepthv2": [0, 1, 2, 3], "litemono": [0, 1, 2]}


class DynamoOptions:
    def __init__(self):
        self.p = argparse.ArgumentParser(description="Dynamo options")
        for flags, kw in _SPEC + _EXTRA:
            self.p.add_argument(*flags, **kw)

    def parse(self, **kwargs):
        self.opt = opt = self.p.parse_args(**kwargs)
        if opt.scales is None:
            opt.scales = list(_DEFAULT_SCALES[opt.depth_model])
        if opt.data_path is None:
            opt.data_path = "data_dir/{}/".format(opt.dataset)
        for key, value in list(vars(opt).items()):
            if value is None and key in _DATASET_DEFAULTS:
                setattr(opt, key, _DATASET_DEFAULTS[key][opt.dataset])
        return opt
