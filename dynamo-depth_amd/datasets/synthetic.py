"""Synthetic triplets with the reference's item contract, for throughput runs without a dataset on disk
(SURVEY.md 8(d)): low-frequency texture shifted by 2*f pixels between frames + pixel noise, KITTI-normalised
intrinsics, ts = 1.  Waymo / nuScenes readers of the reference need the processed datasets (absent here); with
--synthetic their shapes are served by the same generator."""
import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as data

NORMALISED_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def synth_frames(gen, height, width, frame_ids, shift=2):
    base = torch.rand(1, 3, height // 8 + 2, width // 8 + 2, generator=gen)
    big = F.interpolate(base, (height + 16, width + 16), mode="bilinear", align_corners=False)[0]
    out = {}
    for f in frame_ids:
        x0 = 8 + shift * f
        out[f] = (big[:, 8:8 + height, x0:x0 + width] + 0.05 * torch.rand(3, height, width, generator=gen)).clamp(0, 1).contiguous()
    return out


class SyntheticTriplets(data.Dataset):
    def __init__(self, data_path=None, filenames=None, height=192, width=640, cam_name=None, img_type=None, frame_idxs=(0, -1, 1),
                 num_scales=3, is_train=False, img_ext=".jpg", load_depth=False, load_mask=False, path=False, length=None, seed=0):
        self.height, self.width, self.frame_idxs, self.num_scales = height, width, list(frame_idxs), num_scales
        self.length = length if length is not None else (len(filenames) if filenames is not None else 1024)
        self.load_depth, self.seed = load_depth, seed
        self.max_lidar_num = 25000

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        gen = torch.Generator().manual_seed(self.seed * 1000003 + index)
        item = {}
        for f, img in synth_frames(gen, self.height, self.width, self.frame_idxs).items():
            item[("color", f, 0)] = img
            item[("color_aug", f, 0)] = img
            item[("ts", f)] = 1
        for s in range(self.num_scales):
            K = NORMALISED_K.copy()
            K[0, :] *= self.width // (2 ** s)
            K[1, :] *= self.height // (2 ** s)
            item[("K", s)] = torch.from_numpy(K)
            item[("inv_K", s)] = torch.from_numpy(np.linalg.pinv(K))
        item["gt_dim"] = torch.tensor([self.height, self.width]).type(torch.int)
        if self.load_depth:
            n = 2000
            rows = torch.randint(0, self.height, (n,), generator=gen).float()
            cols = torch.randint(0, self.width, (n,), generator=gen).float()
            z = 2 + 40 * torch.rand(n, generator=gen)
            lidar = torch.stack([rows, cols, z], 1)
            item["depth_gt"] = torch.cat((lidar, torch.zeros(self.max_lidar_num - n, 3)))
            item["depth_valid"] = torch.cat((torch.ones(n), torch.zeros(self.max_lidar_num - n)))
        item["index"] = index
        return item


class _NeedsData(SyntheticTriplets):
    name = "?"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "the {} reader needs the processed dataset of the reference's prepare_data/ (not available in this build); "
            "run with --synthetic to train on synthetic triplets of the {} shape".format(self.name, self.name))


class WaymoDataset(_NeedsData):
    name = "waymo"


class nuScenesDataset(_NeedsData):
    name = "nuscenes"
