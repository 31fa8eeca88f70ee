"""KITTI loader over the reference's processed layout
<data_path>/<folder>/image_0{2,3}/{rgb/<img_type>,depth}/<frame:010>.{jpg,png,npy} (reference datasets/kitti_dataset.py:7-128)."""
import os

import numpy as np
import PIL.Image as pil

from .base_dataset import BaseDataset


class KITTIDataset(BaseDataset):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # intrinsics normalised by the image size; 4th column zero
        self.K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
        self.full_res_shape = (1242, 375)
        self.side_map = {"2": 2, "3": 3, "l": 2, "r": 3}

    def get_intrinsic(self, folder):
        return self.K

    def get_gt_dim(self, folder, frame_index, side):
        tag = "S_rect_0{}".format(self.side_map[side])
        with open(os.path.join(self.data_path, folder, "calib_cam_to_cam.txt"), "r") as fh:
            _, width, height = [l for l in fh.read().splitlines() if tag in l][0].split()
        return int(float(height)), int(float(width))

    def _cam(self, side):
        return "image_0{}".format(self.side_map[side])

    def get_img_path(self, folder, frame_index, side):
        return os.path.join(self.data_path, folder, self._cam(side), "rgb", self.img_type, "{:010}{}".format(frame_index, self.img_ext))

    def get_color(self, folder, frame_index, side, do_flip):
        img = self.loader(self.get_img_path(folder, max(frame_index, 0) if frame_index == -1 else frame_index, side))
        return img.transpose(pil.FLIP_LEFT_RIGHT) if do_flip else img

    def get_color_bytes(self, folder, frame_index, side):
        with open(self.get_img_path(folder, max(frame_index, 0) if frame_index == -1 else frame_index, side), "rb") as fh:
            return fh.read()

    def get_depth(self, folder, frame_index, side, do_flip):
        if frame_index == -1:
            frame_index = 0
        lidar = np.load(os.path.join(self.data_path, folder, self._cam(side), "depth", "{:010}.npy".format(frame_index)))
        if do_flip:
            lidar[:, 1] = self.full_res_shape[0] - lidar[:, 1]
        lidar[:, 0] = np.minimum(lidar[:, 0], self.full_res_shape[1] - 1)
        lidar[:, 1] = np.minimum(lidar[:, 1], self.full_res_shape[0] - 1)
        return lidar

    def get_mask(self, folder, frame_index, side, do_flip):
        if frame_index == -1:
            frame_index = 0
        base = os.path.join(self.data_path, folder, self._cam(side), "mask", "{:010}".format(frame_index))
        if not os.path.exists(base + "_sem.npy"):
            z = np.zeros(self.full_res_shape[::-1])
            return z, z.copy()
        return np.load(base + "_sem.npy"), np.load(base + "_mot.npy")
