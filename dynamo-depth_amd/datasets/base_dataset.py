"""Item contract of the Dynamo-Depth loaders (reference datasets/base_dataset.py:24-204): one dict per sample with
('color',f,0), ('color_aug',f,0) (3,H,W) float in [0,1]; ('K',s), ('inv_K',s) (4,4) for every scale; ('ts',f);
'gt_dim'; optionally 'depth_gt' (25000,3) [row, col, z] + 'depth_valid' (25000,); 'index'.
torchvision is not available on the MI355X image, so ToTensor / Resize / ColorJitter are restated on PIL + torch."""
import random

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image


def pil_loader(path):
    with open(path, "rb") as fh:
        with Image.open(fh) as img:
            return img.convert("RGB")


def to_tensor(pic):
    return torch.from_numpy(np.asarray(pic, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)


def _gray(x):
    return 0.2989 * x[0:1] + 0.587 * x[1:2] + 0.114 * x[2:3]


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1)


class ColorJitter:
    """torchvision.transforms.ColorJitter on float tensors, as the reference uses it (datasets/base_dataset.py:60-71,83-95):
    brightness / contrast / saturation factors in [0.8,1.2], hue shift in [-0.1,0.1], applied in a random order.  The
    arithmetic follows torchvision.transforms.functional_tensor (_blend, rgb_to_grayscale, _rgb2hsv, _hsv2rgb), restated --
    torchvision is not installed on the MI355X image.  `draw()` is one call of ColorJitter.get_params."""

    def __init__(self, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1)):
        self.ranges = (brightness, contrast, saturation, hue)

    def draw(self):
        vals = [random.uniform(*r) for r in self.ranges]
        order = list(range(4))
        random.shuffle(order)
        return order, vals

    @staticmethod
    def row(params):
        """The 9-float row the device kernel reads: [apply, fn_idx[4], brightness, contrast, saturation, hue]."""
        if params is None:
            return torch.tensor([0.0, 0, 1, 2, 3, 1.0, 1.0, 1.0, 0.0])
        order, vals = params
        return torch.tensor([1.0] + [float(o) for o in order] + [float(v) for v in vals])

    @staticmethod
    def _hue(x, shift):
        r, g, b = x[0], x[1], x[2]
        maxc, minc = x.max(0)[0], x.min(0)[0]
        eqc = maxc == minc
        cr = maxc - minc
        ones = torch.ones_like(maxc)
        s = cr / torch.where(eqc, ones, maxc)
        div = torch.where(eqc, ones, cr)
        rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
        hr = (maxc == r) * (bc - gc)
        hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
        hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
        h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
        h = (h + shift) % 1.0
        v = maxc
        i = torch.floor(h * 6.0)
        f = h * 6.0 - i
        p = (v * (1.0 - s)).clamp(0, 1)
        q = (v * (1.0 - s * f)).clamp(0, 1)
        t = (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
        i = i.long() % 6
        sel = lambda a0, a1, a2, a3, a4, a5: torch.stack([a0, a1, a2, a3, a4, a5]).gather(0, i[None])[0]  # noqa: E731
        return torch.stack([sel(v, q, p, p, t, v), sel(t, v, v, q, p, p), sel(p, p, t, v, v, q)])

    def apply(self, x, params):
        order, (bri, con, sat, hue) = params
        for op in order:
            if op == 0:
                x = _blend(x, torch.zeros_like(x), bri)
            elif op == 1:
                x = _blend(x, _gray(x).mean(), con)
            elif op == 2:
                x = _blend(x, _gray(x), sat)
            else:
                x = self._hue(x, hue)
        return x


class BaseDataset(data.Dataset):
    def __init__(self, data_path, filenames, height, width, cam_name, img_type, frame_idxs, num_scales, is_train=False,
                 img_ext=".jpg", load_depth=False, load_mask=False, path=False, device_preprocess=False, jitter_per_frame=True,
                 device_decode=False, device_resize=True):
        super().__init__()
        self.data_path, self.filenames = data_path, filenames
        self.height, self.width = height, width
        self.cam_name, self.img_type = cam_name, img_type
        self.num_scales, self.frame_idxs = num_scales, frame_idxs
        self.is_train, self.img_ext = is_train, img_ext
        self.loader = pil_loader
        self.jitter = ColorJitter()
        self.aug_freq = 0.5
        self.give_path, self.load_mask, self.load_depth = path, load_mask, load_depth
        self.max_lidar_num = 25000
        # device_preprocess: hand over the decoded uint8 frames + the drawn augmentation parameters; ToTensor, flip, jitter and
        # the target pyramid then run on the GPU (hipops.inputs) instead of in the DataLoader workers.
        # jitter_per_frame: the reference passes its ColorJitter OBJECT to preprocess() and calls it once per frame on a tensor
        # (base_dataset.py:92-95,159-164), so every frame of a triplet gets its own draw -- despite the docstring's stated
        # intent; True reproduces that behaviour, False draws once per sample.
        self.device_preprocess, self.jitter_per_frame = device_preprocess, jitter_per_frame
        # device_decode (with device_preprocess): hand over the COMPRESSED frames -- the file bytes and the parsed marker segments
        # -- and let the GPU decode them (hipops.jpeg, bit for bit PIL's result): the workers then only read files.  Holds for
        # baseline colour JPEGs; decided once on the first sample, anything else keeps the PIL path.  device_resize: frames of
        # another size than the training resolution (KITTI's `original` image type: four sizes around 1242x375) are resized on the
        # GPU as well (hipops.resize: Pillow's bicubic, bit for bit); without it only files at the training resolution qualify.
        self._device_decode = None if (device_decode and device_preprocess and img_ext in (".jpg", ".jpeg")) else False
        self.device_resize = bool(device_resize)

    def __len__(self):
        return len(self.filenames)

    @property
    def device_decode(self):
        if self._device_decode is None:             # decided on first use: the subclass has finished its constructor by then
            self._device_decode = self._probe_device_decode()
        return self._device_decode

    def _sample_parts(self, index):
        parts = self.filenames[index].split()
        return parts[0], int(parts[1]), (parts[2] if len(parts) == 3 else "l")

    def _probe_device_decode(self):
        try:
            from hipops import jpeg
            folder, frame, side = self._sample_parts(0)
            data = self.get_color_bytes(folder, frame, side)
            _, geom = jpeg.parse_header(data)
        except Exception:
            return False
        if geom[2] != 3 or (geom[:2] != (self.width, self.height) and not self.device_resize):
            return False                        # greyscale, or another size without the device resize: PIL on the host
        # fixed record size for the collate: twice the first file, at least a quarter of its raw pixels (a larger frame travels as pixels)
        self._jpeg_cap = max(4096, (max(geom[0] * geom[1] * 3 // 4, 2 * len(data)) + 4095) // 4096 * 4096)
        return True

    def _compressed_frame(self, folder, frame_index, side):
        """(zero-padded file bytes, DDJpegHeader record) of a frame the device decoder takes, None for any other frame -- a
        progressive / greyscale / CMYK file, more bytes than the fixed record holds (a quality-100 frame), another size than the
        training resolution when the device resize is off: the sample then travels as decoded pixels (PIL, like the reference) and `collate` turns the
        rest of its batch into pixels too.  One unusual file must not end a run."""
        from hipops import jpeg
        data = self.get_color_bytes(folder, frame_index, side)
        try:
            rec, geom = jpeg.parse_header(data)
        except jpeg.UnsupportedJpeg:
            return None
        if geom[2] != 3 or len(data) > self._jpeg_cap or (geom[:2] != (self.width, self.height) and not self.device_resize):
            return None
        buf = np.zeros(self._jpeg_cap, dtype=np.uint8)
        buf[:len(data)] = np.frombuffer(data, dtype=np.uint8)
        return buf, rec

    def _host_frame(self, folder, frame_index, side):
        img = self.get_color(folder, frame_index, side, False)
        if img.size != (self.width, self.height):
            img = img.resize((self.width, self.height), Image.BICUBIC)
        return np.asarray(img, dtype=np.uint8)

    def collate(self, samples):
        """DataLoader collate_fn: batches are uniform -- either every sample carries compressed frames (device decode) or every
        sample carries pixels.  A batch with one host-decoded sample (see _compressed_frame) has its other samples decoded here
        from the bytes they carry (PIL: the same pixels the device decoder produces, bit for bit)."""
        if any("frames_u8" in s for s in samples) and any("jpeg_bytes" in s for s in samples):
            import io
            import struct
            for s in samples:
                if "jpeg_bytes" not in s:
                    continue
                raw, hdr = s.pop("jpeg_bytes").numpy(), s.pop("jpeg_hdr").numpy()
                frames = []
                for i in range(raw.shape[0]):
                    (length,) = struct.unpack_from("<i", hdr[i].tobytes(), 4)          # DDJpegHeader.data_end = the file's length
                    with Image.open(io.BytesIO(raw[i, :length].tobytes())) as img:
                        img = img.convert("RGB")
                        if img.size != (self.width, self.height):           # device_resize: the bytes are the file's native size;
                            img = img.resize((self.width, self.height), Image.BICUBIC)      # the host-decoded sample came through _host_frame
                        frames.append(np.asarray(img, dtype=np.uint8))
                s["frames_u8"] = torch.from_numpy(np.stack(frames))
        return data.default_collate(samples)

    def __getitem__(self, index):
        item = {}
        flip = self.is_train and random.random() > 0.5
        parts = self.filenames[index].split()
        folder, frame = parts[0], int(parts[1])
        side = parts[2] if len(parts) == 3 else "l"
        compressed = []
        on_device = self.device_decode
        if on_device:
            compressed = [self._compressed_frame(folder, frame + f, side) for f in self.frame_idxs]
            if any(c is None for c in compressed):
                on_device = False               # this sample travels as pixels
                compressed = [self._host_frame(folder, frame + f, side) for f in self.frame_idxs]
        for f in self.frame_idxs:
            if not self.device_decode:
                img = self.get_color(folder, frame + f, side, flip and not self.device_preprocess)
                if img.size != (self.width, self.height):
                    img = img.resize((self.width, self.height), Image.BICUBIC)
                item[("color", f, 0)] = img
            item[("ts", f)] = self.get_timestep(folder, frame, f)
            gh, gw = self.get_gt_dim(folder, frame + f, side)
            item["gt_dim"] = torch.tensor([gh, gw]).type(torch.int)
        for s in range(self.num_scales):
            K = self.get_intrinsic(folder).copy()
            K[0, :] *= self.width // (2 ** s)
            K[1, :] *= self.height // (2 ** s)
            item[("K", s)] = torch.from_numpy(K)
            item[("inv_K", s)] = torch.from_numpy(np.linalg.pinv(K))
        augment = self.is_train and random.random() < self.aug_freq
        shared = self.jitter.draw() if (augment and not self.jitter_per_frame) else None
        drawn = {f: ((self.jitter.draw() if self.jitter_per_frame else shared) if augment else None) for f in self.frame_idxs}
        if self.device_preprocess:
            if on_device:
                item["jpeg_bytes"] = torch.from_numpy(np.stack([c[0] for c in compressed]))     # (F,cap) file bytes, zero padded
                item["jpeg_hdr"] = torch.from_numpy(np.stack([c[1] for c in compressed]))       # (F,HEADER_BYTES) DDJpegHeader records
            elif self.device_decode:
                item["frames_u8"] = torch.from_numpy(np.stack(compressed))              # the host-decoded exception (see _compressed_frame)
            else:
                frames = [np.asarray(item.pop(("color", f, 0)), dtype=np.uint8) for f in self.frame_idxs]
                item["frames_u8"] = torch.from_numpy(np.stack(frames))                  # (F,H,W,3), frame order = frame_idxs
            item["jitter"] = torch.stack([ColorJitter.row(drawn[f]) for f in self.frame_idxs])
            item["flip"] = torch.tensor(int(flip), dtype=torch.int32)
        else:
            for f in self.frame_idxs:
                t = to_tensor(item[("color", f, 0)])
                item[("color", f, 0)] = t
                item[("color_aug", f, 0)] = t if drawn[f] is None else self.jitter.apply(t, drawn[f])
        if self.load_depth:
            lidar = torch.from_numpy(self.get_depth(folder, frame, side, flip).astype(np.float32))
            pad = self.max_lidar_num - lidar.shape[0]
            item["depth_gt"] = torch.cat((lidar, torch.zeros(pad, 3)))
            item["depth_valid"] = torch.cat((torch.ones(lidar.shape[0]), torch.zeros(pad)))
        if self.load_mask:
            sem, mot = self.get_mask(folder, frame, side, flip)
            item["sem_mask"] = torch.from_numpy(sem).type(torch.uint8)
            item["mot_mask"] = torch.from_numpy(mot).type(torch.uint8)
        if self.give_path:
            item["paths"] = parts
        item["index"] = index
        return item

    # dataset specific
    def get_color(self, folder, frame_index, side, do_flip):
        raise NotImplementedError

    def get_color_bytes(self, folder, frame_index, side):
        """The frame's image file as bytes (device_decode); datasets that can name the file implement it."""
        raise NotImplementedError

    def get_depth(self, folder, frame_index, side, do_flip):
        raise NotImplementedError

    def get_mask(self, folder, frame_index, side, do_flip):
        raise NotImplementedError

    def get_intrinsic(self, folder):
        raise NotImplementedError

    def get_timestep(self, folder, frame_index, offset):
        return 1

    def get_gt_dim(self, folder, frame_index, side):
        raise NotImplementedError
