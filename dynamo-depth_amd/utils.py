"""Small helpers kept from the reference's utils.py surface (readlines, write_to_file, join_dir, interp,
make_ind_map, cart2polar, hsv_to_rgb, sec_to_hm[_str], get_model_ckpt_name, get_filenames, is_edge).
The cv2 / imageio / matplotlib based video + colour-map helpers of the reference are visualisation only
(out of scope) and are imported lazily so the hot path has no such dependency."""
import os
import os.path as osp

import numpy as np
import torch
import torch.nn.functional as F


def readlines(filename):
    with open(filename, "r") as fh:
        return fh.read().splitlines()


def write_to_file(data_list, fname, bool_newline=True):
    with open(fname, "w") as fh:
        fh.writelines([d + "\n" for d in data_list] if bool_newline else data_list)


def join_dir(*tree):
    """os.path.join + makedirs, tolerant of racing ranks (reference utils.py:70-79)."""
    path = osp.join(*tree)
    try:
        os.makedirs(path, exist_ok=True)
    except OSError:
        pass
    return path


def interp(x, shape, mode="bilinear", align_corners=False):
    """(B,C,H,W) -> (B,C,*shape); bilinear, align_corners=False (reference utils.py:98-101)."""
    return F.interpolate(x, shape, mode=mode, align_corners=align_corners)


def get_model_ckpt_name(load_path):
    parts = load_path.split("/")
    if "logs" in parts:
        i = parts.index("logs")
        return parts[i + 1], parts[i + 3]
    if "ckpt" in parts:
        return parts[parts.index("ckpt") + 1], "ckpt"
    name = "[{}]".format("-".join(parts))
    print("Loaded path (={}) does not appear to be under logs/ or ckpt/".format(load_path))
    print("\tUsing general model_name=`{}` and ckpt_name=`ckpt`.".format(name))
    return name, "ckpt"


def get_filenames(segment_name, opt):
    rgb_dir = osp.join(opt.data_path, segment_name, opt.cam_name, "rgb", opt.eval_img_type)
    idx = sorted(int(osp.splitext(f)[0]) for f in os.listdir(rgb_dir) if osp.splitext(f)[1] == opt.eval_img_ext)
    return ["{} {}".format(segment_name, i) for i in idx]


def is_edge(filename, opt):
    seg, frame = filename.split()[0], int(filename.split()[1])
    lo, hi = frame + int(np.min(opt.frame_ids)), frame + int(np.max(opt.frame_ids))
    base = osp.join(opt.data_path, seg, opt.cam_name, "rgb", opt.eval_img_type)
    return not (osp.exists(osp.join(base, "{:06}{}".format(lo, opt.eval_img_ext)))
                and osp.exists(osp.join(base, "{:06}{}".format(hi, opt.eval_img_ext))))


def make_ind_map(height, width):
    """(1,H,W,2) identity sampling grid with top-left (-1,-1) (reference utils.py:119-125)."""
    v = torch.arange(0, height) / height * 2 - 1
    h = torch.arange(0, width) / width * 2 - 1
    return torch.stack([h.unsqueeze(0).repeat(height, 1), v.unsqueeze(1).repeat(1, width)]).permute(1, 2, 0).unsqueeze(0)


def cart2polar(cart):
    assert cart.shape[-1] == 2, "Last dimension must contain y and x vector component"
    r = torch.sqrt(torch.sum(cart ** 2, -1))
    theta = torch.atan(cart[..., 0] / cart[..., 1])
    theta = torch.where(torch.isnan(theta), torch.zeros_like(theta), theta)
    theta = theta + (cart[..., 1] < 0) * torch.pi
    return r, (5 * torch.pi / 2 - theta) % (2 * torch.pi)


def hsv_to_rgb(image):
    assert isinstance(image, torch.Tensor) and image.ndim >= 3 and image.shape[-3] == 3
    h, s, v = image[..., 0, :, :], image[..., 1, :, :], image[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    hi = hi.long()
    table = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    return torch.gather(table, -3, torch.stack([hi, hi + 6, hi + 12], dim=-3))


def sec_to_hm(t):
    t = int(t)
    return t // 3600, (t // 60) % 60, t % 60


def sec_to_hm_str(t):
    return "{:02d}h{:02d}m{:02d}s".format(*sec_to_hm(t))


def make_mp4(images, filename, fps=30, quality=8, macro_block_size=1, bgr=True):
    import imageio  # visualisation only
    if osp.splitext(filename)[1] == "":
        filename += ".mp4"
    frames = np.stack(images, axis=0)
    imageio.mimwrite(filename, frames[..., ::-1] if bgr else frames, fps=fps, quality=quality, macro_block_size=macro_block_size)


def score_map_vis(score_map, cmap="bone", vminmax=None, max_perc=95):
    import matplotlib as mpl
    import matplotlib.cm as cm
    arr = score_map.squeeze().cpu().numpy() if torch.is_tensor(score_map) else score_map
    vmin, vmax = (arr.min(), np.percentile(arr, max_perc)) if vminmax is None else vminmax
    return cm.ScalarMappable(norm=mpl.colors.Normalize(vmin=vmin, vmax=vmax), cmap=cmap).to_rgba(arr)[:, :, :3]
