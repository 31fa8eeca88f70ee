"""TEST INFRASTRUCTURE -- CPU restatement of the INPUT SIDE of a training step, in plain torch fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(dynamo-depth_amd/) never does.

What it restates (reference paths relative to the reference checkout):
  * datasets/base_dataset.py:83-95 (`preprocess`): `f = ToTensor(img)`; `('color',f,0) = f`; `('color_aug',f,0) = color_aug(f)`
    where `color_aug` is torchvision's `ColorJitter` object applied to a float TENSOR -- so its `forward` draws a fresh
    permutation and fresh factors on every call, i.e. per FRAME (base_dataset.py:159-164 picks jitter-or-identity per sample);
  * datasets/base_dataset.py:118-131: horizontal flip of the loaded frames (`get_color(..., do_flip)`);
  * Trainer.py:722-734 (`apply_img_resize`): ('color',0,s) = clamp(Resize(BICUBIC, antialias)(('color',0,s-1)), 0, 1).

Third-party arithmetic: torchvision 0.13.1 (README.md:31) is absent from this image, so `ColorJitter` cannot be executed here:
its tensor path (`torchvision.transforms.functional_tensor`: `_blend`, `rgb_to_grayscale`, `adjust_brightness/contrast/
saturation/hue`, `_rgb2hsv`, `_hsv2rgb`, and `ColorJitter.forward`'s op loop) is restated below from its published source --
PARITY UNPINNED at exactly this point.  The tensor `Resize` is `F.interpolate(mode='bicubic',
align_corners=False, antialias=True)`, executed by torch itself.
"""
import torch
import torch.nn.functional as F


def rgb_to_grayscale(img):
    """functional_tensor.rgb_to_grayscale: (…,3,H,W) -> (…,1,H,W)."""
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(dim=-3)


def _blend(img1, img2, ratio):
    ratio = float(ratio)
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 1.0)


def adjust_brightness(img, factor):
    return _blend(img, torch.zeros_like(img), factor)


def adjust_contrast(img, factor):
    mean = torch.mean(rgb_to_grayscale(img), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, factor)


def adjust_saturation(img, factor):
    return _blend(img, rgb_to_grayscale(img), factor)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = hr + hg + hb
    h = torch.fmod((h / 6.0 + 1.0), 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp((v * (1.0 - s)), 0.0, 1.0)
    q = torch.clamp((v * (1.0 - (s * f))), 0.0, 1.0)
    t = torch.clamp((v * (1.0 - (s * (1.0 - f)))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img, hue_factor):
    h, s, v = _rgb2hsv(img).unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


def color_jitter(img, order, brightness, contrast, saturation, hue):
    """ColorJitter.forward with injected draws: `order` = the permutation fn_idx (0 brightness, 1 contrast, 2 saturation, 3 hue)."""
    for fn_id in order:
        fn_id = int(fn_id)
        if fn_id == 0:
            img = adjust_brightness(img, brightness)
        elif fn_id == 1:
            img = adjust_contrast(img, contrast)
        elif fn_id == 2:
            img = adjust_saturation(img, saturation)
        else:
            img = adjust_hue(img, hue)
    return img


def prepare_inputs(frames_u8, params, flip, scales, target=0):
    """frames_u8: {f: (B,H,W,3) uint8} decoded, resized frames; params: {f: (B,9) float [apply, order0..3, b, c, s, h]};
    flip: (B,) bool.  Returns the loader/Trainer dict entries ('color',f,0), ('color_aug',f,0), ('color',target,s)."""
    out = {}
    for f, u8 in frames_u8.items():
        B = u8.shape[0]
        color = u8.permute(0, 3, 1, 2).float().div(255)          # ToTensor
        color = torch.stack([color[b].flip(-1) if bool(flip[b]) else color[b] for b in range(B)])
        aug = []
        for b in range(B):
            p = params[f][b]
            if float(p[0]) > 0.5:
                aug.append(color_jitter(color[b], [int(x) for x in p[1:5]], float(p[5]), float(p[6]), float(p[7]), float(p[8])))
            else:
                aug.append(color[b])
        out[("color", f, 0)] = color
        out[("color_aug", f, 0)] = torch.stack(aug)
    prev = out[("color", target, 0)]
    for s in scales:
        if s == 0:
            continue
        h, w = prev.shape[-2] // 2, prev.shape[-1] // 2
        prev = torch.clamp(F.interpolate(prev, (h, w), mode="bicubic", align_corners=False, antialias=True), 0, 1)
        out[("color", target, s)] = prev
    return out
