"""Input side of a training step on the device: ToTensor + flip + per-frame ColorJitter of the uploaded
uint8 triplets, the bicubic-antialias target pyramid, and a double-buffered prefetcher that overlaps the upload and these
kernels with the previous step.  Reference: datasets/base_dataset.py:83-95,118-131,159-164; Trainer.py:722-734."""
import ctypes as C

import torch

from . import abi
from . import lib as L


def prepare_frames(frames_u8, params, flip):
    """frames_u8 (B,F,H,W,3) uint8, params (B,F,9) fp32, flip (B,) int32, all on the GPU -> color, color_aug (F,B,3,H,W) fp32."""
    if not frames_u8.is_cuda:
        raise L.DynamoHipError("prepare_frames runs on the GPU (the host loaders prepare their samples themselves)")
    lib = L.load()
    B, F, H, W, _ = frames_u8.shape
    frames_u8 = frames_u8.contiguous()
    params = params.to(device=frames_u8.device, dtype=torch.float32).contiguous()
    flip = flip.to(device=frames_u8.device, dtype=torch.int32).contiguous()
    color = torch.empty(F, B, 3, H, W, dtype=torch.float32, device=frames_u8.device)
    aug = torch.empty_like(color)
    ws = torch.empty(max(lib.dd_prepare_frames_workspace_bytes(B, F) // 4, 1), dtype=torch.float32, device=frames_u8.device)
    L.check(lib.dd_prepare_frames(abi.ptr(frames_u8), abi.ptr(params), abi.ptr(flip), B, F, H, W, abi.ptr(color), abi.ptr(aug), abi.ptr(ws),
                                  L.current_stream()), "dd_prepare_frames")
    return color, aug


def pyramid_down2(img):
    """clamp(bicubic-antialias resize to half size, 0, 1) of a (B,C,H,W) fp32 GPU tensor (one level of Trainer.apply_img_resize)."""
    if not img.is_cuda:
        raise L.DynamoHipError("pyramid_down2 runs on the GPU")
    img = img.float().contiguous()
    B, Cc, H, W = img.shape
    out = torch.empty(B, Cc, H // 2, W // 2, dtype=torch.float32, device=img.device)
    L.check(L.load().dd_pyramid_down2(abi.ptr(img), B * Cc, H, W, abi.ptr(out), L.current_stream()), "dd_pyramid_down2")
    return out


def pack_rgb(img):
    """(B,3,H,W) fp32 GPU tensor -> (B,H,W,3) pixel-interleaved copy (same values): the layout the photometric kernel gathers its
    source taps from (DDPhotoArgs.source_packed; one 12-byte load per tap instead of three planes)."""
    if not img.is_cuda:
        raise L.DynamoHipError("pack_rgb runs on the GPU")
    img = img.float().contiguous()
    B, Cc, H, W = img.shape
    if Cc != 3 or (H * W) % 4:
        raise L.DynamoHipError("pack_rgb takes (B,3,H,W) images with H*W a multiple of 4, got %r" % (tuple(img.shape),))
    out = torch.empty(B, H, W, 3, dtype=torch.float32, device=img.device)
    L.check(L.load().dd_pack_rgb(abi.ptr(img), B, H, W, abi.ptr(out), L.current_stream()), "dd_pack_rgb")
    return out


class DevicePrefetcher:
    """Iterates a DataLoader one batch ahead: while step k runs, batch k+1 is uploaded from pinned memory and prepared
    (`prepare`: Trainer.process_inputs -- ToTensor / flip / jitter / pyramid kernels) on a side stream.  The consumer's
    stream waits on the batch's event; the tensors are handed over with record_stream so that the caching allocator
    does not recycle them early."""

    def __init__(self, loader, prepare, device):
        self.loader, self.prepare, self.device = loader, prepare, device
        self.stream = torch.cuda.Stream(device=device)
        self._events = []          # the last few hand-over events stay alive: the consumer may be enqueued steps ahead of the GPU

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        with torch.cuda.stream(self.stream):
            self.prepare(batch)
            event = torch.cuda.Event()
            event.record(self.stream)
        return batch, event

    def __iter__(self):
        it = iter(self.loader)
        try:
            staged = self._stage(next(it))
        except StopIteration:
            return
        while staged is not None:
            batch, event = staged
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            main = torch.cuda.current_stream(self.device)
            main.wait_event(event)
            self._events = (self._events + [event])[-8:]
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)
            staged = self._stage(nxt) if nxt is not None else None      # enqueued behind nothing the consumer waits for
            yield batch
