"""Loss operators of Dynamo-Depth with the reference's Python signatures (tools.py:6-326), backed by the
HIP library instead of chains of ATen kernels.

This module is the drop-in operator boundary: BackprojectDepth, Project3D, SSIM,
GroundPlane, DepthMetrics, disp_to_depth, depth_to_disp, compute_smooth_loss, compute_errors, torch_and.
Every differentiable operator is a torch.autograd.Function over the C ABI (hipops.functions); tensors must
be on the GPU -- there is no CPU implementation of the loss path and none is silently substituted.
(DepthMetrics is evaluation bookkeeping on gathered LiDAR hits and stays in plain torch.)
"""
import numpy as np
import torch
import torch.nn as nn

from hipops import functions as HF
from hipops import lib as L


def disp_to_depth(disp, min_depth, max_depth):
    """Sigmoid disparity -> (scaled disparity, depth): 1/max + (1/min - 1/max)*disp (reference tools.py:291-298)."""
    return HF.DispToDepthFn.apply(disp, float(min_depth), float(max_depth))


def depth_to_disp(depth, min_depth, max_depth):
    """Inverse of disp_to_depth (reference tools.py:301-308)."""
    lo, hi = 1 / max_depth, 1 / min_depth
    return (1 / depth - lo) / (hi - lo)


def compute_smooth_loss(inp, img=None):
    """Edge-aware first-order smoothness of a (B,C,H,W) tensor (reference tools.py:311-326)."""
    return HF.SmoothLossFn.apply(inp, img, False)


class BackprojectDepth(nn.Module):
    """depth (B,1,h,w), inv_K (B,4,4) -> homogeneous points (B,4,h*w) (reference tools.py:167-197).
    `pix_coords` is kept as a buffer-like attribute because Trainer.get_ground_depth reads it."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        ys, xs = np.meshgrid(range(height), range(width), indexing="ij")
        pix = np.stack([xs.reshape(-1), ys.reshape(-1), np.ones(height * width)], 0).astype(np.float32)
        self.pix_coords = nn.Parameter(torch.from_numpy(pix).unsqueeze(0).repeat(batch_size, 1, 1), requires_grad=False)

    def forward(self, depth, inv_K):
        if depth.shape[-2:] != (self.height, self.width):
            raise ValueError("BackprojectDepth built for {}x{}, got {}".format(self.height, self.width, tuple(depth.shape[-2:])))
        return HF.BackprojectFn.apply(depth, inv_K)


class Project3D(nn.Module):
    """points (B,4,N), K, T|None -> (sampling grid (B,h,w,2) in [-1,1], ego motion (B,3,N)) (reference tools.py:200-224)."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        return HF.Project3DFn.apply(points, K, T, self.height, self.width, float(self.eps))


class SSIM(nn.Module):
    """clamp((1 - SSIM(x,y))/2, 0, 1) with 3x3 box statistics over reflection-padded images (reference tools.py:227-257)."""

    def __init__(self):
        super().__init__()
        self.C1, self.C2 = 0.01 ** 2, 0.03 ** 2

    def forward(self, x, y):
        return HF.SSIMFn.apply(x, y)


class GroundPlane(nn.Module):
    """RANSAC ground plane y = w1*x + w2*z + w3 on the bottom `g_prior` of a point map (reference tools.py:76-164).
    forward(points (B,3,H,W)) -> (distance (B,1,H,W), plane (B,3,1)), both detached.  The candidate indices come
    from the global NumPy RNG like the reference's (tools.py:125-127) unless `rand_idx` is injected."""

    def __init__(self, num_points_per_it=5, max_it=25, tol=0.1, g_prior=0.5, vertical_axis=1):
        super().__init__()
        if vertical_axis != 1:
            raise NotImplementedError("the HIP ground-plane kernels assume the camera convention y = down (vertical_axis=1)")
        self.num_points_per_it, self.max_it, self.tol, self.g_prior, self.vertical_axis = num_points_per_it, max_it, tol, g_prior, vertical_axis

    def draw_indices(self, B, H, W):
        n_ground = int(self.g_prior * H) * W
        total = self.num_points_per_it * self.max_it
        return np.stack([np.random.choice(np.arange(n_ground), total, replace=True) for _ in range(B)])

    def forward(self, points, rand_idx=None):
        import ctypes as C
        from hipops import abi
        pts = HF._dev(points.detach(), "points")
        B, _, H, W = pts.shape
        if rand_idx is None:
            rand_idx = self.draw_indices(B, H, W)
        ridx = torch.as_tensor(np.asarray(rand_idx), dtype=torch.int32).to(pts.device).contiguous()
        lib = L.load()
        dist = torch.empty(B, 1, H, W, dtype=torch.float32, device=pts.device)
        plane = torch.empty(B, 3, dtype=torch.float32, device=pts.device)
        ws = HF._ws(lib.dd_ground_workspace_bytes(B, H, W, self.max_it), pts.device)
        L.check(lib.dd_ground_plane(abi.ptr(pts), abi.ptr(ridx), B, H, W, self.num_points_per_it, self.max_it, float(self.tol),
                                    float(self.g_prior), abi.ptr(dist), abi.ptr(plane), abi.ptr(ws), L.current_stream()), "dd_ground_plane")
        return dist, plane.unsqueeze(-1)


def torch_and(*args):
    out = args[0]
    for a in args:
        assert out.size() == a.size(), "Sizes must match: [{}]".format(", ".join(str(x.size()) for x in args))
        out = torch.logical_and(out, a)
    return out


def compute_errors(gt, pred):
    """abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 (Eigen et al.), reference tools.py:269-288."""
    ratio = torch.max(gt / pred, pred / gt)
    a1, a2, a3 = [(ratio < 1.25 ** k).float().mean() for k in (1, 2, 3)]
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


class DepthMetrics(nn.Module):
    """Sparse-LiDAR depth metrics with per-image median scaling (reference tools.py:6-73)."""

    def __init__(self, img_bound, min_depth, max_depth):
        super().__init__()
        self.depth_metric_names = ["de:abs_rel", "de:sq_rel", "de:rms", "de:log_rms", "da:a1", "da:a2", "da:a3"]
        self.img_bound, self.min_depth, self.max_depth = img_bound, min_depth, max_depth

    def forward(self, inputs, outputs, mask=None):
        disp_pred = outputs[("disp_scaled", 0, 0)]
        names = self.depth_metric_names
        if mask is None and disp_pred.is_cuda:
            return dict(zip(names, self.device_metrics(inputs, disp_pred)[1].unbind(0)))
        if mask is not None and disp_pred.is_cuda and mask.dim() == 3 and not mask.is_floating_point() and int(mask.max()) < 256 and int(mask.min()) >= 0:
            return self.device_metrics_masked(inputs, disp_pred, mask)
        return self._forward_torch(inputs, outputs, mask)

    def device_metrics_masked(self, inputs, disp_pred, mask):
        """The mask branch (reference tools.py:23-25,58-72) on the device: per-label errors of every sample in the same launch
        as the unmasked metrics; the host only builds the dict (one transfer for the label set, one for the totals)."""
        import ctypes as C
        names = self.depth_metric_names
        B, _, H, W = disp_pred.shape
        disp = disp_pred.detach().to(torch.float32).contiguous()
        lidar = inputs["depth_gt"].to(disp.device, torch.float32).contiguous()
        valid = inputs["depth_valid"].to(disp.device, torch.float32).contiguous()
        dims = inputs["gt_dim"].to(disp.device, torch.int32).contiguous()
        m8 = mask.to(disp.device, torch.uint8).contiguous()
        M = lidar.shape[1]
        lib = L.load()
        per = torch.empty((B, 8), dtype=torch.float32, device=disp.device)
        mean = torch.empty(7, dtype=torch.float32, device=disp.device)
        per_label = torch.empty((B, 256, 8), dtype=torch.float32, device=disp.device)
        nbytes = lib.dd_depth_metrics_masked_workspace_bytes(B, M)
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=disp.device)
        bound = (C.c_double * 4)(*[float(v) for v in self.img_bound])
        L.check(lib.dd_depth_metrics_masked(disp.data_ptr(), B, H, W, lidar.data_ptr(), valid.data_ptr(), M, dims.data_ptr(), bound,
                                            float(self.min_depth), float(self.max_depth), m8.data_ptr(), m8.shape[1], m8.shape[2], per.data_ptr(),
                                            mean.data_ptr(), per_label.data_ptr(), ws.data_ptr(), nbytes, L.current_stream()), "dd_depth_metrics_masked")
        metrics = dict(zip(names, mean.unbind(0)))
        labels = [int(l) for l in torch.unique(m8).tolist()]
        cnt = per_label[:, :, 7]                                                     # (B,256)
        weighted = (per_label[:, :, :7].double() * cnt.double().unsqueeze(-1)).sum(0).tolist()     # sum_b err * cnt, as the reference accumulates
        total = cnt.double().sum(0).tolist()
        for i, m in enumerate(names):
            metrics["{}_mask".format(m)] = {l: [weighted[l][i], int(total[l])] for l in labels}
        return metrics

    def device_metrics(self, inputs, disp_pred):
        """(per_sample (B,8), mean (7,)) from dd_depth_metrics: one workgroup per sample, exact medians, no host sync
        (the per-sample loop with .item() syncs of the reference, tools.py:27-66, in one launch)."""
        import ctypes as C
        B, _, H, W = disp_pred.shape
        disp = disp_pred.detach().to(torch.float32).contiguous()
        lidar = inputs["depth_gt"].to(disp.device, torch.float32).contiguous()
        valid = inputs["depth_valid"].to(disp.device, torch.float32).contiguous()
        dims = inputs["gt_dim"].to(disp.device, torch.int32).contiguous()
        M = lidar.shape[1]
        lib = L.load()
        per = torch.empty((B, 8), dtype=torch.float32, device=disp.device)
        mean = torch.empty(7, dtype=torch.float32, device=disp.device)
        nbytes = lib.dd_depth_metrics_workspace_bytes(B, M)
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=disp.device)
        bound = (C.c_double * 4)(*[float(v) for v in self.img_bound])
        L.check(lib.dd_depth_metrics(disp.data_ptr(), B, H, W, lidar.data_ptr(), valid.data_ptr(), M, dims.data_ptr(), bound,
                                     float(self.min_depth), float(self.max_depth), per.data_ptr(), mean.data_ptr(), ws.data_ptr(), nbytes,
                                     L.current_stream()), "dd_depth_metrics")
        return per, mean

    def _forward_torch(self, inputs, outputs, mask=None):
        disp_pred = outputs[("disp_scaled", 0, 0)]
        names = self.depth_metric_names
        metrics = {m: 0 for m in names}
        labels = []
        if mask is not None:
            labels = [l.item() for l in torch.unique(mask)]
            metrics.update({"{}_mask".format(m): {l: [0, 0] for l in labels} for m in names})
        for bi in range(disp_pred.size(0)):
            lidar, valid = inputs["depth_gt"][bi], inputs["depth_valid"][bi]
            gh, gw = inputs["gt_dim"][bi][0].item(), inputs["gt_dim"][bi][1].item()
            top, bottom = int(self.img_bound[0] * gh), int(self.img_bound[1] * gh)
            left, right = int(self.img_bound[2] * gw), int(self.img_bound[3] * gw)
            keep = torch_and(valid, lidar[:, 0] >= top, lidar[:, 0] < bottom, lidar[:, 1] >= left, lidar[:, 1] < right,
                             lidar[:, 2] > self.min_depth, lidar[:, 2] < self.max_depth)
            rows, cols = lidar[:, 0][keep].long(), lidar[:, 1][keep].long()
            full = 1 / nn.functional.interpolate(disp_pred[bi][None], (gh, gw), mode="bilinear", align_corners=False).squeeze()
            gt, pd = lidar[:, 2][keep], full[rows, cols]
            pd = torch.clamp(pd * (torch.median(gt) / torch.median(pd)), self.min_depth, self.max_depth)
            errs = compute_errors(gt, pd)
            for i, m in enumerate(names):
                metrics[m] += errs[i]
            if mask is not None:
                at = mask[bi][rows, cols]
                for l in labels:
                    pick = at == l
                    cnt = int(pick.sum())
                    if cnt == 0:
                        continue
                    e2 = compute_errors(gt[pick], pd[pick])
                    for i, m in enumerate(names):
                        metrics["{}_mask".format(m)][l][0] += e2[i].item() * cnt
                        metrics["{}_mask".format(m)][l][1] += cnt
        for m in names:
            metrics[m] = metrics[m] / disp_pred.size(0)
        return metrics
