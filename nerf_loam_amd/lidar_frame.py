"""LidarFrame: one scan's returns, unit ray directions and pose.

Mirror of /root/reference/src/lidarFrame.py (+ src/utils/sample_util.py for the per-iteration ray
subset): same constructor and accessors, the +2000 m world offset (lidarFrame.py:18), rays_d =
points / (||points|| + 1e-8), and `sample_rays(N)` drawing N rays without replacement by Gumbel
top-k on the host and storing a boolean `sample_mask` (rays keep dataset order)."""
import numpy as np
import torch
import torch.nn as nn

from .se3pose import OptimizablePose


def sampling_without_replacement(logp, k):
    g = -torch.log(-torch.log(torch.rand_like(logp) + 1e-7) + 1e-7)
    return (logp + g).topk(k, dim=-1)[1]


def sample_rays(mask, num_samples):
    B, H, W = mask.shape
    probs = (mask / (mask.sum() + 1e-9)).reshape(B, -1)
    idx = sampling_without_replacement(torch.log(probs + 1e-9), num_samples)
    return (torch.zeros_like(probs).scatter_(-1, idx, 1).reshape(B, H, W) > 0)


class LidarFrame(nn.Module):
    def __init__(self, index, points, pointsCos, pose=None, new_keyframe=False):
        super().__init__()
        self.index = index
        self.num_point = len(points)
        self.points = points
        self.pointsCos = pointsCos
        if (not new_keyframe) and (pose is not None):
            pose = np.array(pose, dtype=np.float64, copy=True)
            pose[:3, 3] += 2000
            self.pose = OptimizablePose.from_matrix(torch.tensor(pose, dtype=torch.float32))
        elif new_keyframe:
            self.pose = pose
        self.rays_d = self.get_rays()
        self.rel_pose = None
        self.sample_mask = None

    def get_pose(self):
        return self.pose.matrix()

    def get_translation(self):
        return self.pose.translation()

    def get_rotation(self):
        return self.pose.rotation()

    def get_points(self):
        return self.points

    def get_pointsCos(self):
        return self.pointsCos

    def set_rel_pose(self, rel_pose):
        self.rel_pose = rel_pose

    def get_rel_pose(self):
        return self.rel_pose

    @torch.no_grad()
    def get_rays(self):
        self.rays_norm = torch.norm(self.points, 2, -1, keepdim=True) + 1e-8
        return (self.points / self.rays_norm).unsqueeze(1).float()

    @torch.no_grad()
    def device_scan(self, device):
        """the frame's returns resident on `device` (cached): input of SdfEngine.select_rays (on-device ray selection)"""
        sc = getattr(self, "_device_scan", None)
        if sc is None or sc["dirs"].device != torch.device(device):
            sc = dict(dirs=self.rays_d.reshape(-1, 3).float().contiguous().to(device),
                      points=self.points.reshape(-1, 3).float().contiguous().to(device),
                      cos=self.pointsCos.reshape(-1).float().contiguous().to(device),
                      mask_u8=torch.zeros(self.num_point, dtype=torch.uint8, device=device))
            self._device_scan = sc
        return sc

    @torch.no_grad()
    def sample_rays(self, N_rays, track=False):
        self.sample_mask = sample_rays(torch.ones((self.num_point, 1))[None, ...], N_rays)[0, ...]
