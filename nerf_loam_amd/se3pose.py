"""OptimizablePose: SE3 6-vector [t(3), w(3)] with R = I + A(theta)[w]x + B(theta)[w]x^2.

Host-side mirror of /root/reference/src/se3pose.py (same class name, constructor, `data` parameter,
matrix()/rotation()/translation()/from_matrix()/log()/copy_from()).  On the hot path the pose lives
on the GPU as a row of SdfEngine.pose6 and its gradient / Adam step run in nl_optim.hip
(k_pose_step); this class is the container the reference's callers hold and pickle."""
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn


def _taylor(x, kind, nth=10):
    """sin(x)/x ("A"), (1-cos x)/x^2 ("B") and (x-sin x)/x^3 ("C") by their (nth+1)-term series sum_i (-1)^i x^(2i) / (2i+k)!,
    k = 1, 2, 3 (se3pose.py:64-93)."""
    k = {"A": 1, "B": 2, "C": 3}[kind]
    ans = torch.zeros_like(x)
    denom = 1.0
    for j in range(2, k + 1):
        denom *= j                                           # k!
    for i in range(nth + 1):
        if i > 0:
            denom *= (2 * i + k - 1) * (2 * i + k)               # (2i+k)! from (2i+k-2)!
        ans = ans + (-1) ** i * x ** (2 * i) / denom
    return ans


class OptimizablePose(nn.Module):
    def __init__(self, init_pose):
        super().__init__()
        assert isinstance(init_pose, torch.Tensor) and init_pose.dtype == torch.float32
        self.register_parameter("data", nn.Parameter(init_pose))

    def copy_from(self, pose):
        self.data = deepcopy(pose.data)

    def translation(self):
        return self.data[:3]

    def rotation(self):
        w = self.data[3:]
        W = self.skew_symmetric(w)
        theta = w.norm(dim=-1)[..., None, None]
        eye = torch.eye(3, device=w.device, dtype=torch.float32)
        return eye + _taylor(theta, "A") * W + _taylor(theta, "B") * (W @ W)

    def matrix(self):
        Rt = torch.eye(4)
        Rt[:3, :3] = self.rotation()
        Rt[:3, 3] = self.translation()
        return Rt

    @classmethod
    def skew_symmetric(cls, w):
        w0, w1, w2 = w.unbind(dim=-1)
        z = torch.zeros_like(w0)
        return torch.stack([torch.stack([z, -w2, w1], -1), torch.stack([w2, z, -w0], -1), torch.stack([-w1, w0, z], -1)], -2)

    # the series helpers of the reference class (se3pose.py:64-93), kept for callers that use them directly
    @classmethod
    def taylor_A(cls, x, nth=10):
        return _taylor(x, "A", nth)

    @classmethod
    def taylor_B(cls, x, nth=10):
        return _taylor(x, "B", nth)

    @classmethod
    def taylor_C(cls, x, nth=10):
        return _taylor(x, "C", nth)

    @classmethod
    def log(cls, R, eps=1e-7):
        trace = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
        theta = ((trace - 1) / 2).clamp(-1 + eps, 1 - eps).acos_()[..., None, None] % np.pi
        lnR = 1 / (2 * _taylor(theta, "A") + 1e-8) * (R - R.transpose(-2, -1))
        return torch.stack([lnR[..., 2, 1], lnR[..., 0, 2], lnR[..., 1, 0]], dim=-1)

    @classmethod
    def from_matrix(cls, Rt, eps=1e-8):
        R, u = Rt[:3, :3], Rt[:3, 3]
        return OptimizablePose(torch.cat([u, cls.log(R)], dim=-1).detach().float().contiguous())
