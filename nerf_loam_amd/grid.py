"""Drop-in for the reference's `grid` extension module (the two live entry points).

/root/reference/third_party/sparse_voxels/src/binding.cpp:10-21: `svo_intersect`,
`inverse_cdf_sampling` - same argument order, tensor layouts, dtypes, freshly allocated outputs on
the input device, asynchronous on the current stream.  Like the reference (CHECK_CUDA,
intersect.cpp:93-96) CPU tensors are rejected: there is no CPU path."""
import torch

from . import ops


def svo_intersect(ray_start, ray_dir, points, children, voxelsize, n_max):
    idx = torch.empty(ray_start.shape[0], ray_start.shape[1], n_max, dtype=torch.int32, device=ray_start.device)
    t0 = torch.empty(idx.shape, dtype=torch.float32, device=ray_start.device)
    t1 = torch.empty(idx.shape, dtype=torch.float32, device=ray_start.device)
    ops.svo_intersect(ray_start, ray_dir, points, children, voxelsize, n_max, idx, t0, t1)
    return idx, t0, t1


def inverse_cdf_sampling(pts_idx, min_depth, max_depth, uniform_noise, probs, steps, fixed_step_size):
    T = uniform_noise.shape[-1]
    shape = (pts_idx.shape[0], pts_idx.shape[1], T)
    s_idx = torch.full(shape, -1, dtype=torch.int32, device=pts_idx.device)
    s_depth = torch.zeros(shape, dtype=torch.float32, device=pts_idx.device)
    s_dists = torch.zeros(shape, dtype=torch.float32, device=pts_idx.device)
    ops.inverse_cdf_sampling(pts_idx, min_depth, max_depth, uniform_noise, probs, steps, fixed_step_size, s_idx, s_depth, s_dists)
    return s_idx, s_depth, s_dists
