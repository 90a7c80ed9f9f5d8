"""GPU (-m gpu): the REFERENCE's own `grid` kernels (third_party/sparse_voxels, hipify-built for gfx950 by
oracle/build_grid_ref.py into oracle/_ref/grid_ref*.so - test infrastructure, never importable from nerf_loam_amd/)
against (a) the C restatement the oracle and every golden rest on (oracle/nl_oracle.c) and (b) the HIP product
(nerf_loam_amd.grid), fed the tensors the reference's Python wrappers would pass (voxel_helpers.py:92-133 G-way batching with
the octree replicated per batch row; :262-347 the [200, L, P] sampler layout in chunks of 800).

This pins the restatement of intersect_gpu.cu:193-272 / sample_gpu.cu:133-239 to the reference itself (missing #1).

Stated tolerances:
  * svo_intersect: idx, t_min, t_max BIT-EXACT.  The slab test has no a*b+c to contract and HIP's __fdividef(1, x) is the
    IEEE division 1.0f / x (clang __clang_hip_math.h), which is what the restatement and the product use.
  * inverse_cdf_sampling: sampled_idx bit-exact.  Depths / dists bit-exact against grid_ref_nc.so (the reference sources
    compiled with -ffp-contract=off); against grid_ref.so (hipcc default: a*b+c contracted to FMA, as nvcc -O2 does) within
    4 ulp of the depth for sampled_depth and 4 ulp of the DEPTH (not of the small difference) for sampled_dists = z - z_low.
"""
import importlib.machinery
import importlib.util
import os

import numpy as np
import pytest
import torch

import helpers as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu
_REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref")


def _load(name):
    path = os.path.join(_REF_DIR, name + ".so")
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (python oracle/build_grid_ref.py in the build container)")
    spec = importlib.util.spec_from_file_location(name, path, loader=importlib.machinery.ExtensionFileLoader(name, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def mods():
    from nerf_loam_amd import _lib, grid
    _lib.require_gpu()
    return dict(ref=_load("grid_ref"), ref_nc=_load("grid_ref_nc"), hip=grid)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def intersect_like_the_wrapper(mod, o, d, centres, structure, voxel, n_max=20):
    """SparseVoxelOctreeRayIntersect.forward (voxel_helpers.py:92-133): S = 1, G batch rows, rays padded with the first ones"""
    N = len(o)
    G = min(256, int(2 * 10 ** 9 / (centres.size + structure.size)))
    K = int(np.ceil(N / G)); Ht = K * G
    fill = np.arange(Ht) % N                              # the wrapper pads with the first rays (it needs N >= G / 2; cyclic here)
    rs = o[fill].reshape(G, K, 3); rd = d[fill].reshape(G, K, 3)
    pts = dev(centres)[None].expand(G, -1, -1).contiguous(); ch = dev(structure)[None].expand(G, -1, -1).contiguous()
    idx, t0, t1 = mod.svo_intersect(dev(rs), dev(rd), pts, ch, float(voxel), n_max)
    torch.cuda.synchronize()
    return [x.reshape(Ht, n_max)[:N].cpu().numpy() for x in (idx, t0, t1)]


def sample_like_the_wrapper(mod, idx, t0, t1, step_size, noise_seed):
    """ray_sample + InverseCDFRaySampling.forward (voxel_helpers.py:571-598, 262-347) with the noise tensor injected"""
    R, P = idx.shape
    dists = np.where(idx == -1, 0, t1 - t0).astype(np.float32)
    tot = np.zeros(R, np.float32)
    for h in range(P):
        tot = (tot + dists[:, h]).astype(np.float32)
    probs = (dists / tot[:, None]).astype(np.float32); steps = (tot / np.float32(step_size)).astype(np.float32)
    G = 200; L = int(np.ceil(R / G)); Ht = G * L
    pad = lambda a: np.concatenate([a, np.repeat(a[:1], Ht - R, 0)], 0)     # noqa: E731
    T = int(np.ceil(pad(steps)).max()) + P
    noise = O.hash_noise(noise_seed, np.arange(R), T)
    a = [pad(x).reshape(G, L, -1) for x in (idx, t0, t1, noise, probs)]
    st = pad(steps).reshape(G, L)
    outs = []
    for c0 in range(0, L, 800):
        args = [dev(x[:, c0:c0 + 800]) for x in a] + [dev(st[:, c0:c0 + 800])]
        outs.append(mod.inverse_cdf_sampling(*args, -1.0))
    torch.cuda.synchronize()
    s_idx, s_dep, s_dst = [torch.cat([r[i] for r in outs], 1).reshape(Ht, -1)[:R].cpu().numpy() for i in range(3)]
    S = int((s_idx != -1).sum(-1).max())
    return s_idx[:, :S], s_dep[:, :S], s_dst[:, :S]


def _ulp_close(a, b, scale, n_ulp=4):
    return np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= n_ulp * np.spacing(np.abs(scale).astype(np.float32)))


def _scenes():
    from nerf_loam_amd import synthetic as S
    import test_device_math_host as T
    out = {}
    for name, (nb, na, seed, voxel) in dict(maicity=(64, 48, 777, 0.2), kitti=(64, 48, 5, 0.3)).items():
        sc = H.build_oracle_scene(nb, na, seed, voxel=voxel)
        pose = np.array([2000.02, 1999.97, 2000.01, 0.004, -0.003, 0.01], np.float32)
        o, d = O.ray_setup(S.unit_dirs(sc["points"]), O.rodrigues(pose[3:]), pose[:3])
        out[name] = (o, d, sc["ms"].centres, sc["ms"].structure, voxel)
    # dense slab, grazing rays: > 20 voxels per ray -> the 20-hit cap in DFS order
    xs, ys, zs = np.meshgrid(np.arange(10000, 10048), np.arange(10000, 10040), np.arange(10000, 10003), indexing="ij")
    oc = O.Octree(); oc.init(256 * 256 * 4, 16, 0.2); oc.insert(np.stack([xs, ys, zs], -1).reshape(-1, 3).astype(np.int32))
    v, c, _ = oc.get_centres_and_children()
    centres, structure = O.grid_features(v, c, 0.2)
    rng = np.random.default_rng(5)
    origin = np.array([1999.0, 2003.7, 2000.31], np.float32)
    tgt = np.stack([rng.uniform(2000.0, 2009.6, 4096), rng.uniform(2000.0, 2008.0, 4096), rng.uniform(1998.5, 2002.0, 4096)], -1).astype(np.float32)
    d = tgt - origin; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    out["slab_cap"] = (np.broadcast_to(origin, d.shape).copy(), d, centres, structure, 0.2)
    # hand-built trees x degenerate rays (axis-parallel: inf / NaN slabs; origin inside; along faces / edges / corners)
    for kind, vox in dict(single=[[10000, 10000, 10000]],
                          block=[[10000 + i, 10000 + j, 10000 + k] for i in range(2) for j in range(2) for k in range(2)],
                          row30=[[10000 + i, 10000, 10000] for i in range(30)]).items():
        oc = O.Octree(); oc.init(256 * 256 * 4, 16, 0.2); oc.insert(np.asarray(vox, np.int32))
        v, c, _ = oc.get_centres_and_children()
        centres, structure = O.grid_features(v, c, 0.2)
        o, d = T._edge_rays()
        out["edge_" + kind] = (o, d, centres, structure, 0.2)
    return out


@pytest.fixture(scope="module")
def scenes():
    return _scenes()


@pytest.mark.parametrize("name", ["maicity", "kitti", "slab_cap", "edge_single", "edge_block", "edge_row30"])
def test_reference_svo_intersect_equals_the_restatement_and_the_product(mods, scenes, name):
    o, d, centres, structure, voxel = scenes[name]
    with np.errstate(all="ignore"):
        oi, o0, o1 = O.svo_intersect(o, d, centres, structure, voxel, 20)
    res = {k: intersect_like_the_wrapper(mods[k], o, d, centres, structure, voxel) for k in ("ref", "ref_nc", "hip")}
    for k, (idx, t0, t1) in res.items():
        assert np.array_equal(idx, oi), (name, k)
        live = oi != -1                                   # slots without a hit keep the caller's zero fill in every implementation
        assert np.array_equal(t0[live], o0[live]) and np.array_equal(t1[live], o1[live]), (name, k)
    if name == "slab_cap":
        assert ((oi != -1).sum(1) == 20).mean() > 0.1


@pytest.mark.parametrize("name,step", [("maicity", 0.1), ("maicity", 0.04), ("kitti", 0.15), ("slab_cap", 0.1)])
def test_reference_inverse_cdf_sampling_equals_the_restatement_and_the_product(mods, scenes, name, step):
    o, d, centres, structure, voxel = scenes[name]
    oi, o0, o1, hits = O.ray_intersect(o, d, centres, structure, voxel, 50.0)
    hr = np.nonzero(hits)[0]
    idx, t0, t1 = oi[hr], o0[hr], o1[hr]
    e_idx, e_dep, e_dst = O.ray_sample(idx, t0, t1, step, noise=lambda n: O.hash_noise(9, np.arange(len(hr)), n))
    valid = e_idx != -1
    res = {k: sample_like_the_wrapper(mods[k], idx, t0, t1, step, 9) for k in ("ref", "ref_nc", "hip")}
    for k, (s_idx, s_dep, s_dst) in res.items():
        assert s_idx.shape == e_idx.shape and np.array_equal(s_idx, e_idx), (name, k)
    for k in ("ref_nc", "hip"):                           # no FMA contraction: bit for bit
        assert np.array_equal(res[k][1][valid], e_dep[valid]), (name, k)
        assert np.array_equal(np.maximum(res[k][2][valid], 0), e_dst[valid]), (name, k)
    assert _ulp_close(res["ref"][1][valid], e_dep[valid], e_dep[valid])
    assert _ulp_close(np.maximum(res["ref"][2][valid], 0), e_dst[valid], e_dep[valid])


def test_reference_kernels_on_the_full_scan(mods):
    """the 64 x 2048 scan of the headline benchmark through the reference's kernels, the restatement and the product"""
    from nerf_loam_amd import synthetic as S
    pts, _ = S.synthetic_scan()
    pose = S.scan_pose()
    oc = O.Octree(); oc.init(256 * 256 * 4, 16, 0.2)
    oc.insert(S.voxel_coords(pts, np.eye(3, dtype=np.float32), pose[:3], 0.2))
    v, c, _ = oc.get_centres_and_children()
    centres, structure = O.grid_features(v, c, 0.2)
    o, d = O.ray_setup(S.unit_dirs(pts), O.rodrigues(pose[3:]), pose[:3])
    oi, o0, o1 = O.svo_intersect(o, d, centres, structure, 0.2, 20)
    for k in ("ref", "hip"):
        idx, t0, t1 = intersect_like_the_wrapper(mods[k], o, d, centres, structure, 0.2)
        live = oi != -1
        assert np.array_equal(idx, oi) and np.array_equal(t0[live], o0[live]) and np.array_equal(t1[live], o1[live]), k
    si, s0, s1, hits = O.ray_intersect(o, d, centres, structure, 0.2, 50.0)
    hr = np.nonzero(hits)[0]
    e_idx, e_dep, e_dst = O.ray_sample(si[hr], s0[hr], s1[hr], 0.1, noise=lambda n: O.hash_noise(9, np.arange(len(hr)), n))
    valid = e_idx != -1
    assert valid.sum() > 1_000_000
    for k in ("ref_nc", "hip", "ref"):
        s_idx, s_dep, s_dst = sample_like_the_wrapper(mods[k], si[hr], s0[hr], s1[hr], 0.1, 9)
        assert np.array_equal(s_idx, e_idx), k
        if k == "ref":
            assert _ulp_close(s_dep[valid], e_dep[valid], e_dep[valid]) and _ulp_close(np.maximum(s_dst[valid], 0), e_dst[valid], e_dep[valid])
        else:
            assert np.array_equal(s_dep[valid], e_dep[valid]) and np.array_equal(np.maximum(s_dst[valid], 0), e_dst[valid]), k
