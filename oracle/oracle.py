"""
oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of NeRF-LOAM's per-iteration SDF path (BASELINE.json `north_star`) used as the
parity checker for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; nerf_loam_amd/ never does.

Style: numpy fp32 with explicit operation order and CLOSED-FORM gradients (no autograd), so it is
an independent restatement of what the reference obtains through torch autograd.  The two CUDA
kernels and the octree are restated in C (oracle/nl_oracle.c, loaded with ctypes).

Reference map (paths under /root/reference):
  ray_setup            src/variations/render_helpers.py:366-381, 460-471
  rodrigues*           src/se3pose.py:18-35, 54-83
  svo_intersect        third_party/sparse_voxels/src/intersect_gpu.cu:193-272  (C)
  ray_intersect        src/variations/voxel_helpers.py:531-567
  ray_sample           src/variations/voxel_helpers.py:571-598 + :262-347 (batch layout G=200,
                       chunk 800) + third_party/sparse_voxels/src/sample_gpu.cu:133-239 (C)
  trilinear_*          src/variations/render_helpers.py:39-93
  decoder_*            src/variations/lidar.py:109-131 (cfg: 16-256-256-1 ReLU MLP)
  sdf_loss             src/criterion.py:16-115
  adam_step            torch/optim/adam.py::_single_tensor_adam (torch 2.10; third-party - the
                       reference pins "PyTorch 1.10", whose functional Adam uses mul_/add_ instead
                       of lerp_ for the first moment; the installed 2.10 semantics are restated)
  embedding tables     src/mapping.py:293-339

Pinning: tests/golden/*.npz hold outputs of the REFERENCE python code (imported from
/root/reference by tests/golden/make_golden.py, with the C kernels above plugged in as its
`grid` extension and oracle/_ref/svo_ref.so as its `svo`); tests/test_oracle_*.py check this
module against them.  The reference has no tests or golden vectors of its own.
"""
import ctypes
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_DEPTH_FILL = np.float32(80.0)          # voxel_helpers.py:24  MAX_DEPTH
N_MAX_HITS = 20                            # voxel_helpers.py:533 (hard-coded; cfg value ignored)
SAMPLER_G = 200                            # voxel_helpers.py:274
SAMPLER_CHUNK = 4 * SAMPLER_G              # voxel_helpers.py:304

f32 = np.float32


def lib():
    """Load (building if necessary) oracle/libnl_oracle.so."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libnl_oracle.so")
    src = os.path.join(_HERE, "nl_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libnl_oracle.so"], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(so)
    P = ctypes.c_void_p
    L.orc_svo_intersect.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int] + [P] * 7
    L.orc_svo_intersect.restype = None
    L.orc_inverse_cdf_sampling.argtypes = [ctypes.c_int] * 4 + [ctypes.c_float] + [P] * 9
    L.orc_inverse_cdf_sampling.restype = None
    L.orc_set_tail_always.argtypes = [ctypes.c_int]
    L.orc_octree_create.argtypes = [ctypes.c_int64]
    L.orc_octree_create.restype = P
    L.orc_octree_destroy.argtypes = [P]
    L.orc_octree_insert.argtypes = [P, P, ctypes.c_int64]
    L.orc_octree_count_nodes.argtypes = [P]
    L.orc_octree_count_nodes.restype = ctypes.c_int64
    L.orc_octree_count_leaf_nodes.argtypes = [P]
    L.orc_octree_count_leaf_nodes.restype = ctypes.c_int64
    L.orc_octree_export.argtypes = [P] * 4
    L.orc_morton_encode.argtypes = [ctypes.c_int] * 3
    L.orc_morton_encode.restype = ctypes.c_uint64
    L.orc_morton_decode.argtypes = [ctypes.c_uint64, P]
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# --------------------------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even, as torch's float->bfloat16 conversion)
# --------------------------------------------------------------------------------------------
def bf16_bits(x):
    x = _c(x, f32)
    u = x.view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    out = rounded.astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        out[nan] = 0x7FC0
    return out


def bf16_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(f32)


def bf16_round(x):
    return bf16_to_f32(bf16_bits(x))


# --------------------------------------------------------------------------------------------
# Octree (C restatement behind the reference's method names)
# --------------------------------------------------------------------------------------------
class Octree:
    """Mirrors torch.classes.svo.Octree (third_party/sparse_octree/src/bindings.cpp:4-31)."""

    def __init__(self):
        self._h = None

    def init(self, grid_dim, feat_dim, voxel_size):
        self._h = lib().orc_octree_create(int(grid_dim))
        self.feat_dim, self.voxel_size = feat_dim, voxel_size

    def insert(self, vox):
        vox = _c(vox, np.int32)
        assert vox.ndim == 2 and vox.shape[1] == 3
        lib().orc_octree_insert(self._h, _p(vox), vox.shape[0])

    def count_nodes(self):
        return int(lib().orc_octree_count_nodes(self._h))

    def count_leaf_nodes(self):
        return int(lib().orc_octree_count_leaf_nodes(self._h))

    def get_centres_and_children(self):
        n = self.count_nodes()
        voxels = np.zeros((n, 4), f32)
        children = -np.ones((n, 8), f32)
        features = -np.ones((n, 8), np.int32)
        lib().orc_octree_export(self._h, _p(voxels), _p(children), _p(features))
        return voxels, children, features

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_octree_destroy(self._h)
            self._h = None


def grid_features(voxels, children, voxel_size):
    """mapping.py:319-327: centres = (xyz + side/2) * voxel_size ; structure = [children, side]."""
    vs = f32(voxel_size)
    centres = ((voxels[:, :3] + voxels[:, 3:4] / f32(2)) * vs).astype(f32)
    structure = np.concatenate([children, voxels[:, 3:4]], 1).astype(np.int32)
    return centres, structure


def assign_embedding_rows(vertex_idx, id2row, next_row):
    """mapping.py:293-317: one NEW row per OCCURRENCE of a vertex id that has no row yet
    (duplicates inside one call all get rows; the last writer wins - numpy fancy assignment,
    like the reference's index assignment on one thread).  Returns the new row count."""
    flat = vertex_idx.reshape(-1).astype(np.int64)
    flat = flat[flat != -1]
    add = flat[id2row[flat] == -1]
    if add.size == 0:
        return next_row
    id2row[add] = np.arange(next_row, next_row + add.size, dtype=np.int32)
    return next_row + add.size


# --------------------------------------------------------------------------------------------
# SE3 (se3pose.py)
# --------------------------------------------------------------------------------------------
def _taylor(theta, which):
    """taylor_A (sin x / x) and taylor_B ((1-cos x)/x^2), 11 terms, fp32, same term order
    (se3pose.py:64-83).  Returns (value, derivative wrt theta)."""
    x = f32(theta)
    val, der = f32(0), f32(0)
    denom = 1.0
    for i in range(11):
        if which == "A":
            if i > 0:
                denom *= (2 * i) * (2 * i + 1)
        else:
            denom *= (2 * i + 1) * (2 * i + 2)
        sign = f32((-1) ** i)
        xp = f32(1) if i == 0 else f32(x ** f32(2 * i))
        val = f32(val + f32(sign * xp) / f32(denom))
        if i > 0:
            der = f32(der + f32(sign * f32(2 * i) * f32(x ** f32(2 * i - 1))) / f32(denom))
    return val, der


def skew(w):
    w0, w1, w2 = [f32(v) for v in w]
    z = f32(0)
    return np.array([[z, -w2, w1], [w2, z, -w0], [-w1, w0, z]], f32)


def rodrigues(w):
    """R = I + A(theta) [w]x + B(theta) [w]x^2     (se3pose.py:24-32)"""
    w = _c(w, f32)
    W = skew(w)
    theta = f32(np.sqrt(f32(f32(w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])))
    A, _ = _taylor(theta, "A")
    B, _ = _taylor(theta, "B")
    return (np.eye(3, dtype=f32) + A * W + B * (W @ W)).astype(f32)


def rodrigues_backward(w, G):
    """dL/dw given G = dL/dR (closed form of autograd through se3pose.py:24-32).
    theta = ||w|| has sub-gradient 0 at w = 0 (torch's norm backward)."""
    w = _c(w, f32)
    G = _c(G, f32)
    W = skew(w)
    W2 = (W @ W).astype(f32)
    theta = f32(np.sqrt(f32(f32(w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])))
    A, dA = _taylor(theta, "A")
    B, dB = _taylor(theta, "B")
    gA = f32((G * W).sum())
    gB = f32((G * W2).sum())
    gW = (A * G + B * (G @ W.T + W.T @ G)).astype(f32)
    gtheta = f32(gA * dA + gB * dB)
    gw = np.array([gW[2, 1] - gW[1, 2], gW[0, 2] - gW[2, 0], gW[1, 0] - gW[0, 1]], f32)
    if theta > 0:
        gw = (gw + gtheta * (w / theta)).astype(f32)
    return gw


def ray_setup(rays_d_sensor, R, t):
    """d_world = d_sensor @ R^T (fixed fp32 order), origin = t   (render_helpers.py:371-376)."""
    d = _c(rays_d_sensor, f32)
    R = _c(R, f32)
    out = np.empty_like(d)
    for i in range(3):
        out[:, i] = (d[:, 0] * R[i, 0] + d[:, 1] * R[i, 1]) + d[:, 2] * R[i, 2]
    o = np.broadcast_to(_c(t, f32).reshape(1, 3), d.shape).copy()
    return o, out


# --------------------------------------------------------------------------------------------
# intersect + post-processing
# --------------------------------------------------------------------------------------------
def svo_intersect(ray_start, ray_dir, centres, structure, voxel_size, n_max=N_MAX_HITS):
    """grid.svo_intersect on a flat ray list (the wrapper's G-way batching, padding and octree
    replication - voxel_helpers.py:97-108 - do not change any per-ray result)."""
    rs, rd = _c(ray_start, f32).reshape(-1, 3), _c(ray_dir, f32).reshape(-1, 3)
    pts, ch = _c(centres, f32), _c(structure, np.int32)
    m = rs.shape[0]
    idx = np.zeros((m, n_max), np.int32)
    t0 = np.zeros((m, n_max), f32)
    t1 = np.zeros((m, n_max), f32)
    lib().orc_svo_intersect(m, pts.shape[0], float(voxel_size), n_max, _p(rs), _p(rd), _p(pts), _p(ch),
                            _p(idx), _p(t0), _p(t1))
    return idx, t0, t1


def ray_intersect(ray_start, ray_dir, centres, structure, voxel_size, max_distance):
    """voxel_helpers.py:531-567.  Returns (idx, t0, t1) [N,H] sorted by t0 and culled, hits[N]."""
    idx, t0, t1 = svo_intersect(ray_start, ray_dir, centres, structure, voxel_size)
    md = f32(max_distance)
    inv = idx == -1
    t0[inv] = md
    t1[inv] = md
    order = np.argsort(t0, axis=-1, kind="stable")
    t0 = np.take_along_axis(t0, order, -1)
    t1 = np.take_along_axis(t1, order, -1)
    idx = np.take_along_axis(idx, order, -1)
    idx[t1 > f32(2) * md] = -1
    idx[t0 > md] = -1
    inv = idx == -1
    t0[inv] = md
    t1[inv] = md
    H = int((idx != -1).sum(-1).max()) if idx.size else 0
    idx, t0, t1 = idx[:, :H], t0[:, :H], t1[:, :H]
    hits = (idx != -1).any(-1)
    return idx, t0, t1, hits


# --------------------------------------------------------------------------------------------
# sampler
# --------------------------------------------------------------------------------------------
def hash_noise(seed, ray_ids, n_steps):
    """Counter-based U(0,1) noise shared bit-for-bit with the HIP sampler
    (nerf_loam_amd/csrc/nl_device_math.h::nl_noise): lowbias32 of (seed, ray id, step), top 24
    bits -> [0,1), clamped to [0.001, 0.999] like voxel_helpers.py:301."""
    ray = np.asarray(ray_ids, dtype=np.uint32).reshape(-1, 1)
    step = np.arange(n_steps, dtype=np.uint32).reshape(1, -1)
    with np.errstate(over="ignore"):
        x = np.uint32(seed) ^ (ray * np.uint32(0x9E3779B1)) ^ (step * np.uint32(0x85EBCA77))
        x = x ^ (x >> np.uint32(16))
        x = x * np.uint32(0x7FEB352D)
        x = x ^ (x >> np.uint32(15))
        x = x * np.uint32(0x846CA68B)
        x = x ^ (x >> np.uint32(16))
    u = (x >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)
    return np.clip(u, f32(0.001), f32(0.999)).astype(f32)


def ray_sample(idx, t0, t1, step_size, noise=None, tail_mode=0):
    """voxel_helpers.py:571-598 (ray_sample) + :262-347 (InverseCDFRaySampling.forward).

    idx/t0/t1: [R,P] for the HIT rays only.  noise: None -> 0.5 (the wrapper's deterministic
    mode), an [R, >=max_steps] array indexed by (hit-ray rank, step), or a callable n_steps -> [R, n_steps]
    (so a full scan does not have to materialise thousands of unused columns).
    tail_mode 0 = reference behaviour (position-dependent tail loop);
    tail_mode 1 = "fixed" extension: the tail loop always runs and tests the ray's own next hit.
    Returns (sampled_idx, sampled_depth, sampled_dists) [R,S] or None (the reference's guard)."""
    R, P = idx.shape
    inv = idx == -1
    dists = (t1 - t0).astype(f32)
    dists[inv] = 0
    tot = np.zeros(R, f32)
    for h in range(P):                       # sequential fp32 row sum
        tot = (tot + dists[:, h]).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        probs = (dists / tot[:, None]).astype(f32)
        steps = (tot / f32(step_size)).astype(f32)
    if tot.max() > 10 * MAX_DEPTH_FILL:
        return None

    G = SAMPLER_G
    L = int(np.ceil(R / G))
    Htot = L * G
    lib().orc_set_tail_always(1 if tail_mode == 1 else 0)

    def pad(a):
        if Htot > R:
            a = np.concatenate([a, np.repeat(a[:1], Htot - R, 0)], 0)
        return a

    idx_p, t0_p, t1_p, probs_p, steps_p = pad(idx), pad(t0), pad(t1), pad(probs), pad(steps)
    max_steps = int(np.ceil(steps_p).astype(np.int64).max()) + P
    if noise is None:
        nz = np.full((Htot, max_steps), 0.5, f32)
    else:
        nz = pad(_c(noise(max_steps) if callable(noise) else noise, f32)[:, :max_steps])
        assert nz.shape[1] == max_steps, "noise must cover max_steps columns"
    s_idx = -np.ones((Htot, max_steps), np.int32)
    s_dep = np.zeros((Htot, max_steps), f32)
    s_dst = np.zeros((Htot, max_steps), f32)

    # view as [G, L, *] and call the kernel on chunks of <=800 rays-per-row, like the wrapper
    def v3(a):
        return a.reshape(G, L, -1)

    for c0 in range(0, L, SAMPLER_CHUNK):
        c1 = min(L, c0 + SAMPLER_CHUNK)
        m = c1 - c0
        args = [np.ascontiguousarray(v3(a)[:, c0:c1]) for a in (idx_p, t0_p, t1_p, nz, probs_p)]
        st = np.ascontiguousarray(steps_p.reshape(G, L)[:, c0:c1])
        o_idx = -np.ones((G, m, max_steps), np.int32)
        o_dep = np.zeros((G, m, max_steps), f32)
        o_dst = np.zeros((G, m, max_steps), f32)
        lib().orc_inverse_cdf_sampling(G, m, P, max_steps, -1.0, _p(args[0]), _p(args[1]), _p(args[2]),
                                       _p(args[3]), _p(args[4]), _p(st), _p(o_idx), _p(o_dep), _p(o_dst))
        v3(s_idx)[:, c0:c1] = o_idx
        v3(s_dep)[:, c0:c1] = o_dep
        v3(s_dst)[:, c0:c1] = o_dst

    lib().orc_set_tail_always(0)
    s_idx, s_dep, s_dst = s_idx[:R], s_dep[:R], s_dst[:R]
    S = int((s_idx != -1).sum(-1).max())
    s_idx, s_dep, s_dst = s_idx[:, :S].copy(), s_dep[:, :S].copy(), s_dst[:, :S].copy()
    s_dst = np.maximum(s_dst, f32(0))
    bad = s_idx == -1
    s_dep[bad] = MAX_DEPTH_FILL
    s_dst[bad] = 0
    return s_idx, s_dep, s_dst


# --------------------------------------------------------------------------------------------
# gather + trilinear interpolation
# --------------------------------------------------------------------------------------------
_CORNER = np.array([[(k >> 2) & 1, (k >> 1) & 1, k & 1] for k in range(8)], np.int32)   # k = 4qx+2qy+qz


def trilinear_forward(xyz, vox, centres, vertex_rows, emb_bits, voxel_size):
    """render_helpers.py:63-93.  xyz [P,3] f32 sample positions, vox [P] voxel (node) ids,
    vertex_rows [n,8] embedding row of each corner vertex, emb_bits [E,C] bf16 bit patterns."""
    vs = f32(voxel_size)
    c = centres[vox]
    p = ((xyz - c) / vs + f32(0.5)).astype(f32)                     # [P,3]
    rows = vertex_rows[vox]                                         # [P,8]
    e = bf16_to_f32(emb_bits[rows])                                 # [P,8,C]
    one_m = (f32(1) - p).astype(f32)
    w = np.empty((p.shape[0], 8), f32)
    for k in range(8):
        tx = p[:, 0] if _CORNER[k, 0] else one_m[:, 0]
        ty = p[:, 1] if _CORNER[k, 1] else one_m[:, 1]
        tz = p[:, 2] if _CORNER[k, 2] else one_m[:, 2]
        w[:, k] = (tx * ty) * tz
    feats = np.zeros((p.shape[0], e.shape[2]), f32)
    for k in range(8):
        feats = (feats + w[:, k:k + 1] * e[:, k]).astype(f32)
    return feats, dict(p=p, one_m=one_m, w=w, e=e, rows=rows)


def trilinear_backward(dfeat, cache, voxel_size, n_rows, want_emb_grad=True, accumulate="fp32"):
    """Closed form of autograd through render_helpers.py:39-70.
    Embedding gradient: each per-(sample,corner) contribution w_k*dfeat is rounded to bf16 (autograd
    casts the gradient of the bf16 `point_feats` operand).  accumulate="fp32": contributions are
    summed in fp32 and the row sum is rounded to bf16 once - what torch's CUDA
    embedding_dense_backward does (the reference runs on a GPU) and what the HIP path implements.
    accumulate="bf16_seq": every add is rounded to bf16, in (sample, corner) order - what torch's CPU
    embedding_dense_backward (bf16 axpy) does; used only to pin this module against the goldens,
    which come from the reference running on CPU."""
    vs = f32(voxel_size)
    p, one_m, w, e, rows = cache["p"], cache["one_m"], cache["w"], cache["e"], cache["rows"]
    P, C = dfeat.shape
    gE = None
    if want_emb_grad and accumulate == "fp32":
        acc = np.zeros((n_rows, C), np.float64)
        for k in range(8):
            contrib = bf16_round((w[:, k:k + 1] * dfeat).astype(f32))
            np.add.at(acc, rows[:, k], contrib.astype(np.float64))
        gE = bf16_bits(acc.astype(f32))
    elif want_emb_grad:
        acc = np.zeros((n_rows, C), f32)
        contrib = bf16_round((w[:, :, None] * dfeat[:, None, :]).astype(f32))      # [P,8,C]
        flat_rows = rows.reshape(-1)
        flat_c = contrib.reshape(-1, C)
        # sequential per row; process "rounds" so that each round touches a row at most once
        order = np.argsort(flat_rows, kind="stable")
        sr = flat_rows[order]
        start = np.r_[0, np.nonzero(np.diff(sr))[0] + 1]
        rank = np.arange(len(sr)) - np.repeat(start, np.diff(np.r_[start, len(sr)]))
        for r in range(int(rank.max()) + 1 if len(rank) else 0):
            sel = order[rank == r]
            acc[flat_rows[sel]] = bf16_round((acc[flat_rows[sel]] + flat_c[sel]).astype(f32))
        gE = bf16_bits(acc)
    dot = np.einsum("pkc,pc->pk", e, dfeat).astype(f32)             # <e_k, dfeat>
    dp = np.zeros((P, 3), f32)
    for k in range(8):
        tx = p[:, 0] if _CORNER[k, 0] else one_m[:, 0]
        ty = p[:, 1] if _CORNER[k, 1] else one_m[:, 1]
        tz = p[:, 2] if _CORNER[k, 2] else one_m[:, 2]
        sx = f32(1) if _CORNER[k, 0] else f32(-1)
        sy = f32(1) if _CORNER[k, 1] else f32(-1)
        sz = f32(1) if _CORNER[k, 2] else f32(-1)
        dp[:, 0] += sx * (ty * tz) * dot[:, k]
        dp[:, 1] += sy * (tx * tz) * dot[:, k]
        dp[:, 2] += sz * (tx * ty) * dot[:, k]
    dxyz = (dp / vs).astype(f32)
    return gE, dxyz


# --------------------------------------------------------------------------------------------
# decoder MLP  16 -> 256 -> 256 -> 1
# --------------------------------------------------------------------------------------------
@dataclass
class DecoderParams:
    W1: np.ndarray
    b1: np.ndarray
    W2: np.ndarray
    b2: np.ndarray
    W3: np.ndarray
    b3: np.ndarray

    def names(self):
        return ["W1", "b1", "W2", "b2", "W3", "b3"]

    def copy(self):
        return DecoderParams(*[getattr(self, n).copy() for n in self.names()])


def decoder_init(seed, in_dim=16, width=256):
    """nn.Linear default init (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias)."""
    rng = np.random.default_rng(seed)

    def lin(o, i):
        k = 1.0 / np.sqrt(i)
        return rng.uniform(-k, k, (o, i)).astype(f32), rng.uniform(-k, k, (o,)).astype(f32)

    W1, b1 = lin(width, in_dim)
    W2, b2 = lin(width, width)
    W3, b3 = lin(1, width)
    return DecoderParams(W1, b1, W2, b2, W3, b3)


def _mm(a, b):
    """fp32 GEMM on the torch-CPU BLAS threads; transposed VIEWS go in as they are (no 1 GB transpose copies on a full scan)"""
    import torch

    def t(x):
        return torch.from_numpy(x if all(st > 0 for st in x.strides) else np.ascontiguousarray(x))
    return (t(a) @ t(b)).numpy()


def _linear_relu(a, W, b):
    """relu(a W^T + b) in fp32: the GEMM, then one fp32 add and one max per element - on the torch-CPU threads, in place (numpy's
    element-wise passes over [P,256] arrays are single-threaded and were a third of the CPU baseline's iteration)"""
    import torch
    t = torch.from_numpy(a if all(st > 0 for st in a.strides) else np.ascontiguousarray(a)) @ torch.from_numpy(np.ascontiguousarray(W.T))
    return t.add_(torch.from_numpy(np.ascontiguousarray(b, dtype=f32))).clamp_(min=0).numpy()


def decoder_forward(x, dp):
    h1 = _linear_relu(x, dp.W1, dp.b1)
    h2 = _linear_relu(h1, dp.W2, dp.b2)
    s = (_mm(h2, dp.W3.T) + dp.b3)[:, 0]
    return s.astype(f32), dict(x=x, h1=h1, h2=h2)


def decoder_backward(ds, cache, dp, want_wgrad=True):
    x, h1, h2 = cache["x"], cache["h1"], cache["h2"]
    ds2 = ds.reshape(-1, 1).astype(f32)
    g = {}
    import torch
    dh2 = torch.from_numpy(_mm(ds2, dp.W3)).mul_(torch.from_numpy(h2) > 0).numpy()        # (exact: a product with 0 / 1)
    dh1 = torch.from_numpy(_mm(dh2, dp.W2)).mul_(torch.from_numpy(h1) > 0).numpy()
    dx = _mm(dh1, dp.W1)
    if want_wgrad:
        # sums over the samples: fp32 GEMMs / row sums per chunk of 32 768 samples, the chunks combined in fp64 - a plain fp32
        # reduction over a full scan's 1.1 M samples (numpy adds rows sequentially along axis 0) carries ~1e-4 of round-off,
        # more than the kernels under test
        acc = {k: 0.0 for k in ("W3", "b3", "W2", "b2", "W1", "b1")}
        for c0 in range(0, len(ds2), 32768):
            sl = slice(c0, c0 + 32768)
            acc["W3"] = acc["W3"] + _mm(ds2[sl].T, h2[sl]).astype(np.float64)
            acc["b3"] = acc["b3"] + ds2[sl].sum(0, dtype=np.float64)
            acc["W2"] = acc["W2"] + _mm(dh2[sl].T, h1[sl]).astype(np.float64)
            acc["b2"] = acc["b2"] + dh2[sl].sum(0, dtype=np.float64)
            acc["W1"] = acc["W1"] + _mm(dh1[sl].T, x[sl]).astype(np.float64)
            acc["b1"] = acc["b1"] + dh1[sl].sum(0, dtype=np.float64)
        g = {k: np.asarray(v).astype(f32) for k, v in acc.items()}
    return dx.astype(f32), g


# --------------------------------------------------------------------------------------------
# loss  (criterion.py)
# --------------------------------------------------------------------------------------------
@dataclass
class LossCfg:
    truncation: float = 0.30
    sdf_weight: float = 10000.0
    fs_weight: float = 1.0
    max_depth: float = 50.0


def sdf_loss(z_vals, sdf, valid, gt_points, cos, cfg: LossCfg):
    """criterion.py:16-115 (l2, no eikonal).  z_vals/sdf/valid [R,S]; gt_points [R,3]; cos [R].
    Returns loss (fp32 scalar), dL/dsdf [R,S] and the integer normalisers."""
    tau = f32(cfg.truncation)
    gp = _c(gt_points, f32)
    d = (np.sqrt((gp[:, 0] * gp[:, 0] + gp[:, 1] * gp[:, 1] + gp[:, 2] * gp[:, 2]).astype(f32)) * cos).astype(f32)
    z = (z_vals * cos[:, None]).astype(f32)
    D = d[:, None]
    front = (z < (D - tau)).astype(f32)
    back = (z > (D + tau)).astype(f32)
    dmask = ((D > 0) & (D < f32(cfg.max_depth))).astype(f32)
    sdfm = ((f32(1) - front) * (f32(1) - back) * dmask).astype(f32)
    n_fs = f32(np.count_nonzero(front))
    n_sdf = f32(np.count_nonzero(sdfm))
    n_tot = f32(n_sdf + n_fs)
    with np.errstate(divide="ignore", invalid="ignore"):
        w_fs = f32(f32(1) - n_fs / n_tot)
        w_sdf = f32(f32(1) - n_sdf / n_tot)
    v = valid.astype(f32)
    N = f32(z.size)
    r_fs = (sdf * front * v - front).astype(f32)
    r_sdf = ((z + sdf * tau) * sdfm * v - D * sdfm).astype(f32)
    fs_loss = f32(np.mean((r_fs * r_fs).astype(f32), dtype=np.float64)) * w_fs
    sd_loss = f32(np.mean((r_sdf * r_sdf).astype(f32), dtype=np.float64)) * w_sdf
    loss = f32(f32(cfg.fs_weight) * fs_loss + f32(cfg.sdf_weight) * sd_loss)
    two_n = f32(2) / N
    dsdf = (f32(cfg.fs_weight) * w_fs * two_n * r_fs * front * v
            + f32(cfg.sdf_weight) * w_sdf * two_n * r_sdf * tau * sdfm * v).astype(f32)
    stats = dict(n_fs=int(n_fs), n_sdf=int(n_sdf), N=int(z.size), w_fs=w_fs, w_sdf=w_sdf,
                 fs_loss=fs_loss, sdf_loss=sd_loss)
    return loss, dsdf, stats


# --------------------------------------------------------------------------------------------
# Adam (torch 2.10 _single_tensor_adam, default branch)
# --------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, bf16=False, beta1=0.9, beta2=0.999, eps=1e-8):
    """One step; `step` is the 1-based step count.  All arrays fp32; with bf16=True every one of
    the seven element-wise results is rounded to the bf16 grid (the tensors ARE bf16 in torch)."""
    rnd = bf16_round if bf16 else (lambda a: a.astype(f32))
    w1 = f32(1 - beta1)
    # exp_avg.lerp_(grad, 1-beta1): weight < 0.5 -> self + weight * (end - self)
    m = rnd((m + w1 * (g - m)).astype(f32))
    v = rnd((v * f32(beta2)).astype(f32))
    v = rnd((v + (f32(1 - beta2) * g) * g).astype(f32))
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    den = rnd(np.sqrt(v).astype(f32))
    den = rnd((den / f32(bc2_sqrt)).astype(f32))
    den = rnd((den + f32(eps)).astype(f32))
    p = rnd((p + (f32(-step_size) * m) / den).astype(f32))
    return p, m, v


# --------------------------------------------------------------------------------------------
# whole iteration
# --------------------------------------------------------------------------------------------
@dataclass
class MapState:
    """Device-independent restatement of the reference's `map_states` dict (mapping.py:319-339)
    with the 2e9-row id table replaced by a per-node row table."""
    centres: np.ndarray            # [n,3] f32     voxel_center_xyz
    structure: np.ndarray          # [n,9] i32     voxel_structure
    vertex_idx: np.ndarray         # [n,8] i32     voxel_vertex_idx (node ids of the corner leaves)
    id2row: np.ndarray             # [n]   i32     voxel_id2embedding_id restricted to node ids
    emb: np.ndarray                # [E,C] uint16  voxel_vertex_emb (bf16 bit patterns)
    voxel_size: float = 0.2

    def vertex_rows(self):
        vr = np.where(self.vertex_idx >= 0, self.id2row[np.maximum(self.vertex_idx, 0)], -1)
        return vr.astype(np.int32)


@dataclass
class Frame:
    rays_d: np.ndarray             # [N,3] unit directions in the sensor frame (lidarFrame.py:47-52)
    points: np.ndarray             # [N,3] sensor-frame returns
    cos: np.ndarray                # [N]   pointsCos
    pose: np.ndarray               # [6]   (t, w)   se3pose.OptimizablePose.data
    optimize_pose: bool = True


@dataclass
class IterCfg:
    step_size: float = 0.1         # metres (mapper: 0.5*voxel, tracker: 0.2*voxel)
    max_distance: float = 50.0
    loss: LossCfg = field(default_factory=LossCfg)
    tail_mode: int = 0
    noise_seed: int = 777          # None -> deterministic 0.5


def render_and_grad(ms: MapState, dec: DecoderParams, frames, cfg: IterCfg,
                    want_emb_grad=True, want_dec_grad=True, ray_id_base=0, emb_accumulate="fp32", eval_rays=None):
    """One forward+backward of the reference iteration (render_helpers.py:356-423 minus the
    optimiser): returns a dict with every intermediate the parity tests compare, or None when the
    reference would skip the iteration.

    eval_rays (checker shortcut, not reference behaviour): boolean mask over ALL rays.  Geometry (hits, samples, loss
    masks and normalisers) is still computed for the whole ray set - the sampler's tail loop depends on a ray's position
    in the batch, SURVEY B5 - but the field / decoder are evaluated only on the samples of the marked rays.  Per-sample
    outputs of those rays (sdf, dsdf, feats, dfeat, dxyz) are exactly those of the full run; the loss value and the
    parameter gradients are sums over the marked rays only and are returned as None / partial."""
    o_l, d_l, Rs = [], [], []
    for fr in frames:
        R = rodrigues(fr.pose[3:])
        o, d = ray_setup(fr.rays_d, R, fr.pose[:3])
        o_l.append(o)
        d_l.append(d)
        Rs.append(R)
    rays_o, rays_d = np.concatenate(o_l), np.concatenate(d_l)
    gt = np.concatenate([fr.points for fr in frames]).astype(f32)
    cos = np.concatenate([fr.cos for fr in frames]).astype(f32)
    frame_of = np.concatenate([np.full(len(fr.rays_d), i, np.int32) for i, fr in enumerate(frames)])
    ds_all = np.concatenate([fr.rays_d for fr in frames]).astype(f32)

    idx, t0, t1, hits = ray_intersect(rays_o, rays_d, ms.centres, ms.structure, ms.voxel_size, cfg.max_distance)
    if hits.sum() <= 0:
        return None
    hr = np.nonzero(hits)[0]
    R_hit = len(hr)
    noise = None
    if cfg.noise_seed is not None:
        noise = lambda n_steps: hash_noise(cfg.noise_seed, hr + ray_id_base, n_steps)    # noqa: E731
    smp = ray_sample(idx[hr], t0[hr], t1[hr], cfg.step_size, noise=noise, tail_mode=cfg.tail_mode)
    if smp is None:
        return None
    s_idx, s_dep, s_dst = smp
    mask = s_idx != -1
    if mask.sum() == 0:
        return None
    S = s_idx.shape[1]
    o_h, d_h = rays_o[hr], rays_d[hr]
    rr, ss = np.nonzero(mask if eval_rays is None else mask & np.asarray(eval_rays, bool)[hr][:, None])
    depth = s_dep[rr, ss]
    xyz = (o_h[rr] + d_h[rr] * depth[:, None]).astype(f32)
    vox = s_idx[rr, ss].astype(np.int64)
    vrows = ms.vertex_rows()
    feats, tcache = trilinear_forward(xyz, vox, ms.centres, vrows, ms.emb, ms.voxel_size)
    sdf_p, dcache = decoder_forward(feats, dec)
    sdf = np.ones((R_hit, S), f32)
    sdf[rr, ss] = sdf_p
    loss, dsdf, stats = sdf_loss(s_dep, sdf, mask, gt[hr], cos[hr], cfg.loss)
    ds_p = dsdf[rr, ss]
    dfeat, gdec = decoder_backward(ds_p, dcache, dec, want_wgrad=want_dec_grad)
    gE, dxyz = trilinear_backward(dfeat, tcache, ms.voxel_size, ms.emb.shape[0], want_emb_grad, emb_accumulate)

    # pose gradients: dt = sum dxyz ; dR = sum depth * dxyz (x) d_sensor ; then Rodrigues tail
    pose_grads = []
    fr_of_sample = frame_of[hr][rr]
    dsn = ds_all[hr][rr]
    for i, fr in enumerate(frames):
        sel = fr_of_sample == i
        gt_ = dxyz[sel].astype(np.float64).sum(0).astype(f32)
        G = np.einsum("p,pi,pj->ij", depth[sel].astype(np.float64), dxyz[sel].astype(np.float64),
                      dsn[sel].astype(np.float64)).astype(f32)
        gw = rodrigues_backward(fr.pose[3:], G)
        pose_grads.append(np.concatenate([gt_, gw]).astype(f32))

    return dict(rays_o=rays_o, rays_d=rays_d, hits=hits, hit_idx=idx, hit_t0=t0, hit_t1=t1,
                s_idx=s_idx, z_vals=s_dep, s_dists=s_dst, valid=mask, sdf=sdf, loss=loss, dsdf=dsdf,
                stats=stats, feats=feats, xyz=xyz, vox=vox, dfeat=dfeat, dxyz=dxyz,
                grad_emb=gE, grad_dec=gdec, grad_pose=pose_grads, n_samples=int(mask.sum()),
                sample_ray=hr[rr], sample_slot=ss, partial=eval_rays is not None)


@dataclass
class AdamState:
    step: int = 0
    m_emb: np.ndarray = None
    v_emb: np.ndarray = None
    m_dec: dict = None
    v_dec: dict = None
    m_pose: list = None
    v_pose: list = None


def optimiser_step(ms: MapState, dec: DecoderParams, frames, out, st: AdamState, lrs,
                   update_decoder=True, update_pose=True):
    """torch.optim.Adam.step over {embeddings (bf16), decoder (fp32), poses (fp32)} with the
    param groups of render_helpers.py:341-353.  Mutates ms.emb, dec, frames[i].pose."""
    st.step += 1
    E, C = ms.emb.shape
    if st.m_emb is None:
        st.m_emb = np.zeros((E, C), f32)
        st.v_emb = np.zeros((E, C), f32)
        st.m_dec = {n: np.zeros_like(getattr(dec, n)) for n in dec.names()}
        st.v_dec = {n: np.zeros_like(getattr(dec, n)) for n in dec.names()}
        st.m_pose = [np.zeros(6, f32) for _ in frames]
        st.v_pose = [np.zeros(6, f32) for _ in frames]
    p = bf16_to_f32(ms.emb)
    g = bf16_to_f32(out["grad_emb"])
    p, st.m_emb, st.v_emb = adam_step(p, g, st.m_emb, st.v_emb, st.step, lrs[0], bf16=True)
    ms.emb = bf16_bits(p)
    if update_decoder:
        for n in dec.names():
            gp = out["grad_dec"][n].reshape(getattr(dec, n).shape)
            q, st.m_dec[n], st.v_dec[n] = adam_step(getattr(dec, n), gp, st.m_dec[n], st.v_dec[n], st.step, lrs[1])
            setattr(dec, n, q)
    if update_pose:
        for i, fr in enumerate(frames):
            if fr.optimize_pose:
                fr.pose, st.m_pose[i], st.v_pose[i] = adam_step(fr.pose, out["grad_pose"][i], st.m_pose[i],
                                                                st.v_pose[i], st.step, lrs[2])


def select_rays(points, cos, pose, mask, optimize_pose=True):
    """Frame restricted to the boolean ray mask (rays keep dataset order, lidarFrame.py:55-57)."""
    from nerf_loam_amd.synthetic import unit_dirs
    return Frame(unit_dirs(points)[mask], points[mask], cos[mask], pose, optimize_pose)


def bundle_adjust(ms, dec, scans, masks, cfg, n_iter, lrs, update_pose=True, update_decoder=True,
                  emb_accumulate="fp32"):
    """render_helpers.py:321-425.  scans: list of dict(points, cos, pose[6], index); masks[f][it].
    Mutates ms.emb / dec / scan poses.  Returns the per-iteration outputs."""
    st = AdamState()
    outs = []
    for it in range(n_iter):
        frames = [select_rays(sc["points"], sc["cos"], sc["pose"], masks[f][it],
                              optimize_pose=(sc["index"] != 0 and update_pose))
                  for f, sc in enumerate(scans)]
        out = render_and_grad(ms, dec, frames, cfg, True, update_decoder, emb_accumulate=emb_accumulate)
        outs.append(out)
        if out is None:
            continue
        optimiser_step(ms, dec, frames, out, st, lrs, update_decoder, update_pose)
        for sc, fr in zip(scans, frames):
            sc["pose"] = fr.pose
    return outs


def track(ms, dec, scan, masks, cfg, n_iter, lr):
    """render_helpers.py:428-514: pose-only Adam; embeddings and decoder are constants."""
    m, v = np.zeros(6, f32), np.zeros(6, f32)
    pose = scan["pose"].copy()
    outs = []
    for it in range(n_iter):
        fr = select_rays(scan["points"], scan["cos"], pose, masks[it])
        out = render_and_grad(ms, dec, [fr], cfg, want_emb_grad=False, want_dec_grad=False)
        outs.append(out)
        if out is None:
            break
        pose, m, v = adam_step(pose, out["grad_pose"][0], m, v, it + 1, lr)
    return pose, outs
