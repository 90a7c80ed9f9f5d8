"""CPU: the per-lane device functions of the HIP kernels (nerf_loam_amd/csrc/nl_device_math.h),
compiled for the host by tests/host_harness.cpp, against the oracle.  Integer/index outputs and
IEEE fp32 geometry must be bit-identical."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hh():
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "host_harness.so")
    src = os.path.join(HERE, "host_harness.cpp")
    hdr = os.path.join(HERE, "..", "nerf_loam_amd", "csrc", "nl_device_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", so])
    return ctypes.CDLL(so)


def p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def scene():
    sc = H.build_oracle_scene(64, 48, 777)
    rng = np.random.default_rng(3)
    pts, cos = sc["points"], sc["cos"]
    pose = np.array([2000.02, 1999.97, 2000.01, 0.004, -0.003, 0.01], np.float32)
    o, d = O.ray_setup(__import__("nerf_loam_amd.synthetic", fromlist=["x"]).unit_dirs(pts), O.rodrigues(pose[3:]), pose[:3])
    return dict(sc=sc, o=o, d=d, cos=cos, pts=pts)


def test_intersect_sort_cull_bit_exact(hh, scene):
    ms = scene["sc"]["ms"]
    o, d = scene["o"], scene["d"]
    N = len(o)
    idx = np.zeros((N, 20), np.int32); t0 = np.zeros((N, 20), np.float32); t1 = np.zeros((N, 20), np.float32)
    cnt = np.zeros(N, np.int32)
    hh.hh_ray_intersect(N, p(o), p(d), p(ms.centres), p(ms.structure), ctypes.c_float(0.2), ctypes.c_float(50.0),
                        p(idx), p(t0), p(t1), p(cnt))
    oi, o0, o1, hits = O.ray_intersect(o, d, ms.centres, ms.structure, 0.2, 50.0)
    Hm = oi.shape[1]
    assert cnt.max() == Hm
    assert np.array_equal(idx[:, :Hm], oi)
    assert np.array_equal(t0[:, :Hm], o0) and np.array_equal(t1[:, :Hm], o1)
    assert (idx[:, Hm:] == -1).all()
    assert np.array_equal(cnt > 0, hits)


def test_raw_intersect_matches_reference_kernel_restatement(hh, scene):
    ms = scene["sc"]["ms"]
    o, d = scene["o"][:500], scene["d"][:500]
    for n_max in (20, 3):
        idx = np.zeros((500, n_max), np.int32); t0 = np.zeros((500, n_max), np.float32); t1 = np.zeros((500, n_max), np.float32)
        hh.hh_svo_intersect_raw(500, p(o), p(d), p(ms.centres), p(ms.structure), ctypes.c_float(0.2), n_max, p(idx), p(t0), p(t1))
        oi, o0, o1 = O.svo_intersect(o, d, ms.centres, ms.structure, 0.2, n_max)
        assert np.array_equal(idx, oi) and np.array_equal(t0, o0) and np.array_equal(t1, o1)


@pytest.mark.parametrize("form", [0, 2])                     # 0: sequential walk, 2: step-parallel formulation (nl_sample_walk_steps)
@pytest.mark.parametrize("tail_mode,use_hash,step", [(0, 1, 0.1), (0, 0, 0.04), (1, 1, 0.1), (0, 1, 0.04)])
def test_sampler_bit_exact(hh, scene, tail_mode, use_hash, step, form):
    ms = scene["sc"]["ms"]
    o, d = scene["o"], scene["d"]
    oi, o0, o1, hits = O.ray_intersect(o, d, ms.centres, ms.structure, 0.2, 50.0)
    hr = np.nonzero(hits)[0]
    R, P = len(hr), oi.shape[1]
    noise = O.hash_noise(777, hr, 4096) if use_hash else None
    s_idx, s_dep, s_dst = O.ray_sample(oi[hr], o0[hr], o1[hr], step, noise=noise, tail_mode=tail_mode)
    S = s_idx.shape[1]
    pad = lambda a, fill: np.ascontiguousarray(np.concatenate([a[hr], np.full((R, 20 - P), fill, a.dtype)], 1))
    hi, h0, h1 = pad(oi, -1), pad(o0, np.float32(50)), pad(o1, np.float32(50))
    cap = S + 8
    g_idx = -np.ones((R, cap), np.int32); g_dep = np.full((R, cap), 80, np.float32); g_dst = np.zeros((R, cap), np.float32)
    cnt = np.zeros(R, np.int32)
    ids = hr.astype(np.uint32)
    hh.hh_sample(R, p(hi), p(h0), p(h1), P, ctypes.c_float(step), 777, use_hash, tail_mode | form, p(ids), cap,
                 p(g_idx), p(g_dep), p(g_dst), p(cnt))
    assert cnt.max() == S
    assert np.array_equal(g_idx[:, :S], s_idx)
    assert np.array_equal(g_dep[:, :S], s_dep)
    assert np.array_equal(g_dst[:, :S], s_dst)


def test_noise_hash(hh):
    rays = np.array([0, 1, 77, 131071, 2 ** 31 + 5], np.uint32)
    out = np.zeros((len(rays), 64), np.float32)
    hh.hh_noise(777, len(rays), p(rays), 64, p(out))
    assert np.array_equal(out, O.hash_noise(777, rays, 64))
    assert out.min() >= 0.001 and out.max() <= 0.999


def test_adam_bf16_bit_exact_vs_torch_golden(hh, golden_dir):
    g = np.load(os.path.join(golden_dir, "adam.npz"))
    pbits = O.bf16_bits(g["p0"]).reshape(-1).copy()
    m = np.zeros_like(pbits); v = np.zeros_like(pbits)
    for t, grad in enumerate(g["gs"]):
        gb = O.bf16_bits(grad).reshape(-1)
        hh.hh_adam_bf16(pbits.size, p(pbits), p(gb), p(m), p(v), ctypes.c_double(0.03), t + 1)
        assert np.array_equal(pbits, O.bf16_bits(g["p_bf16"][t]).reshape(-1)), f"step {t}"


def test_adam_f32_vs_torch_golden(hh, golden_dir):
    g = np.load(os.path.join(golden_dir, "adam.npz"))
    pv = g["p0"].reshape(-1).copy(); m = np.zeros_like(pv); v = np.zeros_like(pv)
    for t, grad in enumerate(g["gs"]):
        gg = np.ascontiguousarray(grad.reshape(-1))
        hh.hh_adam_f32(pv.size, p(pv), p(gg), p(m), p(v), ctypes.c_double(0.03), t + 1)
        np.testing.assert_allclose(pv, g["p_f32"][t].reshape(-1), rtol=1e-5, atol=1e-8)


def test_se3_vs_reference_golden(hh, golden_dir):
    g = np.load(os.path.join(golden_dir, "se3.npz"))
    for w, G, R, gw in zip(g["w"], g["G"], g["R"], g["gw"]):
        Ro = np.zeros(9, np.float32); go = np.zeros(3, np.float32)
        hh.hh_rodrigues(p(np.ascontiguousarray(w)), p(Ro))
        hh.hh_rodrigues_bwd(p(np.ascontiguousarray(w)), p(np.ascontiguousarray(G.reshape(-1))), p(go))
        np.testing.assert_allclose(Ro.reshape(3, 3), R, rtol=0, atol=3e-7)
        np.testing.assert_allclose(go, gw, rtol=3e-6, atol=3e-7)


def test_trilinear_weights_and_dp(hh):
    rng = np.random.default_rng(0)
    P = 1000
    c = (rng.integers(9990, 10100, (P, 3)).astype(np.float32) + 0.5) * np.float32(0.2)
    x = (c + rng.uniform(-0.1, 0.1, (P, 3)).astype(np.float32)).astype(np.float32)
    pp = np.zeros((P, 3), np.float32); w = np.zeros((P, 8), np.float32)
    hh.hh_trilinear(P, p(x), p(c), ctypes.c_float(0.2), p(pp), p(w))
    emb = O.bf16_bits(rng.normal(0, 0.01, (8, 16)).astype(np.float32))
    feats, cache = O.trilinear_forward(x, np.arange(P) % 1, c[:1].repeat(1, 0) * 0 + c[:1], np.arange(8, dtype=np.int32)[None], emb, 0.2) if False else (None, None)
    po = ((x - c) / np.float32(0.2) + np.float32(0.5)).astype(np.float32)
    assert np.array_equal(pp, po)
    q = 1 - po
    for k in range(8):
        tx = po[:, 0] if k & 4 else q[:, 0]; ty = po[:, 1] if k & 2 else q[:, 1]; tz = po[:, 2] if k & 1 else q[:, 2]
        assert np.array_equal(w[:, k], (tx * ty) * tz)
    np.testing.assert_allclose(w.sum(1), 1.0, atol=1e-5)
    dot = rng.normal(size=(P, 8)).astype(np.float32)
    dp = np.zeros((P, 3), np.float32)
    hh.hh_trilinear_dp(P, p(pp), p(dot), p(dp))
    eps = 1e-3
    for a in range(3):                       # finite differences of sum_k w_k dot_k
        pa, pb = pp.copy(), pp.copy(); pa[:, a] += eps; pb[:, a] -= eps
        wa = np.zeros((P, 8), np.float32); wb = np.zeros((P, 8), np.float32)
        for pv, wv in ((pa, wa), (pb, wb)):
            qq = 1 - pv
            for k in range(8):
                wv[:, k] = (pv[:, 0] if k & 4 else qq[:, 0]) * (pv[:, 1] if k & 2 else qq[:, 1]) * (pv[:, 2] if k & 1 else qq[:, 2])
        fd = ((wa - wb) * dot).sum(1) / (2 * eps)
        np.testing.assert_allclose(dp[:, a], fd, rtol=2e-2, atol=2e-3)


def test_loss_gradient_matches_oracle(hh):
    rng = np.random.default_rng(1)
    R, S = 40, 12
    gt = rng.uniform(5, 30, (R, 3)).astype(np.float32); cos = rng.uniform(0.05, 1, R).astype(np.float32)
    d = (np.sqrt((gt * gt).sum(1, dtype=np.float32)) * cos).astype(np.float32)
    zv = (np.linalg.norm(gt, axis=1)[:, None] + rng.normal(0, 0.4, (R, S))).astype(np.float32)
    valid = rng.random((R, S)) < 0.8
    zv[~valid] = 80.0
    sdf = rng.normal(0, 0.5, (R, S)).astype(np.float32); sdf[~valid] = 1.0
    loss, dsdf, st = O.sdf_loss(zv, sdf, valid, gt, cos, O.LossCfg())
    rr, ss = np.nonzero(valid)
    z = (zv * cos[:, None]).astype(np.float32)[rr, ss]
    dd = np.ascontiguousarray(d[rr]); sp = np.ascontiguousarray(sdf[rr, ss]); z = np.ascontiguousarray(z)
    n = len(rr)
    ds = np.zeros(n, np.float32); fr = np.zeros(n, np.int32); sm = np.zeros(n, np.int32)
    two_n = np.float32(2) / np.float32(R * S)
    hh.hh_loss(n, p(sp), p(z), p(dd), ctypes.c_float(st["w_fs"]), ctypes.c_float(st["w_sdf"]), ctypes.c_float(two_n),
               ctypes.c_float(1.0), ctypes.c_float(10000.0), ctypes.c_float(0.3), ctypes.c_float(50.0), p(ds), p(fr), p(sm))
    np.testing.assert_allclose(ds, dsdf[rr, ss], rtol=1e-6, atol=1e-12)


# ------------------------------------------------------------------------------------------------
# edge cases of the reference kernels (SURVEY 4 list): hand-built octrees, degenerate rays
# ------------------------------------------------------------------------------------------------
def _tiny_tree(vox):
    oc = O.Octree(); oc.init(256 * 256 * 4, 16, 0.2); oc.insert(np.asarray(vox, np.int32))
    v, c, f = oc.get_centres_and_children()
    return O.grid_features(v, c, 0.2)


def _edge_rays():
    base = np.array([10000, 10000, 10000], np.float32) * np.float32(0.2)
    o, d = [], []
    c0 = base + np.float32(0.1)                                      # centre of voxel (10000,10000,10000)
    o.append(c0 + [-1.0, 0, 0]); d.append([1, 0, 0])               # axis-parallel through the centre (dy = dz = 0 -> inf/NaN slabs)
    o.append(c0 + [0, -1.0, 0]); d.append([0, 1, 0])
    o.append(c0.copy()); d.append([0.6, 0.64, 0.48])               # origin INSIDE the voxel (t_min = 0)
    o.append(c0 + [-1.0, 0.1, 0]); d.append([1, 0, 0])             # exactly along a voxel face (y = face plane)
    o.append(c0 + [-1.0, 0.1, 0.1]); d.append([1, 0, 0])           # exactly along a voxel edge
    o.append(c0 + [-1.0, -1.0, -1.0]); d.append(np.ones(3) / np.sqrt(3))     # through the diagonal / corners
    o.append(c0 + [-1.0, 0, 0]); d.append([-1, 0, 0])              # pointing away: no hit
    o.append(c0 + [-1.0, 5.0, 0]); d.append([1, 0, 0])             # parallel miss
    o.append(c0 + [-1.0, 0.05, 0.02]); d.append([1, 1e-4, -2e-4])  # grazing, nearly parallel
    o.append(c0 + [30.0, 0.03, 0.01]); d.append([-1, 0, 0])        # far origin, reversed direction
    o = np.array(o, np.float32); d = np.array(d, np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


@pytest.mark.parametrize("vox", [
    [[10000, 10000, 10000]],                                                         # single voxel
    [[10000 + i, 10000 + j, 10000 + k] for i in range(2) for j in range(2) for k in range(2)],   # 2x2x2 block
    [[10000 + i, 10000, 10000] for i in range(30)],                                  # a row of 30: > 20 hits along +x
])
def test_intersect_edge_cases_bit_exact(hh, vox):
    centres, structure = _tiny_tree(vox)
    o, d = _edge_rays()
    N = len(o)
    with np.errstate(all="ignore"):
        oi, o0, o1, hits = O.ray_intersect(o, d, centres, structure, 0.2, 50.0)
    idx = np.zeros((N, 20), np.int32); t0 = np.zeros((N, 20), np.float32); t1 = np.zeros((N, 20), np.float32); cnt = np.zeros(N, np.int32)
    hh.hh_ray_intersect(N, p(o), p(d), p(centres), p(structure), ctypes.c_float(0.2), ctypes.c_float(50.0), p(idx), p(t0), p(t1), p(cnt))
    Hm = oi.shape[1]
    assert np.array_equal(cnt > 0, hits)
    assert np.array_equal(idx[:, :Hm], oi) and np.array_equal(t0[:, :Hm], o0) and np.array_equal(t1[:, :Hm], o1)
    if len(vox) == 30:
        assert cnt.max() == 20                                                       # the hard-coded cap (B2)
    if len(vox) == 1:
        assert hits[0] and hits[2] and not hits[6] and not hits[7]
        assert o0[2, 0] == 0.0                                                       # origin inside: t_min clamps to 0


def test_sampler_edge_cases_bit_exact(hh):
    """zero-length intersections, a single hit, many hits, tiny and huge step counts"""
    md = np.float32(50)
    rows = [
        ([5], [1.0], [1.2]),                                         # one interval
        ([5, 6], [1.0, 1.2], [1.2, 1.2]),                            # zero-length second interval (grazing corner)
        ([5, 6, 7], [1.0, 1.0, 1.3], [1.0, 1.3, 1.35]),              # zero-length FIRST interval, tie in t_min
        (list(range(20)), list(np.arange(20) * 0.2 + 2), list(np.arange(20) * 0.2 + 2.2)),   # 20 contiguous intervals
        ([9], [3.0], [3.0000002]),                                    # one-ulp interval
        ([3, 4], [0.0, 0.5], [0.4, 7.5]),                             # origin inside + one long interval (many steps)
    ]
    R = len(rows)
    idx = -np.ones((R, 20), np.int32); t0 = np.full((R, 20), md, np.float32); t1 = np.full((R, 20), md, np.float32)
    for i, (a, b, c) in enumerate(rows):
        idx[i, :len(a)] = a; t0[i, :len(a)] = b; t1[i, :len(a)] = c
    P = 20
    for step, tail in ((0.1, 0), (0.04, 0), (0.1, 1)):
        noise = O.hash_noise(11, np.arange(R), 4096)
        with np.errstate(all="ignore"):
            s_idx, s_dep, s_dst = O.ray_sample(idx, t0, t1, step, noise=noise, tail_mode=tail)
        S = s_idx.shape[1]
        cap = S + 4
        g_idx = -np.ones((R, cap), np.int32); g_dep = np.full((R, cap), 80, np.float32); g_dst = np.zeros((R, cap), np.float32)
        cnt = np.zeros(R, np.int32)
        ids = np.arange(R, dtype=np.uint32)
        for form in (0, 2):                                           # sequential walk / step-parallel formulation
            g_idx[:] = -1; g_dep[:] = 80; g_dst[:] = 0; cnt[:] = 0
            hh.hh_sample(R, p(idx), p(t0), p(t1), P, ctypes.c_float(step), 11, 1, tail | form, p(ids), cap, p(g_idx), p(g_dep), p(g_dst), p(cnt))
            assert cnt.max() == S, form
            assert np.array_equal(g_idx[:, :S], s_idx) and np.array_equal(g_dep[:, :S], s_dep) and np.array_equal(g_dst[:, :S], s_dst), form
        valid = s_idx != -1
        assert (np.diff(np.where(valid, s_dep, np.inf), axis=1)[valid[:, 1:]] >= 0).all()      # depths monotone along a ray
        if tail == 1:                                                                   # "fixed" sampler covers every interval fully
            span = ((t1 - t0) * (idx != -1)).sum(1)
            np.testing.assert_allclose(s_dst.sum(1), span, rtol=0, atol=2e-5)


def test_three_term_bf16_split_is_exact(hh):
    """The arithmetic fact behind the decoder GEMMs on the bf16 matrix cores: nl_split3_bf16 writes an fp32
    value as three bf16 terms whose sum is the value EXACTLY, and any product of two bf16 terms is exact in fp32 - so the
    matrix core forms exact partial products and only the fp32 accumulation rounds, as in an fp32 GEMM."""
    rng = np.random.default_rng(0)
    v = np.concatenate([
        rng.normal(size=200000).astype(np.float32), (rng.normal(size=50000) * 1e-6).astype(np.float32),
        (rng.normal(size=50000) * 1e6).astype(np.float32), rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30, 3.4e38, 1.1754944e-38, np.float32(1) + np.float32(2 ** -23), 0.1, 1 / 3], np.float32)])
    v = v[np.isfinite(v) & ((np.abs(v) >= 2.0 ** -100) | (v == 0))]        # results far above the bf16 subnormal range, as in the kernels
    n = len(v)
    hi, mid, lo = (np.empty(n, np.uint16) for _ in range(3))
    hh.hh_split3_bf16(n, p(v), p(hi), p(mid), p(lo))
    f = lambda b: (b.astype(np.uint32) << 16).view(np.float32)
    H_, M_, L_ = f(hi).astype(np.float64), f(mid).astype(np.float64), f(lo).astype(np.float64)
    assert np.array_equal(H_ + M_ + L_, v.astype(np.float64))                 # exact (float64 holds the three-term sum exactly)
    assert (np.abs(M_) <= np.abs(H_) * 2.0 ** -7 + 0).all() and (np.abs(L_) <= np.abs(H_) * 2.0 ** -15).all()
    # a bf16 x bf16 product has a 16-bit significand: exact in fp32
    a, b = f(hi[:100000]), f(mid[100000:200000])
    assert np.array_equal((a.astype(np.float64) * b.astype(np.float64)).astype(np.float32).astype(np.float64), a.astype(np.float64) * b.astype(np.float64))
    # hence sum_k a_k b_k == sum_k sum_{p,q} a_k^(p) b_k^(q) exactly (checked in float64 on a 256-term dot product of fp32 values)
    x, w = rng.normal(size=256).astype(np.float32), rng.normal(size=256).astype(np.float32)
    parts = []
    for arr in (x, w):
        h_, m_, l_ = (np.empty(256, np.uint16) for _ in range(3))
        hh.hh_split3_bf16(256, p(arr), p(h_), p(m_), p(l_))
        parts.append([f(h_).astype(np.float64), f(m_).astype(np.float64), f(l_).astype(np.float64)])
    from fractions import Fraction
    exact = sum(Fraction(float(a_)) * Fraction(float(b_)) for a_, b_ in zip(x, w))
    nine = sum(Fraction(float(pa[k])) * Fraction(float(pb[k])) for pa in parts[0] for pb in parts[1] for k in range(256))
    assert exact == nine


def test_select_key_is_a_bijection_and_matches_the_numpy_restatement(hh):
    n = 1 << 20
    out = np.empty(n, np.uint32)
    hh.hh_select_key(n, 12345, p(out))
    assert len(np.unique(out)) == n
    x = np.arange(n, dtype=np.uint64) ^ np.uint64((12345 * 0x9E3779B9 + 0x7F4A7C15) & 0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    assert np.array_equal(out.astype(np.uint64), x)


def test_unit_dir_is_the_reference_frames_host_arithmetic(hh):
    """a1 (LidarFrame.get_rays, /root/reference/src/lidarFrame.py:47-52): nl_unit_dir - what nl_unit_dirs and the ray-selection kernels
    evaluate on the device - against the reference's two torch lines run here, bit for bit: scan-like points, tiny / huge / axis-aligned
    ones and the zero point (0 / 1e-8 = 0); and the generator-side restatement S.unit_dirs against both."""
    import torch
    from nerf_loam_amd import synthetic as S
    rng = np.random.default_rng(5)
    pts = np.concatenate([H.scene_points(64, 256, 3)[0], rng.normal(size=(200000, 3)).astype(np.float32) * 30,
                          rng.normal(size=(2000, 3)).astype(np.float32) * 1e-12, rng.normal(size=(2000, 3)).astype(np.float32) * 1e12,
                          np.eye(3, dtype=np.float32) * 7.5, np.zeros((1, 3), np.float32)]).astype(np.float32)
    t = torch.from_numpy(pts)
    ref_norm = torch.norm(t, 2, -1, keepdim=True) + 1e-8
    ref_d = (t / ref_norm).float().numpy()
    d, nrm = np.empty_like(pts), np.empty(len(pts), np.float32)
    hh.hh_unit_dirs(len(pts), p(pts), p(d), p(nrm))
    assert np.array_equal(d.view(np.uint32), ref_d.view(np.uint32))
    assert np.array_equal(nrm.view(np.uint32), ref_norm.numpy()[:, 0].view(np.uint32))
    assert np.array_equal(S.unit_dirs(pts).view(np.uint32), ref_d.view(np.uint32))
    assert np.array_equal(d[-1], np.zeros(3, np.float32)) and np.all(np.abs(np.linalg.norm(d[:200000].astype(np.float64), axis=1) - 1) < 1e-6)


def test_relu_mask_identities_behind_the_bf16_backward_gemms():
    """with dH2[i][j] = m(i,j) * dsdf_i * w3_j (m = 0/1 ReLU mask of H2),
        dH2 @ W2      == dsdf[:,None] * (m @ (w3[:,None] * W2))            (dgrad, gemm_mask_x)
        dH2.T @ H1    == w3[:,None] * (m.T @ (dsdf[:,None] * H1))           (dW2, k_decoder_wgrad2_x)
    exactly (checked in rational arithmetic on a small case, and to float64 round-off on the decoder's sizes)."""
    from fractions import Fraction as Fr
    rng = np.random.default_rng(4)
    n, w = 5, 6
    F = lambda a: np.vectorize(lambda x: Fr(float(x)))(a.astype(np.float32))
    m = (rng.random((n, w)) > 0.5).astype(np.float32); ds = rng.normal(size=n); w3 = rng.normal(size=w)
    W2 = rng.normal(size=(w, w)); H1 = np.maximum(rng.normal(size=(n, w)), 0)
    mF, dsF, w3F, W2F, H1F = F(m), F(ds), F(w3), F(W2), F(H1)
    dH2 = mF * dsF[:, None] * w3F[None, :]
    assert (dH2.dot(W2F) == dsF[:, None] * mF.dot(w3F[:, None] * W2F)).all()
    assert (dH2.T.dot(H1F) == w3F[:, None] * mF.T.dot(dsF[:, None] * H1F)).all()
    n, w = 64, 256
    m = (rng.random((n, w)) > 0.5).astype(np.float64); ds = rng.normal(size=n); w3 = rng.normal(size=w)
    W2 = rng.normal(size=(w, w)); H1 = np.maximum(rng.normal(size=(n, w)), 0)
    dH2 = m * ds[:, None] * w3[None, :]
    np.testing.assert_allclose(dH2 @ W2, ds[:, None] * (m @ (w3[:, None] * W2)), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dH2.T @ H1, w3[:, None] * (m.T @ (ds[:, None] * H1)), rtol=1e-12, atol=1e-12)


def test_six_of_nine_split_products_stay_below_fp32_accumulation_noise():
    """gemm mode 2 (prepared for the next round): the forward GEMM without the lo x lo, lo x mid, mid x lo partial products.
    Emulated in float64 on a decoder-shaped case: the error of leaving them out is several times SMALLER than the rounding an
    fp32 GEMM commits by accumulating in fp32, while the nine-product sum is exact."""
    rng = np.random.default_rng(0)

    def split3(x):
        def tr(v):
            return (np.ascontiguousarray(v, np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        hi = tr(x); r = (x - hi).astype(np.float32); mid = tr(r); lo = (r - mid).astype(np.float32)
        assert np.array_equal(hi.astype(np.float64) + mid + lo, x.astype(np.float64))
        return [p.astype(np.float64) for p in (hi, mid, lo)]

    M, K, N = 1024, 256, 256
    H1 = np.maximum(rng.normal(size=(M, K)) * 0.5, 0).astype(np.float32)
    W2T = (rng.uniform(-1, 1, size=(K, N)) / 16).astype(np.float32)
    a, b = split3(H1), split3(W2T)
    exact = H1.astype(np.float64) @ W2T.astype(np.float64)
    scale = np.abs(H1).astype(np.float64) @ np.abs(W2T).astype(np.float64)
    s9 = sum(a[i] @ b[j] for i in range(3) for j in range(3))
    s6 = sum(a[i] @ b[j] for i in range(3) for j in range(3) if i + j <= 2)
    f32 = (H1 @ W2T).astype(np.float64)
    rms = lambda s_: float(np.sqrt((((s_ - exact) / scale) ** 2).mean()))
    assert rms(s9) < 1e-15
    assert np.abs(s6 - exact).max() / scale.max() < 2.0 ** -23 and rms(s6) < 0.5 * rms(f32)


def test_eight_product_forward_drops_less_than_2_to_the_minus_30_of_a_product(hh):
    """The default forward GEMM of the decoder (gemm mode 3) forms eight of the nine partial products of the three-term splits: all but
    x_lo * w_lo.  In exact rational arithmetic: (i) the eight products sum to x * w - x_lo * w_lo exactly; (ii) |x_lo| < 2^-15 |x| for
    every fp32 x (truncation splits 24 significand bits 8 + 8 + 8: x_lo is what lies below bit 15 under the leading one), hence
    |x_lo * w_lo| < 2^-30 |x * w| - 2^-6 of ONE rounding of the fp32 accumulation the products then enter (2^-24 relative), so the
    dropped terms of a 256-deep dot product are bounded by 2^-30 sum |x_k| |w_k| against the accumulation's own ~2^-24 sum |x_k| |w_k|."""
    from fractions import Fraction
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.normal(size=60000).astype(np.float32), (rng.normal(size=20000) * 1e-5).astype(np.float32),
                        (rng.normal(size=20000) * 1e5).astype(np.float32),
                        rng.integers(0, 2 ** 32, 100000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    v = v[np.isfinite(v) & (np.abs(v) >= 2.0 ** -100) & (np.abs(v) < 2.0 ** 100)]
    n = len(v) // 2 * 2
    v = v[:n]
    hi, mid, lo = (np.empty(n, np.uint16) for _ in range(3))
    hh.hh_split3_bf16(n, p(v), p(hi), p(mid), p(lo))
    f = lambda b: (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    H_, M_, L_ = f(hi), f(mid), f(lo)
    assert (np.abs(L_) < np.abs(v.astype(np.float64)) * 2.0 ** -15).all()                 # (ii), all values incl. raw bit patterns
    x, w = slice(0, n // 2), slice(n // 2, n)
    # (i) + the bound, exactly, on a sample of pairs (Fractions of floats are exact)
    for k in range(0, n // 2, 97):
        xs = [Fraction(float(a[x][k])) for a in (H_, M_, L_)]
        ws = [Fraction(float(a[w][k])) for a in (H_, M_, L_)]
        exact = Fraction(float(v[x][k])) * Fraction(float(v[w][k]))
        eight = sum(xs[a] * ws[b] for a in range(3) for b in range(3) if a + b < 4)
        assert exact - eight == xs[2] * ws[2]
        assert abs(xs[2] * ws[2]) * 2 ** 30 < abs(exact) or exact == 0
    # a 256-deep dot product: the dropped part against the bound and against what fp32 accumulation loses anyway
    X = rng.normal(size=(64, 256)).astype(np.float32); W = rng.normal(size=(64, 256)).astype(np.float32)
    parts = []
    for arr in (X, W):
        h_, m_, l_ = (np.empty(arr.size, np.uint16) for _ in range(3))
        hh.hh_split3_bf16(arr.size, p(np.ascontiguousarray(arr.reshape(-1))), p(h_), p(m_), p(l_))
        parts.append([f(a).reshape(arr.shape) for a in (h_, m_, l_)])
    dropped = np.abs((parts[0][2] * parts[1][2]).sum(1))                                  # float64: exact enough for a bound check
    bound = (np.abs(X.astype(np.float64)) * np.abs(W.astype(np.float64))).sum(1) * 2.0 ** -30
    assert (dropped <= bound).all()
    fp32_acc = np.abs(np.cumsum((X * W).astype(np.float32), axis=1, dtype=np.float32)[:, -1].astype(np.float64) - (X.astype(np.float64) * W.astype(np.float64)).sum(1))
    assert np.median(dropped) < 0.05 * max(np.median(fp32_acc), 1e-30)                    # far inside the accumulation's own round-off


def test_fp16_pair_split(hh):
    """The arithmetic behind gemm modes 4 / 5 (nl_device_math.h nl_split2_f16): (i) the software fp32 -> fp16 conversion
    the weight-plane kernels use is IEEE round-to-nearest-even incl. subnormals (== numpy, what v_cvt_pk_f16_f32 does in the decoder kernels);
    (ii) hi + lo reproduces the scaled operand to 2^-23 relative (or 2^-25 absolute where lo is subnormal), saturating at +-65504;
    (iii) every fp16 x fp16 product is exact in fp32; (iv) hi hi' + hi lo' + lo hi' differs from the exact product of the operands by less than
    2^-21 of it - in exact rational arithmetic - i.e. below the rounding noise of the 256-deep fp32 accumulation the products enter."""
    from fractions import Fraction
    rng = np.random.default_rng(2)
    v = np.concatenate([
        rng.normal(size=200000).astype(np.float32), (rng.normal(size=50000) * 1e-4).astype(np.float32), (rng.normal(size=50000) * 3e4).astype(np.float32),
        rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.99, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 2.0 ** -14, 6.1e-5, 0.1, 1 / 3, np.inf, -np.inf], np.float32)])
    v = v[~np.isnan(v)]
    n = len(v)
    h, back = np.empty(n, np.uint16), np.empty(n, np.float32)
    hh.hh_f16_convert(n, p(v), p(h), p(back))
    with np.errstate(over="ignore"):
        ref = v.astype(np.float16)
    assert np.array_equal(h, ref.view(np.uint16))                                   # (i) bit for bit, subnormals and the overflow boundary included
    assert np.array_equal(back.view(np.uint32), ref.astype(np.float32).view(np.uint32))
    # (ii) the split of operands in the ranges the decoder sees, at the kernels' scales
    for scale, x in ((16.0, np.abs(rng.normal(0, 1.0, 100000)).astype(np.float32)), (256.0, rng.uniform(-0.07, 0.07, 100000).astype(np.float32)),
                     (64.0, rng.normal(0, 0.02, 100000).astype(np.float32)), (4096.0, (rng.uniform(-0.06, 0.06, 100000) * rng.uniform(-0.06, 0.06, 100000)).astype(np.float32))):
        hi, lo = np.empty(len(x), np.uint16), np.empty(len(x), np.uint16)
        hh.hh_split2_f16(len(x), p(x), ctypes.c_float(scale), p(hi), p(lo))
        H_, L_ = hi.view(np.float16).astype(np.float64), lo.view(np.float16).astype(np.float64)
        xs = x.astype(np.float64) * scale
        assert (np.abs(H_ + L_ - xs) <= np.maximum(np.abs(xs) * 2.0 ** -23, 2.0 ** -25)).all()
        assert (np.abs(L_) <= np.abs(xs) * 2.0 ** -11 + 2.0 ** -25).all()
        # (iii) products of two fp16 values carry at most 22 significand bits
        a, b = H_[:50000], L_[50000:]
        assert np.array_equal((a * b).astype(np.float32).astype(np.float64), a * b)
    big = np.array([1e6, -1e6, 70000.0], np.float32)                                # saturation instead of inf
    hi, lo = np.empty(3, np.uint16), np.empty(3, np.uint16)
    hh.hh_split2_f16(3, p(big), ctypes.c_float(1.0), p(hi), p(lo))
    assert np.array_equal(hi.view(np.float16).astype(np.float32), [65504.0, -65504.0, 65504.0]) and not lo.any()
    # (iv) three of the four partial products, exact arithmetic
    x = np.abs(rng.normal(0, 1.0, 400)).astype(np.float32) + np.float32(0.02)
    w = rng.uniform(-0.07, 0.07, 400).astype(np.float32); w[np.abs(w) < 1e-3] = np.float32(0.01)
    xh, xl, wh, wl = (np.empty(400, np.uint16) for _ in range(4))
    hh.hh_split2_f16(400, p(x), ctypes.c_float(16.0), p(xh), p(xl)); hh.hh_split2_f16(400, p(w), ctypes.c_float(256.0), p(wh), p(wl))
    F = lambda bits: [Fraction(float(t)) for t in bits.view(np.float16).astype(np.float64)]     # noqa: E731
    for X, Wv, a, b, c, d in zip(x, w, F(xh), F(xl), F(wh), F(wl)):
        exact = Fraction(float(X)) * 16 * Fraction(float(Wv)) * 256
        three = a * c + a * d + b * c
        assert abs(exact - three) * 2 ** 21 < abs(exact)
        assert abs(exact - (three + b * d)) * 2 ** 22 < abs(exact)                 # all four: only the operands' own last-bit rounding is left
