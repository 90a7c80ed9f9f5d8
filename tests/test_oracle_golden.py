"""CPU: pin the oracle (oracle/oracle.py + oracle/nl_oracle.c) against fixtures produced by the
REFERENCE python code (tests/golden/make_golden.py).  Tolerances are fp32 round-off of re-ordered
sums; integer/index outputs must be identical."""
import os

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O


def assert_mostly_close(a, b, atol, max_bad_frac=1e-4, hard=None):
    """Adam's first step moves every element by lr*g/(|g|+eps): elements whose gradient is ~eps are
    legitimately sensitive to round-off, so allow a vanishing fraction of outliers bounded by `hard`."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    assert (d > atol).mean() <= max_bad_frac, (d > atol).mean()
    if hard is not None:
        assert d.max() <= hard, d.max()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_se3_rotation_and_gradient(golden_dir):
    g = load(golden_dir, "se3")
    for w, G, R, gw in zip(g["w"], g["G"], g["R"], g["gw"]):
        np.testing.assert_allclose(O.rodrigues(w), R, rtol=0, atol=2e-7)
        np.testing.assert_allclose(O.rodrigues_backward(w, G), gw, rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("tag,bf16", [("bf16", True), ("f32", False)])
def test_adam_matches_torch(golden_dir, tag, bf16):
    g = load(golden_dir, "adam")
    p = O.bf16_round(g["p0"]) if bf16 else g["p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for t, grad in enumerate(g["gs"]):
        grad = O.bf16_round(grad) if bf16 else grad
        p, m, v = O.adam_step(p, grad, m, v, t + 1, 0.03, bf16=bf16)
        if bf16:
            assert np.array_equal(O.bf16_bits(p), O.bf16_bits(g["p_" + tag][t])), f"step {t}"
        else:
            np.testing.assert_allclose(p, g["p_" + tag][t], rtol=1e-5, atol=1e-8)


def _voxel(g):
    return float(g["voxel_size"]) if "voxel_size" in g.files else H.VOXEL


def _scene(g):
    sc = H.build_oracle_scene(int(g["n_beams"]), int(g["n_azimuth"]), int(g["seed"]), voxel=_voxel(g))
    # raw row numbering of duplicate vertex ids is assignment-order dependent in the reference
    # (SURVEY B7): take the fixture's table, check ours covers the same vertices
    assert np.array_equal(sc["ms"].id2row >= 0, g["id_table"] >= 0)
    assert sc["E"] == int(g["n_emb_rows"])
    sc["ms"].id2row = g["id_table"].copy()
    return sc


def tie_rays(out):
    """Hit rays whose sorted hit list holds two voxels with EQUAL t_min.  The reference orders such
    ties with torch.sort (unstable; its CPU and CUDA back-ends differ), the oracle and the HIP path
    define the order as stable (DFS order first) - results on these rays are unspecified in the
    reference and are excluded from element-wise comparison (DESIGN.md "ties")."""
    hr = np.nonzero(out["hits"])[0]
    t0, idx = out["hit_t0"][hr], out["hit_idx"][hr]
    eq = (t0[:, 1:] == t0[:, :-1]) & (idx[:, 1:] != -1)
    return eq.any(1)


def _check_iter(out, g, it, strict_loss=True):
    ok = ~tie_rays(out)
    assert ok.mean() > 0.98
    assert np.array_equal(out["valid"][ok], g[f"it{it}_valid"][ok])
    np.testing.assert_allclose(out["z_vals"][ok], g[f"it{it}_z_vals"][ok], rtol=0, atol=2e-5)
    assert np.abs(out["sdf"] - g[f"it{it}_sdf"])[ok].max() < 1e-4          # north_star tolerance
    assert np.abs(out["sdf"] - g[f"it{it}_sdf"])[ok].mean() < 1e-6
    np.testing.assert_allclose(out["loss"], g[f"it{it}_loss"], rtol=2e-5 if ok.all() else 2e-3)


@pytest.mark.parametrize("case", ["map_1f_1it", "map_kitti_1f_1it", "map_ncd_1f_1it"])
def test_mapping_one_iteration(golden_dir, case):
    """maicity mapper settings, and those of the kitti (voxel 0.3 m, step 0.15 m) and ncd (voxel 0.2 m, step 0.04 m: up to 58
    samples per ray) configs"""
    g = load(golden_dir, case)
    sc = _scene(g)
    masks = H.unpack_masks(g["masks"], len(sc["points"]))
    dec = O.decoder_init(int(g["seed"]))
    scans = [dict(points=sc["points"], cos=sc["cos"], pose=g["poses0"][0].copy(), index=1)]
    cfg = O.IterCfg(step_size=float(g["step_size"]))
    outs = O.bundle_adjust(sc["ms"], dec, scans, masks, cfg, 1, list(g["lrs"]), emb_accumulate="bf16_seq")
    out = outs[0]
    assert np.array_equal(out["hits"], g["it0_ray_mask"][0])
    assert np.array_equal(out["hit_idx"], g["it0_hit_idx"][0])
    np.testing.assert_allclose(out["hit_t0"], g["it0_hit_t0"][0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["hit_t1"], g["it0_hit_t1"][0], rtol=0, atol=1e-5)
    _check_iter(out, g, 0)
    for n in dec.names():
        ref = g["decG_" + n]
        got = out["grad_dec"][n].reshape(ref.shape)
        assert np.abs(got - ref).max() <= 5e-5 * np.abs(ref).max() + 1e-9, n
    np.testing.assert_allclose(out["grad_pose"][0], g["pose_grad_last"][0], rtol=1e-3, atol=2e-7)
    ge = O.bf16_to_f32(H.scatter_rows(sc["E"], g["emb_grad_rows"], g["emb_grad_vals"]))
    got = O.bf16_to_f32(out["grad_emb"])
    assert np.array_equal(got != 0, ge != 0) or (np.abs(got - ge).max() < 1e-5)
    assert np.linalg.norm(got - ge) <= 2e-3 * np.linalg.norm(ge)
    # after one Adam step
    ef = H.scatter_rows(sc["E"], g["emb_final_rows"], g["emb_final_vals"], O.bf16_bits(H.init_embeddings(sc["E"], int(g["seed"]))))
    mism = (sc["ms"].emb != ef).mean()
    assert mism < 2e-3, mism
    for n in dec.names():
        assert_mostly_close(getattr(dec, n), g["decF_" + n], atol=2e-5, max_bad_frac=5e-3, hard=2 * 0.005)
    np.testing.assert_allclose(scans[0]["pose"], g["poses_final"][0], rtol=0, atol=2e-4)


def test_mapping_three_iterations(golden_dir):
    g = load(golden_dir, "map_1f_3it")
    sc = _scene(g)
    masks = H.unpack_masks(g["masks"], len(sc["points"]))
    dec = O.decoder_init(int(g["seed"]))
    scans = [dict(points=sc["points"], cos=sc["cos"], pose=g["poses0"][0].copy(), index=1)]
    cfg = O.IterCfg(step_size=float(g["step_size"]))
    outs = O.bundle_adjust(sc["ms"], dec, scans, masks, cfg, int(g["n_iter"]), list(g["lrs"]),
                           emb_accumulate="bf16_seq")
    _check_iter(outs[0], g, 0)
    # later iterations see parameters after Adam steps whose first step is +-lr regardless of
    # gradient magnitude: tiny gradient differences can flip a few signs, so compare loosely
    for it in (1, 2):
        ok = ~tie_rays(outs[it])
        assert np.array_equal(outs[it]["valid"][ok], g[f"it{it}_valid"][ok])
        assert np.abs(outs[it]["sdf"] - g[f"it{it}_sdf"])[ok].mean() < 5e-3
        np.testing.assert_allclose(outs[it]["loss"], g[f"it{it}_loss"], rtol=5e-2)
    np.testing.assert_allclose(scans[0]["pose"][:3], g["poses_final"][0][:3], rtol=0, atol=5e-3)


def test_mapping_two_frames_frozen(golden_dir):
    g = load(golden_dir, "map_2f_2it_frozen")
    sc = _scene(g)
    masks = H.unpack_masks(g["masks"], len(sc["points"]))
    dec = O.decoder_init(int(g["seed"]))
    scans = [dict(points=sc["points"], cos=sc["cos"], pose=g["poses0"][f].copy(), index=f + 1) for f in range(2)]
    cfg = O.IterCfg(step_size=float(g["step_size"]))
    outs = O.bundle_adjust(sc["ms"], dec, scans, masks, cfg, 2, list(g["lrs"]), update_pose=False,
                           update_decoder=False, emb_accumulate="bf16_seq")
    _check_iter(outs[0], g, 0)
    assert np.array_equal(outs[0]["hits"], g["it0_ray_mask"][0])
    np.testing.assert_allclose(np.stack([s["pose"] for s in scans]), g["poses_final"], rtol=0, atol=0)


@pytest.mark.parametrize("case", ["track_2it", "track_kitti_2it", "track_ncd_2it"])
def test_tracking_two_iterations(golden_dir, case):
    """tracker settings of the maicity, kitti (voxel 0.3 m, step 0.06 m, lr 0.06) and ncd (step 0.02 m, lr 0.04) configs"""
    g = load(golden_dir, case)
    sc = _scene(g)
    masks = H.unpack_masks(g["masks"], len(sc["points"]))
    dec = O.decoder_init(int(g["seed"]))
    scan = dict(points=sc["points"], cos=sc["cos"], pose=g["pose0"].copy(), index=int(g["frame_index"]))
    cfg = O.IterCfg(step_size=float(g["step_size"]))
    pose, outs = O.track(sc["ms"], dec, scan, masks, cfg, 2, float(g["lr"]))
    _check_iter(outs[0], g, 0)
    np.testing.assert_allclose(outs[1]["grad_pose"][0], g["pose_grad_last"], rtol=5e-2, atol=1e-5)
    np.testing.assert_allclose(pose, g["pose_final"], rtol=0, atol=3e-4)
    assert np.array_equal(outs[-1]["hits"], g["hit_mask"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_criterion_on_caller_tensors(golden_dir, tag):
    """oracle.sdf_loss against the reference's OWN Criterion.forward + autograd on tensors it did not produce itself
    (make_golden.py run_criterion_case): the checker of nerf_loam_amd.criterion.Criterion's caller-tensor path is pinned to the reference class"""
    g = load(golden_dir, "criterion")
    N = int(g[f"{tag}_n"])
    ray_mask = np.unpackbits(g[f"{tag}_ray_mask"])[:N].astype(bool)
    z = g[f"{tag}_z_vals"]
    valid = np.unpackbits(g[f"{tag}_valid"], axis=-1)[:, :z.shape[1]].astype(bool)
    loss, dsdf, st = O.sdf_loss(z, g[f"{tag}_sdf"], valid, g[f"{tag}_points"][ray_mask], g[f"{tag}_cos"][ray_mask],
                                O.LossCfg(truncation=0.3, sdf_weight=10000.0, fs_weight=1.0, max_depth=50.0))
    assert abs(float(loss) - float(g[f"{tag}_loss"])) <= 1e-6 * float(g[f"{tag}_loss"])
    assert abs(float(st["fs_loss"]) - float(g[f"{tag}_fs_loss"])) <= 1e-6 * float(g[f"{tag}_fs_loss"])
    assert abs(float(st["sdf_loss"]) - float(g[f"{tag}_sdf_loss"])) <= 1e-6 * float(g[f"{tag}_sdf_loss"])
    ref = g[f"{tag}_dsdf"]
    assert np.abs(dsdf - ref).max() <= 1e-5 * np.abs(ref).max() and np.array_equal(dsdf != 0, ref != 0)
