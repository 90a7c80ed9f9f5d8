"""GPU (-m gpu): the two workgroup layouts of the fused fp16-pair decoder kernel (include/nerfloam_hip.h NL_KERNEL_LAYOUT, round 6).

Layout 1 = one 8-wave workgroup per CU (k_decoder, rounds 1-5); layout 2 = two independent 4-wave workgroups per CU (k_decoder2).  Per element
the two perform the same arithmetic in the same order: sdf, dL/dsdf and the saved ReLU words must be IDENTICAL; dX (its 256-deep sum is split over the
four waves in layout 2) and the weight gradients (another partition of the samples over the slabs) agree to fp32 re-association.  Both against the
oracle elsewhere (tests/test_gpu_parity.py runs under the default layout, which is 2 for every engine: its slab count is nl_decoder_grid_hint())."""
import os

import numpy as np
import pytest
import torch

import helpers as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nl():
    from nerf_loam_amd import _lib, ops, pipeline
    _lib.require_gpu()
    return dict(L=_lib, ops=ops, P=pipeline)


def _scene(golden_dir, case):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    sc = H.build_oracle_scene(int(g["n_beams"]), int(g["n_azimuth"]), int(g["seed"]))
    sc["ms"].id2row = g["id_table"].copy()
    masks = H.unpack_masks(g["masks"], len(sc["points"]))
    dec_np = O.decoder_init(int(g["seed"]))
    frames = [O.select_rays(sc["points"], sc["cos"], g["poses0"][0].copy(), masks[0][0], optimize_pose=True)]
    return sc, dec_np, frames, float(g["step_size"])


def _run(nl, sc, dec_np, frames, step, layout, gemm, train, one_call, n_slabs=None, scale=1.0):
    P = nl["P"]
    ms = sc["ms"]
    m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, ms.emb, ms.voxel_size)
    dec = P.DecoderDevice(dec_np.W1 * scale, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
    n = len(frames[0].rays_d)
    old = os.environ.get("NL_N_SLABS")
    if n_slabs is not None:
        os.environ["NL_N_SLABS"] = str(n_slabs)
    try:
        eng = P.SdfEngine(max_rays=n, samples_per_ray_cap=64, max_frames=2, gemm_mode=gemm, wgrad2_mode=2, dec_layout=layout)
    finally:
        if n_slabs is not None:
            os.environ.pop("NL_N_SLABS") if old is None else os.environ.__setitem__("NL_N_SLABS", old)
    fr = frames[0]
    eng.set_rays(fr.rays_d, fr.points, fr.cos)
    eng.set_poses(fr.pose[None], [1])
    cfg = P.IterConfig(step_size=step)
    eng.begin_call(m, dec)
    if one_call:
        eng.bind(m, dec, cfg, train_decoder=train)
        eng.run_bound(1)
    else:
        eng.forward_backward(m, dec, cfg, train_decoder=train)
    torch.cuda.synchronize()
    Pn = eng.stats()["P"]
    tiles = (Pn + 63) // 64
    out = {"P": Pn, "sdf": eng.sdf[:Pn].cpu().numpy(), "dsdf": eng.dsdf[:Pn].cpu().numpy(), "dX": eng.dX[:Pn].cpu().numpy(),
           "g_emb": eng.g_emb_total().cpu().numpy().copy(), "g_pose": eng.g_pose.cpu().numpy().copy(), "loss": eng.loss_value(cfg)["loss"]}
    if train:
        out["gdec"] = dec.grad.cpu().numpy().copy()
        out["mask"] = eng.relu2_mask[:tiles * 512].cpu().numpy().copy()
    return out


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("case", ["map_1f_1it", "map_ncd_1f_1it"])
@pytest.mark.parametrize("gemm", [4, 5])
@pytest.mark.parametrize("train", [True, False])
def test_two_workgroups_per_cu_equal_the_one_workgroup_kernel(nl, golden_dir, case, gemm, train):
    sc, dec_np, frames, step = _scene(golden_dir, case)
    for one_call in (False, True):
        a = _run(nl, sc, dec_np, frames, step, 1, gemm, train, one_call)
        b = _run(nl, sc, dec_np, frames, step, 2, gemm, train, one_call)
        assert a["P"] == b["P"] and a["P"] > 1000
        assert np.array_equal(a["sdf"].view(np.uint32), b["sdf"].view(np.uint32)), "sdf differs between the layouts"
        assert np.array_equal(a["dsdf"].view(np.uint32), b["dsdf"].view(np.uint32)), "dL/dsdf differs between the layouts"
        assert abs(a["loss"] - b["loss"]) <= 1e-12 * abs(a["loss"])         # (the same residuals; fp64 atomics in another order)
        assert _rel(b["dX"], a["dX"]) < 2e-6, _rel(b["dX"], a["dX"])
        # the embedding gradient accumulates bf16-ROUNDED contributions (the reference's embedding_dense_backward semantics): a last-bit difference of dX
        # moves a contribution by 2^-9 of itself now and then
        assert _rel(b["g_emb"], a["g_emb"]) < 2e-4 and _rel(b["g_pose"], a["g_pose"]) < 1e-5, (_rel(b["g_emb"], a["g_emb"]), _rel(b["g_pose"], a["g_pose"]))
        if train:
            assert np.array_equal(a["mask"], b["mask"]), "saved ReLU words differ (the dW2 kernel reads them)"
            L = nl["L"]
            for name, sl in (("W1", slice(L.OFF_W1, L.OFF_B1)), ("b1", slice(L.OFF_B1, L.OFF_W2)), ("W2", slice(L.OFF_W2, L.OFF_B2)),
                             ("b2", slice(L.OFF_B2, L.OFF_W3)), ("W3", slice(L.OFF_W3, L.OFF_B3)), ("b3", slice(L.OFF_B3, L.OFF_B3 + 1))):
                assert _rel(b["gdec"][sl], a["gdec"][sl]) < 2e-5, (name, _rel(b["gdec"][sl], a["gdec"][sl]))


def test_layout_two_with_few_and_with_odd_slab_counts(nl, golden_dir):
    """layout 2 forced on engines with 3 and 8 slabs (fewer workgroups than tiles: the persistent loop; an odd count: the dW2 kernel's grid is clamped
    separately) against the default engine: the gradient is the same sum whatever the partition."""
    sc, dec_np, frames, step = _scene(golden_dir, "map_1f_1it")
    ref = _run(nl, sc, dec_np, frames, step, 1, 4, True, False)
    for ns in (3, 8):
        for one_call in (False, True):
            b = _run(nl, sc, dec_np, frames, step, 2, 4, True, one_call, n_slabs=ns)
            assert np.array_equal(ref["sdf"].view(np.uint32), b["sdf"].view(np.uint32)) and np.array_equal(ref["mask"], b["mask"])
            assert _rel(b["gdec"], ref["gdec"]) < 2e-5 and _rel(b["dX"], ref["dX"]) < 2e-6


def test_w1_operand_planes_follow_the_optimiser(nl, golden_dir):
    """The W1 operand planes of the weight workspace (W1F / W1X, read by layout 2 only) are rebuilt by the optimiser's one launch: three training
    iterations under layout 2 end at the same decoder as under layout 1 to rounding, and a stale plane would not."""
    sc, dec_np, frames, step = _scene(golden_dir, "map_1f_1it")
    P = nl["P"]
    res = {}
    for layout in (1, 2):
        ms = sc["ms"]
        m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, ms.emb.copy(), ms.voxel_size)
        dec = P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
        eng = P.SdfEngine(max_rays=len(frames[0].rays_d), samples_per_ray_cap=64, max_frames=2, gemm_mode=4, wgrad2_mode=2, dec_layout=layout)
        fr = frames[0]
        eng.set_rays(fr.rays_d, fr.points, fr.cos); eng.set_poses(fr.pose[None], [1])
        cfg = P.IterConfig(step_size=step)
        eng.begin_call(m, dec)
        sdfs = []
        for it in range(3):
            eng.forward_backward(m, dec, cfg, train_decoder=True)
            sdfs.append(eng.sdf[:eng.stats()["P"]].cpu().numpy().copy())
            eng.optimiser_step(m, dec, cfg, update_decoder=True)
        torch.cuda.synchronize()
        # the planes against a fresh rebuild from the updated parameters
        ws = dec.W2T.clone()
        nl["ops"].decoder_transpose_w2(dec.params, ws)
        torch.cuda.synchronize()
        assert torch.equal(ws.view(torch.int32), dec.W2T.view(torch.int32)), "operand planes after the optimiser step != planes rebuilt from the parameters"
        res[layout] = (sdfs, dec.params.cpu().numpy().copy())
    for it in range(3):
        assert np.abs(res[1][0][it] - res[2][0][it]).max() < 2e-5, it
    moved = np.abs(res[1][1] - np.concatenate([dec_np.W1.ravel(), dec_np.b1, dec_np.W2.ravel(), dec_np.b2, dec_np.W3.ravel(), dec_np.b3])).max()
    assert moved > 1e-4 and np.abs(res[1][1] - res[2][1]).max() < 0.05 * moved


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("one_call", [False, True])
def test_clipped_operands_are_reported(nl, golden_dir, layout, one_call):
    """The fp16-pair arithmetic clips what leaves its range (the reference's decoder is unbounded fp32, lidar.py:109-123); round 6 makes that loud:
    NL_SAT_* bits in the weight workspace's sticky status word, latched into the call status by the optimiser.  Each way out of the range, under both
    kernel layouts, on the stage-wise and on the one-call path; a clean call reports nothing and begin_call clears the word."""
    L, P = nl["L"], nl["P"]
    sc, dec_np, frames, step = _scene(golden_dir, "map_1f_1it")
    fr = frames[0]
    n = len(fr.rays_d)
    cases = {
        "clean": (dict(), 1.0, 0),
        # (the golden scene: |emb| <= 0.048, |W1| <= 0.25, |W2|, |w3| <= 0.0625)
        "X": (dict(), 2.0e5, L.NL_SAT_X),                                    # |X| ~ 2000 > 1023
        "X_train_only": (dict(), 12000.0, L.NL_SAT_X),                       # |X| up to 574: inside X * 2^6, outside U = sigma dsdf 16 X (trainable decoder only)
        "NaN": (dict(nan=True), 1.0, L.NL_SAT_X),
        "planes": (dict(W2=5000.0), 1.0, L.NL_SAT_PLANES),                   # |W2| up to 312 > 256
        "H1": (dict(W1=200.0), 2000.0, L.NL_SAT_H1),                         # |W1| <= 50, |X| <= 96: planes and X inside, H1 (std ~2300 per element) beyond 4094
        "Q": (dict(W2=100.0, W3=100.0), 1.0, L.NL_SAT_Q),                    # |w3 W2| <= 39 < 64, |W2| <= 6.2: planes inside, the masked 256-deep sums (std ~150) beyond 64
    }
    for name, (wscale, escale, want) in cases.items():
        ms = sc["ms"]
        emb = O.bf16_to_f32(ms.emb) * np.float32(escale)
        if wscale.get("nan"):
            emb = emb.copy(); emb[ms.id2row[ms.id2row >= 0][:50]] = np.nan
        m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, O.bf16_bits(emb), ms.voxel_size)
        dec = P.DecoderDevice(dec_np.W1 * np.float32(wscale.get("W1", 1.0)), dec_np.b1, dec_np.W2 * np.float32(wscale.get("W2", 1.0)), dec_np.b2,
                              dec_np.W3 * np.float32(wscale.get("W3", 1.0)), dec_np.b3)
        eng = P.SdfEngine(max_rays=n, samples_per_ray_cap=64, max_frames=2, gemm_mode=4, wgrad2_mode=2, dec_layout=layout)
        eng.set_rays(fr.rays_d, fr.points, fr.cos); eng.set_poses(fr.pose[None], [1])
        cfg = P.IterConfig(step_size=step)
        for train in ((True, False) if name in ("clean", "X_train_only") else (True,)):
            eng.begin_call(m, dec)
            assert dec.range_status() == 0
            if one_call:
                eng.bind(m, dec, cfg, train_decoder=train, update_decoder=False, update_emb=False, update_pose=False)
                eng.run_bound()
            else:
                eng.forward_backward(m, dec, cfg, train_decoder=train)
                eng.optimiser_step(m, dec, cfg, update_decoder=False, update_emb=False, update_pose=False)
            eng.call_status()
            st = dec.range_status()
            expect = want if (train or name != "X_train_only") else 0
            if expect == 0:
                assert st == 0 and not eng.saturated, (name, train, st)
            else:
                assert st & expect and eng.saturated, (name, train, st)
                if name != "planes":
                    assert not st & L.NL_SAT_PLANES, (name, st)
    # the forward-only kernel (render_rays / get_scores) reports through the same word
    ms = sc["ms"]
    m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, O.bf16_bits(O.bf16_to_f32(ms.emb) * np.float32(2.0e5)), ms.voxel_size)      # |X| ~ 2000
    dec = P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
    eng = P.SdfEngine(max_rays=n, samples_per_ray_cap=64, max_frames=2, gemm_mode=4, dec_layout=layout)
    eng.set_rays(fr.rays_d, fr.points, fr.cos); eng.set_poses(fr.pose[None], [1])
    eng.begin_call(m, dec, emb_state=False)
    eng.forward_only(m, dec, P.IterConfig(step_size=step))
    assert dec.range_status(clear=True) & L.NL_SAT_X and dec.range_status() == 0
