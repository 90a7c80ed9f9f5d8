"""GPU (-m gpu): the touched-rows embedding optimiser (include/nerfloam_hip.h NlTouchedRows) against the dense sweep.

The reference's torch.optim.Adam steps over the whole [E,16] bf16 table every iteration (render_helpers.py:341-353,421-423) and is
constructed anew per call (:353).  A row that never received a gradient has zero gradient and zero moments and does not move, so the
HIP path records the rows the scatter touches since begin_call and (i) sweeps only those in the optimiser, (ii) clears only those at the
next begin_call - bit-identical to the dense sweep + E-sized memset (SdfEngine(sparse_adam=False)), at a cost proportional to the touched
rows.  Checked here: several calls of several iterations with different rays each, on a table much larger than what the rays touch, a
table that grows between calls (rows appended, like Mapping.update_grid_features does per frame), and an unusable iteration in between
(skip_mode: the accumulators are cleared, nothing else moves)."""
import os

import numpy as np
import pytest
import torch

import helpers as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _scene(golden_dir, pad_rows):
    from nerf_loam_amd import pipeline as P
    g = np.load(os.path.join(golden_dir, "map_1f_3it.npz"))
    sc = H.build_oracle_scene(int(g["n_beams"]), int(g["n_azimuth"]), int(g["seed"]))
    sc["ms"].id2row = g["id_table"].copy()
    masks = H.unpack_masks(g["masks"], len(sc["points"]))[0]               # [iteration][point]
    emb = np.concatenate([sc["ms"].emb, np.zeros((pad_rows, 16), np.uint16)])
    dec_np = O.decoder_init(int(g["seed"]))
    return g, sc, masks, emb, dec_np, P


def _run_pair(P, sc, masks, emb_bits, dec_np, pose0, one_call, grow_after_call=0):
    """two calls of three iterations each (second call: the masks in reverse order + one unusable iteration in front) on TWO engines in lock
    step - touched-rows bookkeeping and dense bookkeeping.  What is under test is the optimiser + its bookkeeping, so both optimisers must see
    the SAME gradients: the forward + backward pass runs on the touched-rows engine only (its scatter records the rows), and its
    accumulators - embedding, decoder, pose - and counter block are copied into the dense engine before each optimiser step.  (Two separate
    backward passes differ in the order their fp32 atomics reach an accumulator row - the sum changes in its last bit, one bf16 rounding in
    ~10^5 flips: scripts/sparse_adam_diag.py - which says nothing about the bookkeeping.)  Every piece of optimiser state is compared after
    EVERY iteration, bit for bit."""
    ms = sc["ms"]
    n_rays = int(masks[0].sum())
    eng = {k: P.SdfEngine(max_rays=n_rays, samples_per_ray_cap=64, max_frames=2, sparse_adam=(k == "sparse")) for k in ("sparse", "dense")}
    dec = {k: P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3) for k in eng}
    cfg = P.IterConfig(step_size=0.1)
    emb_t = {k: torch.from_numpy(emb_bits.view(np.int16).copy()).cuda() for k in eng}
    out = {}

    def state(k, m):
        e = eng[k]
        return dict(emb=m.emb.cpu().numpy().copy(), m=e.emb_m.cpu().numpy().copy(), v=e.emb_v.cpu().numpy().copy(), g=e.g_emb.cpu().numpy().copy(),
                    dec=dec[k].params.cpu().numpy().copy(), dec_m=dec[k].m.cpu().numpy().copy(), pose=e.pose6[0].cpu().numpy().copy(),
                    adam=e.adam_state[:4].cpu().numpy().copy())

    for call in range(2):
        m = {}
        for k in eng:
            if call == 1 and grow_after_call:                              # the map grew: rows appended (zeros), like a new frame's vertices
                emb_t[k] = torch.cat([emb_t[k], torch.zeros(grow_after_call, 16, dtype=torch.int16, device="cuda")])
            m[k] = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, emb_t[k].cpu().numpy().view(np.uint16), ms.voxel_size)
            eng[k].set_poses(pose0[None], [1])
            eng[k].begin_call(m[k], dec[k])
            if one_call:
                eng[k].bind(m[k], dec[k], cfg, train_decoder=True, skip_mode=1)
        order = [0, 1, 2] if call == 0 else [None, 2, 1, 0]
        for it in order:
            fr = O.select_rays(sc["points"], sc["cos"], pose0, masks[0 if it is None else it])
            for k in eng:                                                  # (None: rays looking away from the map - no hit, the step is skipped on the device)
                eng[k].set_rays(-fr.rays_d if it is None else fr.rays_d, fr.points, fr.cos)
            es, ed = eng["sparse"], eng["dense"]
            if one_call:
                es.run_bound(1)
            else:
                es.forward_backward(m["sparse"], dec["sparse"], cfg, train_decoder=True)
            ed.g_emb.copy_(es.g_emb); ed.g_pose.copy_(es.g_pose); dec["dense"].grad.copy_(dec["sparse"].grad); ed.counters.copy_(es.counters)
            for k in eng:
                if one_call:
                    eng[k].run_bound(2)
                else:
                    eng[k].optimiser_step(m[k], dec[k], cfg, skip_mode=1)
            torch.cuda.synchronize()
            a, b_ = state("sparse", m["sparse"]), state("dense", m["dense"])
            for key in a:
                assert np.array_equal(a[key], b_[key]), (call, it, key)
        for k in eng:
            steps, skipped, overflow = eng[k].call_status()
            assert (steps, skipped, overflow) == ((3, 0, False) if call == 0 else (3, 1, False)), k
            emb_t[k] = m[k].emb.clone()
        out[call] = state("sparse", m["sparse"])
        lst, cnt, flags = eng["sparse"]._touched[:3]
        n = int(cnt.item())
        rows = np.sort(lst[:n].cpu().numpy())
        assert len(np.unique(rows)) == n                                    # every row listed once
        live = np.nonzero((out[call]["m"] != 0).any(1) | (out[call]["v"] != 0).any(1))[0]
        assert np.isin(live, rows).all()                                    # every row that carries moments is listed ...
        bits = np.unpackbits(flags.cpu().numpy().view(np.uint8), bitorder="little")
        assert np.array_equal(np.nonzero(bits)[0], rows)                    # ... and flagged; nothing else is
        out[call]["touched"] = n
    return out


@pytest.mark.parametrize("one_call", [False, True])
@pytest.mark.parametrize("pad_rows,grow", [(0, 0), (300000, 0), (50000, 20000)])
def test_touched_rows_adam_equals_the_dense_sweep(golden_dir, one_call, pad_rows, grow):
    g, sc, masks, emb, dec_np, P = _scene(golden_dir, pad_rows)
    pose0 = g["poses0"][0].copy()
    res = _run_pair(P, sc, masks, emb, dec_np, pose0, one_call=one_call, grow_after_call=grow)
    for call in (0, 1):
        assert not res[call]["g"].any()                                     # the optimiser leaves the accumulators cleared
        moved = int((res[call]["m"] != 0).any(1).sum())
        assert 0 < moved <= res[call]["touched"] < 40000                    # a few 10^4 rows of the (up to 3.5 x 10^5-row) table
    assert not np.array_equal(res[0]["emb"][:len(emb)], emb) and not np.array_equal(res[1]["emb"], np.concatenate([res[0]["emb"], np.zeros((grow, 16), np.uint16)]))


def test_begin_call_cost_does_not_depend_on_the_table_size(golden_dir):
    """begin_call + the optimiser step on a 2 x 10^6-row table (64 MB of bf16 rows, 320 MB of optimiser state) cost what they cost on the
    one-scan table: within 30 % + 10 us (timed with events over 20 calls; the dense engine pays the E-sized memset and sweep)"""
    g, sc, masks, emb, dec_np, P = _scene(golden_dir, 0)
    pose0 = g["poses0"][0].copy()
    ms = sc["ms"]
    fr = O.select_rays(sc["points"], sc["cos"], pose0, masks[0])
    times = {}
    for tag, pad, sparse in (("small", 0, True), ("large", 2000000, True), ("large_dense", 2000000, False)):
        emb_p = np.concatenate([ms.emb, np.zeros((pad, 16), np.uint16)])
        m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, emb_p, ms.voxel_size)
        dec = P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
        eng = P.SdfEngine(max_rays=len(fr.rays_d), samples_per_ray_cap=64, max_frames=2, sparse_adam=sparse)
        eng.set_rays(fr.rays_d, fr.points, fr.cos); eng.set_poses(pose0[None], [1])
        cfg = P.IterConfig(step_size=0.1)
        for _ in range(3):
            eng.begin_call(m, dec); eng.bind(m, dec, cfg, train_decoder=False, update_decoder=False); eng.run_bound(); eng.run_bound()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tb = to = 0.0
        for _ in range(20):
            ev[0].record(); eng.begin_call(m, dec); ev[1].record()
            eng.run_bound(1)
            ev[2].record(); eng.run_bound(2); ev[3].record()
            torch.cuda.synchronize()
            tb += ev[0].elapsed_time(ev[1]); to += ev[2].elapsed_time(ev[3])
        times[tag] = (tb / 20, to / 20)
    H.record_gpu_metric("sparse_adam_cost_ms", **{f"{k}_{n}": v for k, (b, o) in times.items() for n, v in (("begin_call", b), ("optimiser", o))})
    (bs, os_), (bl, ol), (bd, od) = times["small"], times["large"], times["large_dense"]
    assert bl <= 1.3 * bs + 0.010 and ol <= 1.3 * os_ + 0.010, times
    # what the dense bookkeeping costs at this size (measured: begin_call 0.037 -> 0.054 ms - a 320 MB memset on top of the call's four
    # small launches -, optimiser step 0.012 -> 0.056 ms; both grow linearly with the table: x 5 at a KITTI-scale 1e7 rows)
    # (the optimiser sweep is the robust half of that statement: 5 x on a quiet box.  The begin_call difference - a 320 MB memset that mostly overlaps the call's
    #  other launches - is 4-17 us by box and is recorded above, not asserted: a 4 us bar failed two runs in six on shared boxes)
    assert od > 3 * ol, times


@pytest.mark.parametrize("one_call", [False, True])
def test_replicated_accumulators_equal_a_single_array(golden_dir, one_call):
    """SdfEngine(emb_grad_copies=K) (NlTouchedRows.copies, round 5): the scatter's waves add into K accumulator arrays, the optimiser's sweep over the touched
    rows sums a row's copies in copy order and clears them.  Two engines in lock step - K = 8 and K = 1 -, both running their own backward pass: (i) the K arrays
    add up to the single array's sums to fp32 round-off (the order of the atomics differs, nothing else), the same rows are touched; (ii) fed the SEQUENTIAL
    fp32 sum of the K arrays as its accumulators, the single-array optimiser ends bit for bit where the K-copy optimiser ends - parameters, moments, and all
    K arrays left cleared; (iii) the next call's reset clears every copy.  On a table much larger than what the rays touch."""
    g, sc, masks, emb, dec_np, P = _scene(golden_dir, 100000)
    pose0 = g["poses0"][0].copy()
    ms = sc["ms"]
    n_rays = int(masks[0].sum())
    K = 8
    eng = {k: P.SdfEngine(max_rays=n_rays, samples_per_ray_cap=64, max_frames=2, emb_grad_copies=c) for k, c in (("copies", K), ("single", 1))}
    dec = {k: P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3) for k in eng}
    cfg = P.IterConfig(step_size=0.1)
    emb_t = {k: torch.from_numpy(emb.view(np.int16).copy()).cuda() for k in eng}
    for call in range(2):
        m = {k: P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, emb_t[k].cpu().numpy().view(np.uint16), ms.voxel_size) for k in eng}
        for k in eng:
            eng[k].set_poses(pose0[None], [1])
            eng[k].begin_call(m[k], dec[k])
            if one_call:
                eng[k].bind(m[k], dec[k], cfg, train_decoder=True, skip_mode=1)
        ec, es = eng["copies"], eng["single"]
        assert ec.emb_grad_copies == K and es.emb_grad_copies == 1
        cap, E = ec._emb_cap, ec.g_emb.shape[0]
        copies = ec._emb_state[:K * cap * 16].view(K, cap, 16)
        assert not copies.any()                                             # (iii) begin_call left every copy cleared
        for it in ([0, 1, 2] if call == 0 else [2, 0]):
            fr = O.select_rays(sc["points"], sc["cos"], pose0, masks[it])
            for k in eng:
                eng[k].set_rays(fr.rays_d, fr.points, fr.cos)
                eng[k].run_bound(1) if one_call else eng[k].forward_backward(m[k], dec[k], cfg, train_decoder=True)
            torch.cuda.synchronize()
            used = int((copies[:, :E] != 0).any(2).any(1).sum())
            assert used == K                                               # the waves really spread over the copies
            tot, ref = ec.g_emb_total().cpu().numpy().astype(np.float64), es.g_emb.cpu().numpy().astype(np.float64)
            assert np.abs(tot - ref).max() <= 2e-5 * np.abs(ref).max()      # (i)
            n = {k: int(eng[k]._touched[1].item()) for k in eng}
            rows = {k: np.sort(eng[k]._touched[0][:n[k]].cpu().numpy()) for k in eng}
            assert np.array_equal(rows["copies"], rows["single"])
            seq = copies[0, :E].clone()                                    # (ii) the sum the sweep forms: copy 0, then + copy 1, + copy 2, ... in fp32
            for c in range(1, K):
                seq = seq + copies[c, :E]
            es.g_emb.copy_(seq); es.g_pose.copy_(ec.g_pose); dec["single"].grad.copy_(dec["copies"].grad); es.counters.copy_(ec.counters)
            for k in eng:
                eng[k].run_bound(2) if one_call else eng[k].optimiser_step(m[k], dec[k], cfg, skip_mode=1)
            torch.cuda.synchronize()
            assert not copies.any() and not es.g_emb.any()                  # every copy cleared by the sweep
            for a_, b_ in ((m["copies"].emb, m["single"].emb), (ec.emb_m, es.emb_m), (ec.emb_v, es.emb_v), (dec["copies"].params, dec["single"].params), (ec.pose6, es.pose6)):
                assert torch.equal(a_, b_), (call, it)
        for k in eng:
            emb_t[k] = m[k].emb.clone()
    import helpers
    helpers.record_gpu_metric("replicated_accumulators", copies=K, rows_touched=n["copies"])


def test_captured_iteration_with_replicated_accumulators_replays_like_one_copy(golden_dir):
    """capture_iteration's warm-up left gradients in accumulator copies 1..K-1 (it cleared copy 0 only), and the first replay's sweep over the
    touched rows added them.  A captured + replayed iteration with emb_grad_copies = 16 against the same engine flow with one copy: the same embeddings
    after three replays (bf16 rounding of differently associated fp32 sums apart)."""
    import helpers as H
    from oracle import oracle as O
    from nerf_loam_amd import pipeline as P
    g = np.load(os.path.join(golden_dir, "map_1f_3it.npz"))
    sc = H.build_oracle_scene(int(g["n_beams"]), int(g["n_azimuth"]), int(g["seed"]))
    sc["ms"].id2row = g["id_table"].copy()
    masks = H.unpack_masks(g["masks"], len(sc["points"]))
    dec_np = O.decoder_init(int(g["seed"]))
    fr = O.select_rays(sc["points"], sc["cos"], g["poses0"][0].copy(), masks[0][0], optimize_pose=True)
    res = {}
    for copies in (1, 16):
        ms = sc["ms"]
        m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, ms.emb.copy(), ms.voxel_size)
        dec = P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
        eng = P.SdfEngine(max_rays=len(fr.rays_d), samples_per_ray_cap=64, max_frames=2, emb_grad_copies=copies)
        eng.set_rays(fr.rays_d, fr.points, fr.cos); eng.set_poses(fr.pose[None], [1])
        cfg = P.IterConfig(step_size=float(g["step_size"]))
        eng.begin_call(m, dec)
        assert eng.emb_grad_copies == copies
        eng.capture_iteration(m, dec, cfg, train_decoder=False, update_decoder=False, update_pose=False)
        for _ in range(3):
            eng.replay()
        torch.cuda.synchronize()
        assert float(eng._emb_state[:eng.emb_grad_copies * eng._emb_cap * 16].abs().max()) == 0.0        # every copy swept clean
        res[copies] = (m.emb_bits().copy(), int(eng.adam_state[0].item()))
    assert res[1][1] == res[16][1] == 3
    a, b = O.bf16_to_f32(res[1][0]), O.bf16_to_f32(res[16][0])
    moved = np.abs(a - O.bf16_to_f32(sc["ms"].emb)).max()
    assert moved > 0.01 and (res[1][0] != res[16][0]).mean() < 2e-3 and np.abs(a - b).max() <= 0.25 * moved
