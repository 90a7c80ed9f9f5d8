"""The prefix scans between the intersect / count pass and the sampler at a rank's share of a scan (4097 .. 32 768 rays): ONE launch
(nl_geometry.hip k_scan_single: every workgroup sums the items in front of it itself; its last workgroup also writes the loss scalars or the
send block of the ray-sharded iteration's first exchange) against the two launches of larger scans and against numpy."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import oracle as O
from test_gpu_parity import _ThreadRanks, make_engine, nl           # noqa: F401  (the virtual-rank rig, the library fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture
def scan_switch(nl):
    lib = nl["L"].lib()
    yield lib.nl_geometry_set_scan_single
    lib.nl_geometry_set_scan_single(1)


@pytest.mark.parametrize("n", [4096, 4097, 5000, 8192, 12345, 16384, 20481, 32768, 32769, 70000])
@pytest.mark.parametrize("flag_mode", [0, 1])
def test_exclusive_scan_sizes_and_paths(nl, scan_switch, n, flag_mode):
    """nl_exclusive_scan_i32 around the boundaries of its three launch shapes (one workgroup <= 4096 < one launch of <= 8 workgroups <= 32 768 <
    two launches), ragged tails included: equal to numpy and, where the single-launch kernel applies, to the two-launch path"""
    ops = nl["ops"]
    rng = np.random.default_rng(n + flag_mode)
    a = rng.integers(-2 if flag_mode else 0, 40, n).astype(np.int32)
    a[rng.random(n) < 0.3] = 0
    want = np.concatenate([[0], np.cumsum((a > 0).astype(np.int64) if flag_mode else a.astype(np.int64))])
    inp = torch.from_numpy(a).cuda()
    ws = torch.zeros(1024, dtype=torch.int32, device="cuda")
    got = {}
    for single in (1, 0):
        assert scan_switch(single) == 0
        out = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        tot = torch.full((1,), -7, dtype=torch.int32, device="cuda")
        ops.exclusive_scan(inp, out, n, flag_mode, tot, ws)
        torch.cuda.synchronize()
        got[single] = (out.cpu().numpy(), int(tot.item()))
        assert np.array_equal(got[single][0], want[:-1]) and got[single][1] == want[-1], (n, flag_mode, single)
    assert np.array_equal(got[0][0], got[1][0])


def _share_scene(extra=0):
    """16 384 rays (16 beams x 1024 azimuth steps) on their own one-scan map: a rank's share of the 64 x 2048 scan in size (+ `extra` rays repeated)"""
    sc = H.build_oracle_scene(16, 1024, 5)
    from nerf_loam_amd import synthetic as S
    pose = S.scan_pose()
    fr = O.select_rays(sc["points"], sc["cos"], pose.copy(), np.ones(len(sc["points"]), bool), optimize_pose=True)
    if extra:
        fr = O.Frame(np.concatenate([fr.rays_d, fr.rays_d[:extra]]), np.concatenate([fr.points, fr.points[:extra]]),
                     np.concatenate([fr.cos, fr.cos[:extra]]), fr.pose, True)
    return sc, fr


def _run_bound_once(nl, sc, fr, lo, hi, install=None, stage_wise=False):
    P = nl["P"]
    dec_np = O.decoder_init(5)
    m, dec, eng = make_engine(nl, sc, dec_np, hi - lo, 1)
    ex = install(eng) if install else None
    eng.set_rays(fr.rays_d[lo:hi], fr.points[lo:hi], fr.cos[lo:hi], np.zeros(hi - lo, np.int32))
    eng.set_poses(fr.pose[None], [1])
    cfg = P.IterConfig(step_size=0.1, noise_seed=11)
    eng.begin_call(m, dec)
    if stage_wise:
        eng.forward_backward(m, dec, cfg, train_decoder=True, ray_id_base=lo)
    else:
        eng.bind(m, dec, cfg, train_decoder=True, ray_id_base=lo)
        eng.run_bound(1)
    torch.cuda.synchronize()
    st = eng.stats()
    if ex is not None:
        ex.reduce_loss_sums()
    n = hi - lo
    out = dict(P=st["P"], R=st["R"], S=st["S"], hit_rank=eng.hit_rank[:n].cpu().numpy(), ray_of_rank=eng.ray_of_rank[:st["R"]].cpu().numpy(),
               samp_off=eng.samp_off[:n].cpu().numpy(), ls=eng.loss_scalars.cpu().numpy().copy(), sdf=eng.sdf[:st["P"]].cpu().numpy(),
               depth=eng.s_depth[:st["P"]].cpu().numpy(), vox=eng.s_vox[:st["P"]].cpu().numpy(), gdec=dec.grad.cpu().numpy().copy(),
               gemb=eng.g_emb.cpu().numpy().copy(), gpose=eng.g_pose.cpu().numpy().copy(), loss=eng.loss_value(cfg)["loss"])
    assert not eng.call_status()[2]
    return out


def test_one_launch_scans_equal_the_two_launch_path_in_the_iteration(nl, scan_switch):
    """a 16 384-ray iteration through nl_iteration (hit-ray scan with compaction; sample-offset scan with the loss scalars written by the scan's
    last workgroup) against the same iteration with the two-launch scans + nl_loss_finalize, and against the stage-wise sequence: every index
    array, the loss scalars, sdf and gradients bit for bit"""
    sc, fr = _share_scene()
    N = len(fr.rays_d)
    assert 8192 < N <= 32768
    scan_switch(1)
    a = _run_bound_once(nl, sc, fr, 0, N)
    s = _run_bound_once(nl, sc, fr, 0, N, stage_wise=True)
    scan_switch(0)
    b = _run_bound_once(nl, sc, fr, 0, N)
    assert a["R"] > 12000 and a["P"] > 100000
    for other in (b, s):
        for k in ("P", "R", "S"):
            assert a[k] == other[k], k
        for k in ("hit_rank", "ray_of_rank", "samp_off", "ls", "depth", "vox", "sdf", "gdec"):
            assert np.array_equal(a[k], other[k]), k
        # (the embedding accumulators and the fp64 pose partials take their atomics in whatever order the workgroups arrive)
        assert np.abs(a["gemb"] - other["gemb"]).max() <= 1e-6 * np.abs(a["gemb"]).max()
        np.testing.assert_allclose(a["gpose"], other["gpose"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("world,extra", [(2, 0), (3, 0), (2, 1)])
def test_virtual_ranks_with_the_send_block_packed_by_the_scan(nl, monkeypatch, scan_switch, world, extra):
    """ray shards of 8192 (world 2), of 5462 / 5462 / 5460 rays (world 3: ragged) and of 8193 / 8192 (the second rank's scan covers 8192 items, the
    send block holds 8208 bytes: its tail is zeroed by the last workgroup) through nl_iteration: the hit-ray scan's launch packs exchange 1's send
    block (nl_ray_intersect_scan_x1) - samples bit-identical to the unsharded iteration, gradients identical on every rank and to round-off of the
    unsharded ones; then the same with the pack as its own launch: bit-identical"""
    import threading
    from nerf_loam_amd import dist as D
    sc, fr = _share_scene(extra)
    N = len(fr.rays_d)
    one = _run_bound_once(nl, sc, fr, 0, N)
    fake = _ThreadRanks(world)
    monkeypatch.setattr(D, "dist", fake)

    def sharded():
        res, errs = [None] * world, []

        def worker(r):
            try:
                fake.tl.rank = r
                torch.cuda.set_device(0)
                lo, hi = D.shard_bounds(N, r, world)
                assert 4096 < hi - lo <= 32768
                res[r] = _run_bound_once(nl, sc, fr, lo, hi, install=lambda eng: D.RayShardedExchange(eng, sparse_rows=False, overlap=True))
            except Exception:                                        # noqa: BLE001
                import traceback
                errs.append(traceback.format_exc()); fake.bar.abort()
        th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join(300) for t in th]
        assert not errs, errs
        return res

    scan_switch(1)
    res = sharded()
    assert sum(x["P"] for x in res) == one["P"] and sum(x["R"] for x in res) == one["R"] and all(x["S"] == one["S"] for x in res)
    assert np.array_equal(np.concatenate([x["depth"] for x in res]), one["depth"])
    assert np.array_equal(np.concatenate([x["vox"] for x in res]), one["vox"])
    assert np.abs(np.concatenate([x["sdf"] for x in res]) - one["sdf"]).max() < 1e-6
    a = res[0]
    np.testing.assert_allclose(a["loss"], one["loss"], rtol=1e-6)
    for b in res[1:]:
        for k in ("gdec", "gemb", "gpose"):
            assert np.array_equal(a[k], b[k]), k
    for k, tol in (("gdec", 1e-5), ("gemb", 1e-5), ("gpose", 1e-6)):
        assert np.linalg.norm(a[k].astype(np.float64) - one[k]) <= tol * np.linalg.norm(one[k].astype(np.float64)), k
    scan_switch(0)
    res2 = sharded()
    for x, y in zip(res, res2):
        np.testing.assert_allclose(x["gpose"], y["gpose"], rtol=1e-9, atol=1e-12)
        for k in ("hit_rank", "ray_of_rank", "samp_off", "ls", "depth", "vox", "sdf", "gdec"):
            assert np.array_equal(x[k], y[k]), k


def test_iteration_records_the_callers_timing_events(nl):
    """NlIterDesc.ev_decoder_begin / ev_decoder_end / ev_wgrad2_end: nl_iteration records the caller's events around the decoder kernel and
    behind dW2 (bench.py's timed region: one C call per step) - the intervals are those kernels' (positive, far below the call), the results do
    not depend on the events, and a later call without timers records nothing"""
    P = nl["P"]
    sc, fr = _share_scene()
    N = len(fr.rays_d)
    m, dec, eng = make_engine(nl, sc, O.decoder_init(5), N, 1)
    eng.set_rays(fr.rays_d, fr.points, fr.cos, np.zeros(N, np.int32)); eng.set_poses(fr.pose[None], [1])
    cfg = P.IterConfig(step_size=0.1, noise_seed=11)
    eng.begin_call(m, dec)
    eng.bind(m, dec, cfg, train_decoder=True)
    eng.run_bound(1)
    torch.cuda.synchronize()
    sdf0, g0 = eng.sdf[:eng.stats()["P"]].clone(), dec.grad.clone()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for x in e:
        x.record()                                              # (creates the handle)
    torch.cuda.synchronize()
    eng.timers = {"decoder": (e[0], e[1]), "wgrad2": (e[1], e[2])}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); eng.run_bound(1); t1.record()
    torch.cuda.synchronize()
    whole, d_ms, w_ms = t0.elapsed_time(t1), e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    assert 0.005 < d_ms < whole and 0.003 < w_ms < whole and d_ms + w_ms < whole, (whole, d_ms, w_ms)
    assert torch.equal(eng.sdf[:eng.stats()["P"]], sdf0) and torch.equal(dec.grad, g0)
    eng.timers = None
    eng.run_bound(1)
    torch.cuda.synchronize()
    d = eng._desc
    assert not d.ev_decoder_begin and not d.ev_decoder_end and not d.ev_wgrad2_end
    assert e[0].elapsed_time(e[1]) == d_ms                       # (not recorded again)
