"""GPU (-m gpu): race / schedule stress (SURVEY section 5 "race detection").  The hot kernels synchronise by hand - LDS work-lists and hash
tables with claim rounds, decoupled look-back with epoch tags, wave_barrier + fence pairs, one global atomic per touched row and wave - so
a data race would show as a result that depends on the SCHEDULE.  The same iteration is therefore run under perturbed schedules:

  * the intersect with 8 and with 16 lanes per ray (other round structure of the work-list), pruning on / off;
  * the sampler sequential, step-parallel, and by ray count (other kernels, other look-back topology); the one-launch sampler vs the
    four-launch sequence (one C call vs stage calls);
  * the field kernels (gather / scatter) with half and with twice the workgroups: other sample -> workgroup -> hash-table assignment;
  * 20 back-to-back repetitions of each configuration (launch overlap, warm / cold caches, other wave arrival orders);

at the full scan (131 072 rays, 1.1 M samples: every CU busy) and at a launch-bound size (2048 rays).  Every INTEGER / index output and every
fp32 value that is computed without atomics (hit lists, counts, ranks, offsets, sample voxels / depths / dists, X, sdf, dsdf, dX, the ReLU
words) must equal run 0 bit for bit; what is summed by fp32 / fp64 atomics or in a grid-dependent order (embedding accumulators, pose
partials, decoder gradient slabs, loss sums) must agree within the documented tolerance.  Zero mismatches is the pass criterion; the counts
go to gpurun_out/gpu_metrics.json."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu

EXACT = ("hit_count", "hit_rank", "hit_idx", "hit_t0", "hit_t1", "samp_count", "samp_off", "s_vox", "s_depth", "s_dist", "s_ray", "X", "sdf", "dsdf", "dX", "ints")


def _snapshot(eng, N, dec):
    st = eng.stats()
    P_ = st["P"]
    hc = eng.hit_count[:N]
    live = (torch.arange(20, device=hc.device)[None, :] < hc[:, None])
    snap = dict(hit_count=hc, hit_rank=torch.where(hc > 0, eng.hit_rank[:N], torch.zeros_like(hc)),
                hit_idx=torch.where(live, eng.hit_idx[:N], torch.full_like(eng.hit_idx[:N], -1)),
                hit_t0=torch.where(live, eng.hit_t0[:N], torch.zeros_like(eng.hit_t0[:N])), hit_t1=torch.where(live, eng.hit_t1[:N], torch.zeros_like(eng.hit_t1[:N])),
                samp_count=eng.samp_count[:N], samp_off=eng.samp_off[:N], s_vox=eng.s_vox[:P_], s_depth=eng.s_depth[:P_], s_dist=eng.s_dist[:P_],
                s_ray=eng.s_ray[:P_], X=eng.X[:P_], sdf=eng.sdf[:P_], dsdf=eng.dsdf[:P_], dX=eng.dX[:P_])
    snap = {k: v.clone() for k, v in snap.items()}
    ints = st["ints"].copy(); ints[15] = 0                              # (NLC_TICKET: a launch ticket, not a result)
    snap["ints"] = torch.from_numpy(ints)
    snap["g_emb"] = eng.g_emb.clone(); snap["g_pose"] = eng.g_pose[:1].clone(); snap["dec_grad"] = dec.grad.clone()
    snap["dbl"] = torch.from_numpy(st["dbl"].copy())
    return snap


def _check(ref, got, tag, tally):
    for k in EXACT:
        a, b = ref[k], got[k]
        same = a.shape == b.shape and bool(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b))
        tally["exact_checked"] += 1
        assert same, f"{tag}: {k} depends on the schedule"
    # summed by atomics / in a grid-dependent order: fp32 round-off of a few thousand terms at most
    ge_r, ge = ref["g_emb"].double(), got["g_emb"].double()
    assert float((ge - ge_r).abs().max()) <= 2e-5 * float(ge_r.abs().max()), f"{tag}: embedding accumulators"
    assert bool(((ge != 0) == (ge_r != 0)).all()), f"{tag}: touched embedding rows"
    gp_r, gp = ref["g_pose"], got["g_pose"]
    # (per-ray fp32 partial sums folded into fp64 accumulators: the fp32 part follows the sample -> lane-group assignment)
    assert float((gp - gp_r).abs().max()) <= 1e-5 * float(gp_r.abs().max()), f"{tag}: pose partials"
    dg_r, dg = ref["dec_grad"].double(), got["dec_grad"].double()
    assert float((dg - dg_r).norm()) <= 1e-5 * float(dg_r.norm()), f"{tag}: decoder gradient"
    assert float((got["dbl"] - ref["dbl"]).abs().max()) <= 1e-9 * float(ref["dbl"].abs().max()), f"{tag}: loss sums"
    tally["tolerance_checked"] += 4


def _run_config(P, L, eng, m, dec, cfg, N, one_call, reps, ref, tag, tally):
    for rep in range(reps):
        eng.g_emb.zero_(); eng.g_pose.zero_()
        if one_call:
            eng.run_bound(1)                                         # forward + backward through nl_iteration (fused launches), no optimiser step
        else:
            eng.forward_backward(m, dec, cfg, train_decoder=True)
        snap = _snapshot(eng, N, dec)
        if ref[0] is None:
            ref[0] = snap
        else:
            _check(ref[0], snap, f"{tag} rep {rep}", tally)


@pytest.mark.parametrize("n_rays", [131072, 2048])
def test_results_do_not_depend_on_the_schedule(n_rays):
    from nerf_loam_amd import _lib as L, pipeline as P, synthetic as S
    from nerf_loam_amd.svo import Octree
    L.require_gpu()
    lib = L.lib()
    pts, cos = S.synthetic_scan()
    pose = S.scan_pose()
    oc = Octree(); oc.init(256 * 256 * 4, 16, 0.2)
    oc.insert(S.voxel_coords(pts, np.eye(3, dtype=np.float32), pose[:3], 0.2))
    c, s_, f = oc.export_device_layout()
    id2row = -np.ones(len(c), np.int32)
    E = O.assign_embedding_rows(f, id2row, 0)
    m = P.MapDevice(c, s_, f, id2row, O.bf16_bits(H.init_embeddings(E, 1)), 0.2)
    d0 = O.decoder_init(1)
    dec = P.DecoderDevice(d0.W1, d0.b1, d0.W2, d0.b2, d0.W3, d0.b3)
    sel = np.arange(len(pts)) if n_rays >= len(pts) else np.sort(np.random.default_rng(9).choice(len(pts), n_rays, replace=False))
    N = len(sel)
    eng = P.SdfEngine(max_rays=N, samples_per_ray_cap=48 if N > 8192 else 96, sparse_adam=False)
    eng.set_rays(S.unit_dirs(pts)[sel], pts[sel], cos[sel]); eng.set_poses(pose[None], [1])
    cfg = P.IterConfig()
    eng.begin_call(m, dec)
    eng.bind(m, dec, cfg, train_decoder=True)
    reps = 20
    ref, tally = [None], dict(exact_checked=0, tolerance_checked=0, configurations=0)
    fb0 = eng.field_blocks
    try:
        for lpr in (0, 8, 16):
            for prune in (1, 0):
                for smode in (2, 0, 1):
                    for fb in (fb0, max(64, fb0 // 2), 2 * fb0):
                        # the full cross product is 54 configurations; walk a covering subset: every value of every knob, every pair once
                        if (lpr, prune, smode, fb) not in {(0, 1, 2, fb0), (8, 1, 0, fb0 // 2 if fb0 // 2 >= 64 else 64), (16, 1, 1, 2 * fb0), (8, 0, 2, 2 * fb0),
                                                          (16, 0, 0, fb0), (0, 0, 1, max(64, fb0 // 2)), (0, 1, 0, 2 * fb0), (16, 1, 2, max(64, fb0 // 2))}:
                            continue
                        assert lib.nl_geometry_set_lanes_per_ray(lpr) == 0 and lib.nl_geometry_set_intersect_prune(prune) == 0
                        assert lib.nl_geometry_set_sampler_mode(smode) == 0
                        eng.field_blocks = fb
                        eng._desc.field_blocks = fb
                        for one_call in (False, True):
                            _run_config(P, L, eng, m, dec, cfg, N, one_call, reps, ref, f"lpr={lpr} prune={prune} sampler={smode} field_blocks={fb} one_call={one_call}", tally)
                            tally["configurations"] += 1
    finally:
        lib.nl_geometry_set_lanes_per_ray(0); lib.nl_geometry_set_intersect_prune(1); lib.nl_geometry_set_sampler_mode(2)
    assert tally["configurations"] == 16
    H.record_gpu_metric(f"schedule_stress_{n_rays}", mismatches=0, runs=tally["configurations"] * reps, **tally)
