"""count + emit pass of the sampler, sequential (one lane per ray) against step-parallel (eight lanes per ray), per ray count; and the
one-launch sampler where it applies.  GPU only.  us per launch (HIP events over 20 launches)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P, ops
L.require_gpu()
w = bench.build_workload(torch.device("cuda"))
rng = np.random.default_rng(3)


def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, step in ((2048, 0.04), (4096, 0.1), (8192, 0.1), (16384, 0.1), (32768, 0.1), (131072, 0.1)):
    sel = np.sort(rng.choice(len(w["points"]), n, replace=False)) if n < len(w["points"]) else np.arange(n)
    eng = P.SdfEngine(max_rays=n, samples_per_ray_cap=96 if n <= 16384 else 16)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
    cfg = P.IterConfig(step_size=step); eng.begin_call(w["map"], w["dec"])
    for _ in range(2): eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=False, want_emb_grad=False)
    ref = [t.clone() for t in (eng.samp_count, eng.s_vox, eng.s_depth, eng.s_dist, eng.s_ray)]
    c = eng.counters
    args = (eng.N, eng.hit_idx, eng.hit_t0, eng.hit_t1, eng.hit_count, eng.hit_rank, eng.ray_of_rank, eng.cos_gt, eng.gt_dist,
            cfg.step_size, cfg.truncation, cfg.max_distance, 0, 0, int(cfg.tail_always), 0, None, None, c, eng.samp_count)
    out = []
    for mode in (0, 1):
        L.lib().nl_geometry_set_sampler_mode(mode)
        tc = timed(lambda: ops.sample_rays(0, *args, None, eng.P_cap, None, None, None, None))
        te = timed(lambda: ops.sample_rays(1, *args, eng.samp_off, eng.P_cap, eng.s_vox, eng.s_depth, eng.s_dist, eng.s_ray))
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, (eng.samp_count, eng.s_vox, eng.s_depth, eng.s_dist, eng.s_ray)))
        out.append(f"{'sequential' if mode == 0 else 'step-parallel'}: count {tc:.1f} emit {te:.1f} same={same}")
    L.lib().nl_geometry_set_sampler_mode(2)
    st = eng.stats()
    print(f"n={n}: " + "; ".join(out) + f"; samples/ray {st['P'] / max(st['R'], 1):.1f}", flush=True)
