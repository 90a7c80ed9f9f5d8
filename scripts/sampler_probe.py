"""cycle stamps of the sampler kernels (wave 0 of every workgroup) on 2048 rays of the bench scan. GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P, ops
L.require_gpu()
w = bench.build_workload(torch.device("cuda"))
rng = np.random.default_rng(3)
for n, step in ((2048, 0.04), (16384, 0.1)):
    sel = np.sort(rng.choice(len(w["points"]), n, replace=False))
    eng = P.SdfEngine(max_rays=n, samples_per_ray_cap=96)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
    cfg = P.IterConfig(step_size=step); eng.begin_call(w["map"], w["dec"])
    for _ in range(2): eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=False, want_emb_grad=False)
    nb = (n + 255) // 256
    c = eng.counters
    seed, N = 0, eng.N
    args = (N, eng.hit_idx, eng.hit_t0, eng.hit_t1, eng.hit_count, eng.hit_rank, eng.ray_of_rank, eng.cos_gt, eng.gt_dist,
            cfg.step_size, cfg.truncation, cfg.max_distance, seed, 0, int(cfg.tail_always), 0, None, None, c, eng.samp_count)
    for emit in (0, 1):
        dbg = torch.zeros(max(nb, 64) * 8, dtype=torch.int64, device="cuda")
        L.lib().nl_geometry_set_debug_buffer(L.ptr(dbg))
        if emit: ops.sample_rays(1, *args, eng.samp_off, eng.P_cap, eng.s_vox, eng.s_depth, eng.s_dist, eng.s_ray)
        else: ops.sample_rays(0, *args, None, eng.P_cap, None, None, None, None)
        torch.cuda.synchronize(); L.lib().nl_geometry_set_debug_buffer(None)
        d = dbg.cpu().numpy().reshape(-1, 8)[:nb]
        ph = np.diff(d[:, :4], axis=1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            if emit: ops.sample_rays(1, *args, eng.samp_off, eng.P_cap, eng.s_vox, eng.s_depth, eng.s_dist, eng.s_ray)
            else: ops.sample_rays(0, *args, None, eng.P_cap, None, None, None, None)
        b.record(); torch.cuda.synchronize()
        print(f"n={n} step={step} emit={emit}: {a.elapsed_time(b) / 10 * 1e3:.1f} us/launch; cycles: loads+tot {ph[:,0].mean():.0f}, layout/row-first {ph[:,1].mean():.0f}, "
              f"walk {ph[:,2].mean():.0f} (max {ph[:,2].max():.0f}); samples/ray mean {eng.stats()['P'] / max(eng.stats()['R'], 1):.1f} max {eng.stats()['S']}")
