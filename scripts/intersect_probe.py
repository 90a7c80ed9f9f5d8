"""time nl_ray_intersect alone at several ray counts (GPU)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P, ops
w = bench.build_workload(torch.device("cuda"))
N = len(w["points"])
eng = P.SdfEngine(max_rays=N, samples_per_ray_cap=48)
m = w["map"]
rng = np.random.default_rng(0)
for n, mode in [(N, "all"), (16384, "first"), (16384, "random"), (2048, "random"), (131072, "shuffled")]:
    if mode == "all": sel = np.arange(N)
    elif mode == "first": sel = np.arange(n)
    elif mode == "shuffled": sel = rng.permutation(N)
    else: sel = np.sort(rng.choice(N, n, replace=False))
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
    def run():
        eng.counters.zero_()
        ops.ray_intersect(eng.N, eng.rays_d_sensor, eng.points_gt, eng.cos_gt, eng.frame_id, eng.poses12, m.blk_hdr, m.blk_ids, m.root_side,
                          m.voxel_size, 50.0, eng.rays_d_world, eng.gt_dist, eng.hit_idx, eng.hit_t0, eng.hit_t1, eng.hit_count, eng.counters, eng.ray_of_rank)
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize()
    print(f"N={n:7d} {mode:9s} {a.elapsed_time(b)/10*1e3:8.1f} us/launch")
