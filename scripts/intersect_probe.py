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
LPR = int(os.environ.get("LPR", "0"))                      # 0 = the library's choice by ray count
L.lib().nl_geometry_set_lanes_per_ray(LPR)
print("lanes per ray:", LPR or "auto (16 up to 16384 rays, else 8)")
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
    nb = (n * (LPR or (16 if n <= 16384 else 8)) + 255) // 256
    dbg = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
    L.lib().nl_geometry_set_debug_buffer(L.ptr(dbg)); run(); torch.cuda.synchronize(); L.lib().nl_geometry_set_debug_buffer(None)
    d = dbg.cpu().numpy().reshape(nb, 8)
    ph = np.diff(d[:, :4], axis=1)
    print(f"N={n:7d} {mode:9s} {a.elapsed_time(b)/10*1e3:8.1f} us/launch | cycles of wave 0: set-up {ph[:,0].mean():.0f}, traversal {ph[:,1].mean():.0f} "
          f"(max {ph[:,1].max():.0f}; rounds mean {d[:,4].mean():.1f} max {d[:,4].max()}), finalise {ph[:,2].mean():.0f} (max {ph[:,2].max():.0f})")
