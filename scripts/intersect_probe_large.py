"""nl_ray_intersect alone on the 150-scan map (bench.build_large_map): time per launch, rounds, share of rays that were started again /
compacted (workgroup thread 0's ray), for both push orders. GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P, ops
dev = torch.device("cuda")
w = bench.build_workload(dev)
lm = bench.build_large_map(w, dev, n_scans=int(os.environ.get("N_SCANS", "150")))
print("map of", os.environ.get("N_SCANS", "150"), "scans:", int(lm["map"].blk_hdr.shape[0]), "children blocks")
N = len(w["points"])
rng = np.random.default_rng(0)
for tag, m, pose in (("single", w["map"], w["pose"]), ("large", lm["map"], lm["poses"][len(lm["poses"]) // 2])):
    for flags in (1,):
        L.lib().nl_geometry_set_intersect_prune(flags)
        for n in ((2048, 16384) if os.environ.get("N_SCANS") else (2048, 16384, N)):
            sel = np.arange(N) if n == N else np.sort(rng.choice(N, n, replace=False))
            eng = P.SdfEngine(max_rays=n, samples_per_ray_cap=4)
            eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(pose[None], [1])
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
            lpr = int(os.environ.get("NL_LANES_PER_RAY") or 0) or (16 if n <= 16384 else 8)      # (the stamps: one row per workgroup)
            nb = (n * lpr + 255) // 256
            dbg = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
            L.lib().nl_geometry_set_debug_buffer(L.ptr(dbg)); run(); torch.cuda.synchronize(); L.lib().nl_geometry_set_debug_buffer(None)
            d = dbg.cpu().numpy().reshape(nb, 8)
            ovf = int(eng.counters[L.NLC_ISECT_OVF].item())
            print(f"{tag:6s} flags {flags} N={n:7d} {a.elapsed_time(b)/10*1e3:8.1f} us | rounds mean {d[:,4].mean():6.1f} max {d[:,4].max():4d} | "
                  f"restarted {d[:,5].mean()*100:5.1f} %  compactions/ray {d[:,6].mean():5.2f} max {d[:,6].max()}  fallback rays {ovf}")
L.lib().nl_geometry_set_intersect_prune(1)
