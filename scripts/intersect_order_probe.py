"""does the ORDER in which the intersect's workgroups meet cheap and expensive rays matter?  The full scan in scan order, reversed, with its 32-ray chunks
interleaved across beams (stride permutation), and fully shuffled; per-workgroup traversal cycles by position.  GPU only."""
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
chunks = np.arange(N).reshape(-1, 32)
nc = len(chunks)
orders = {"scan order": np.arange(N), "reversed": np.arange(N)[::-1].copy(),
          "chunks interleaved over beams": chunks[(np.arange(nc) % 64) * (nc // 64) + np.arange(nc) // 64].reshape(-1),
          "chunks, stride 1657": chunks[(np.arange(nc) * 1657) % nc].reshape(-1), "rays shuffled": rng.permutation(N)}
for name, sel in orders.items():
    assert len(np.unique(sel)) == N
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
    def run():
        eng.counters.zero_()
        ops.ray_intersect(eng.N, eng.rays_d_sensor, eng.points_gt, eng.cos_gt, eng.frame_id, eng.poses12, m.blk_hdr, m.blk_ids, m.root_side,
                          m.voxel_size, 50.0, eng.rays_d_world, eng.gt_dist, eng.hit_idx, eng.hit_t0, eng.hit_t1, eng.hit_count, eng.counters, eng.ray_of_rank)
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    nb = (N * 8 + 255) // 256
    dbg = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
    L.lib().nl_geometry_set_debug_buffer(L.ptr(dbg)); run(); torch.cuda.synchronize(); L.lib().nl_geometry_set_debug_buffer(None)
    d = dbg.cpu().numpy().reshape(nb, 8)
    life = (d[:, 3] - d[:, 0]).astype(np.float64)
    t0 = d[:, 0].min()
    start, end = (d[:, 0] - t0) / 1e3, (d[:, 3] - t0) / 1e3
    q = np.array_split(np.arange(nb), 8)
    print(f"{name:32s} {a.elapsed_time(b) / 20 * 1e3:7.1f} us/launch | workgroup life k-cycles: mean {life.mean() / 1e3:5.1f} max {life.max() / 1e3:5.1f}; by eighth of the grid: "
          + " ".join(f"{life[i].mean() / 1e3:5.1f}" for i in q) + f" | last start {start.max():6.1f} k, last end {end.max():6.1f} k (stamp clock)", flush=True)
