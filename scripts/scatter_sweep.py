"""k_trilinear_bwd at the launch-bound shapes (a rank's share of an 8-GPU iteration: 16 384 interleaved rays; 2048 rays): HIP-event time
against the number of workgroups (= the span of samples one 8-lane group walks and one wave aggregates in its table).  GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P, ops, dist as D
L.require_gpu()
dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
N = len(w["points"])
lo, hi = D.shard_bounds(N, 0, 8)
rs = np.random.default_rng(5)
shapes = {"rank share 16384": D.interleaved_order(N, 8)[lo:hi], "2048 rays": np.sort(rs.choice(N, 2048, replace=False)), "full scan": np.arange(N)}
for n in (4096, 8192, 32768, 65536):
    shapes[f"{n} rays"] = np.sort(rs.choice(N, n, replace=False))
BLOCKS = [int(b) for b in os.environ.get("BLOCKS", "128,256,512,768,1024,1536,2048,3072,4096").split(",")]
m = w["map"]
pose0 = w["pose"]
if "--large-map" in sys.argv:                                     # the 150-scan map of bench.py large_map: a ray crosses many more voxels, ~1 sample each
    lm = bench.build_large_map(w, dev)
    m, pose0 = lm["map"], lm["poses"][len(lm["poses"]) // 2]
    shapes = {k: v for k, v in shapes.items() if k != "full scan"}
for name, sel in shapes.items():
    eng = P.SdfEngine(max_rays=len(sel), samples_per_ray_cap=96 if "--large-map" in sys.argv else 48, device=dev)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(pose0[None], [1])
    cfg = P.IterConfig(); eng.begin_call(m, w["dec"])
    for _ in range(2):
        eng.forward_backward(m, w["dec"], cfg, train_decoder=True)
    Pn = eng.stats()["P"]
    def run(blocks, touched):
        ops.trilinear_bwd(eng.loss_scalars, eng.s_vox, eng.s_depth, eng.s_ray, eng.rays_d_world, eng.rays_d_sensor, eng.frame_id,
                          eng.poses12, eng.F, m.centres, m.vertex_rows, m.emb, m.voxel_size, eng.dX, eng.g_emb, eng.g_pose, blocks,
                          eng._touched if touched else None)
    print(f"{name}: P = {Pn}, engine launches {2 * eng.field_blocks} workgroups")
    if "--probes" in sys.argv:                                    # the product's launch under 2 .. 16 open-addressing probes per insert
        for pr in (1, 2, 4, 8, 16):
            L.lib().nl_field_set_probes(pr)
            run(2 * eng.field_blocks, 1); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run(2 * eng.field_blocks, 1)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            print(f"  probes {pr:2d}: {best:7.1f} us")
        L.lib().nl_field_set_probes(16)
        for fl in (-1, 0, 1, 2, 3, 4):                             # mid-span write-out of a full table: never / with >= fl sample steps left
            L.lib().nl_field_set_midspan_flush(fl)
            run(2 * eng.field_blocks, 1); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run(2 * eng.field_blocks, 1)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            print(f"  mid-span flush {fl:2d}: {best:7.1f} us")
        L.lib().nl_field_set_midspan_flush(2)
        nb = 2 * eng.field_blocks                                  # per-phase cycle stamps of every workgroup (scripts/scatter_probe.py)
        dbg = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
        L.lib().nl_field_set_debug_buffer(L.ptr(dbg)); run(nb, 1); torch.cuda.synchronize(); L.lib().nl_field_set_debug_buffer(None)
        d = dbg.cpu().numpy().reshape(nb, 8); d = d[(d[:, [0, 1, 2, 3, 5]] > 0).all(1)]
        ph = np.diff(d[:, [0, 1, 2, 3, 5]], axis=1)
        tot = d[:, 5] - d[:, 0]
        print("  cycles per workgroup (%d active): " % len(d) + ", ".join(f"{n} {v:.0f}" for n, v in zip(("init", "sample loop", "last run flush", "table flush + touched rows"), ph.mean(0)))
              + "; whole workgroup: median %.0f, 90 %% %.0f, 99 %% %.0f, max %.0f; table flush max %.0f (the counters of different XCDs are not comparable: no kernel span)"
              % (np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), ph[:, 3].max()))
        continue
    L.lib().nl_field_set_one_round(0)
    for blocks in BLOCKS + [-1]:
        if blocks < 0:                                            # the product's launch: 2 x field_blocks workgroups, one-round rule on
            L.lib().nl_field_set_one_round(1); blocks = 2 * eng.field_blocks
        ts = []
        for touched in (0, 1):
            run(blocks, touched); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run(blocks, touched)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            ts.append(best)
        print(f"  blocks {blocks:5d}  samples / group {max(-(-Pn // (blocks * 32)), 2):4d}: {ts[0]:7.1f} us   with the touched-rows record {ts[1]:7.1f} us")
