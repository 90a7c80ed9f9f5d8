"""per-step time of the bench iteration from a cold start: how long the device takes to reach its steady clock (HIP events around every step)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P
L.require_gpu()
dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48, device=dev)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
torch.cuda.synchronize(); time.sleep(2.0)                     # idle: clocks fall back
n = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for k in range(n):
    eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
    eng.optimiser_step(w["map"], w["dec"], cfg, update_decoder=True)
    ev[k + 1].record()
torch.cuda.synchronize()
t = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(n)])
for a, b in ((0, 1), (1, 2), (2, 5), (5, 10), (10, 25), (25, 50), (50, 100), (100, 200), (200, 400)):
    print(f"steps {a:3d}..{b:3d}: mean {t[a:b].mean():.4f} ms  min {t[a:b].min():.4f}  (cumulative time {t[:b].sum():7.1f} ms)")
