"""Full-scan mapping iteration: stage-wise ctypes calls (forward_backward + optimiser_step) against one C call per iteration (run_bound) against a
hipGraph replay of that call - what the launch path costs at the headline size.  GPU only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P
L.require_gpu()
dev = torch.device("cuda")
w = bench.build_workload(dev)
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])


def timed(fn, n=60, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def stage():
    eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
    eng.optimiser_step(w["map"], w["dec"], cfg, update_decoder=True)


res = {}
for rep in range(2):
    res.setdefault("stage_wise", []).append(timed(stage))
    eng.bind(w["map"], w["dec"], cfg, train_decoder=True, update_decoder=True)
    res.setdefault("one_c_call", []).append(timed(eng.run_bound))
    g = torch.cuda.CUDAGraph()
    eng.run_bound(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        eng.run_bound()
    res.setdefault("graph_replay", []).append(timed(g.replay))
    del g
for k, v in res.items():
    print(f"{k:14s} ms/iteration " + "  ".join(f"{x:.4f}" for x in v))
