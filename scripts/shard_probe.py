"""What the ranks of a K-GPU run do, measured on ONE GPU: the full iteration (no exchanges) on each rank's rays of the bench
scan, for contiguous blocks of the beam-major scan and for the interleaved order of nerf_loam_amd.dist.interleaved_order.
max over ranks of T bounds the strong-scaling efficiency (the exchanges come on top).  GPU only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P, dist as D
L.require_gpu()
w = bench.build_workload(torch.device("cuda"))
N = len(w["points"])
cfg = P.IterConfig()

def t_iter(sel):
    eng = P.SdfEngine(max_rays=len(sel), samples_per_ray_cap=48)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
    eng.begin_call(w["map"], w["dec"])
    def step():
        eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
        eng.optimiser_step(w["map"], w["dec"], cfg)
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(12): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 12 * 1e3, eng.stats()["P"]

t1, p1 = t_iter(np.arange(N))
print(f"K=1: {t1:.3f} ms, {p1} samples")
for K in (2, 4, 8):
    for name, order in (("contiguous", np.arange(N)), ("interleaved", D.interleaved_order(N, K))):
        ts, ps = [], []
        for r in range(K):
            lo, hi = D.shard_bounds(N, r, K)
            t, p = t_iter(order[lo:hi]); ts.append(t); ps.append(p)
        print(f"K={K} {name:11s}: per-rank ms {' '.join(f'{t:.3f}' for t in ts)} | samples min {min(ps)} max {max(ps)} | "
              f"T(1)/max T = {t1 / max(ts):.2f} (ideal {K})")
