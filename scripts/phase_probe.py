"""Per-phase shader-clock breakdown of k_decoder (workgroup 0) on the bench workload. GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P
L.require_gpu()
w = bench.build_workload(torch.device("cuda"))
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
dbg = torch.zeros(256 + 8 * 1024, dtype=torch.int64, device="cuda")     # (+ k_decoder2's per-workgroup records)
for train in (True, False):
    for _ in range(2):
        eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train)
    L.lib().nl_decoder_set_debug_buffer(L.ptr(dbg))
    eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train)
    torch.cuda.synchronize()
    L.lib().nl_decoder_set_debug_buffer(None)
    d = dbg.cpu().numpy()[:256].reshape(16, 16)[:, :11]
    names = ["A:loadX", "B:H1", "C:loop", "C:epi", "D:loss", "E:dH2", "F:loop", "F:epi", "H:dH1", "I:L1bwd"]
    ph = np.diff(d[2:10], axis=1)
    print("train" if train else "frozen", "cycles/phase (mean over tiles 2..9):")
    for n, v in zip(names, ph.mean(0)):
        print(f"  {n:10s} {v:10.0f}")
    print("  total/tile", (d[3:10, 0] - d[2:9, 0]).mean())
