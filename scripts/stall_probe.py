#!/usr/bin/env python3
"""where do the ~60 ms pauses in long launch-bound loops come from?  One C call per step (nl_iteration), an event after every step and the host
clock around every call: per-step device intervals (event to event) and host call times - a device-side gap without a slow host call = the device
stalled; a slow host call = the runtime blocked the launching thread.  `python scripts/stall_probe.py [full|track] [steps]`"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from nerf_loam_amd import pipeline as P                          # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "track"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
N = len(w["points"])
if mode == "full":
    eng = P.SdfEngine(max_rays=N, samples_per_ray_cap=48, device=dev)
    eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
    eng.begin_call(w["map"], w["dec"])
    eng.bind(w["map"], w["dec"], P.IterConfig(), train_decoder=True)
else:
    sel = np.sort(np.random.default_rng(3).choice(N, 2048, replace=False))
    eng = P.SdfEngine(max_rays=2048, samples_per_ray_cap=96, device=dev)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
    eng.begin_call(w["map"], None, emb_state=False)
    eng.bind(w["map"], w["dec"], P.IterConfig(step_size=0.04), train_decoder=False, want_emb_grad=False, want_pose_grad=True, update_emb=False,
             update_decoder=False, update_pose=True, skip_mode=2)
for _ in range(20):
    eng.run_bound()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = np.zeros(steps)
ev[0].record()
t_all = time.perf_counter()
for k in range(steps):
    t0 = time.perf_counter()
    eng.run_bound()
    ev[k + 1].record()
    host[k] = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
devi = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)])
print(f"{mode}: {steps} steps in {t_all * 1e3:.1f} ms; device interval median {np.median(devi):.4f} ms, max {devi.max():.3f} ms at step {int(devi.argmax())}; "
      f"host call median {np.median(host) * 1e3:.4f} ms, max {host.max() * 1e3:.3f} ms at step {int(host.argmax())}")
big = np.nonzero(devi > 5 * np.median(devi) + 0.5)[0]
print("device intervals > 5x median:", [(int(k), round(float(devi[k]), 2)) for k in big[:20]])
bigh = np.nonzero(host > 2e-3)[0]
print("host calls > 2 ms:", [(int(k), round(float(host[k]) * 1e3, 2)) for k in bigh[:20]])
