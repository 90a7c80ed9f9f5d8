#!/usr/bin/env python3
"""does the device stall when load resumes after seconds of idle?  One-C-call steps with an event after every step, started (a) right after
other GPU work, (b) after `idle` seconds of host sleep: device intervals per step.  `python scripts/wake_probe.py [track|full] [idle_s]`"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from nerf_loam_amd import pipeline as P                          # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "track"
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
N = len(w["points"])
if mode == "full":
    eng = P.SdfEngine(max_rays=N, samples_per_ray_cap=48, device=dev)
    eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
    eng.begin_call(w["map"], w["dec"])
    eng.bind(w["map"], w["dec"], P.IterConfig(), train_decoder=True)
    steps = 80
else:
    sel = np.sort(np.random.default_rng(3).choice(N, 2048, replace=False))
    eng = P.SdfEngine(max_rays=2048, samples_per_ray_cap=96, device=dev)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
    eng.begin_call(w["map"], None, emb_state=False)
    eng.bind(w["map"], w["dec"], P.IterConfig(step_size=0.04), train_decoder=False, want_emb_grad=False, want_pose_grad=True, update_emb=False,
             update_decoder=False, update_pose=True, skip_mode=2)
    steps = 700
for _ in range(5):
    eng.run_bound()
torch.cuda.synchronize()


def burst(tag):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(steps):
        eng.run_bound()
        ev[k + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)])
    big = np.nonzero(d > 3 * np.median(d))[0]
    cum = np.cumsum(d)
    print(f"{tag}: wall {wall * 1e3:.1f} ms, device sum {d.sum():.1f} ms, median {np.median(d):.4f}, max {d.max():.3f} ms at step {int(d.argmax())} "
          f"(= {cum[int(d.argmax())] - d.max():.1f} ms after the load started); intervals > 3x median: {[(int(k), round(float(d[k]), 2)) for k in big[:12]]}; "
          f"first 10: {[round(float(x), 3) for x in d[:10]]}")


burst("warm (right after other work)")
for rep in range(3):
    time.sleep(idle)
    burst(f"after {idle:.0f} s idle, #{rep}")
