#!/usr/bin/env python3
"""the tracker leg on the 150-scan map, an event after every step: which of the 310 steps of bench.py's track_2048 leg make one of its three
100-step blocks read ~0.8 ms per step?  (stall_probe.py on the one-scan map shows no device-side gap.)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402

dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
lm = bench.build_large_map(w, dev, 150, 3.0, 0.2)
out, eng = bench.tracker_step_on_map(w, lm, dev, 0.04, 0.005, steps=1, with_parity=False)
torch.cuda.synchronize()
steps = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = np.zeros(steps)
cnt = []
ev[0].record()
for k in range(steps):
    t0 = time.perf_counter()
    eng.run_bound()
    ev[k + 1].record()
    host[k] = time.perf_counter() - t0
torch.cuda.synchronize()
devi = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)])
print(f"device interval median {np.median(devi):.4f} ms, max {devi.max():.3f} at step {int(devi.argmax())}, sum {devi.sum():.1f} ms; host max {host.max() * 1e3:.2f} ms")
big = np.nonzero(devi > 3 * np.median(devi))[0]
print("device intervals > 3x median:", [(int(k), round(float(devi[k]), 3)) for k in big[:40]])
print("per 50 steps:", [round(float(devi[i:i + 50].mean()), 4) for i in range(0, steps, 50)])
st = eng.stats(); print("stats", {k: st[k] for k in ("P", "R", "S", "H", "overflow", "guard")}, "status", eng.call_status())
