#!/usr/bin/env python3
"""is it the WAIT that is late?  Blocks of 100 one-C-call tracker steps (20 ms of device work each), the block's end awaited (a) by
torch.cuda.synchronize(), (b) by spinning on event.query(): wall time per block against the device time between the block's first and last event.
`python scripts/sync_probe.py [blocks] [cpu_work]`: cpu_work = 1 runs a multi-threaded numpy GEMM between blocks (what bench.py's oracle legs do)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from nerf_loam_amd import pipeline as P                          # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cpu_work = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
N = len(w["points"])
sel = np.sort(np.random.default_rng(3).choice(N, 2048, replace=False))
eng = P.SdfEngine(max_rays=2048, samples_per_ray_cap=96, device=dev)
eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
eng.begin_call(w["map"], None, emb_state=False)
eng.bind(w["map"], w["dec"], P.IterConfig(step_size=0.04), train_decoder=False, want_emb_grad=False, want_pose_grad=True, update_emb=False,
         update_decoder=False, update_pose=True, skip_mode=2)
for _ in range(20):
    eng.run_bound()
torch.cuda.synchronize()
A = np.random.default_rng(0).standard_normal((1500, 1500)).astype(np.float32)
for mode in ("synchronize", "spin on event.query()"):
    walls, devs = [], []
    for b in range(blocks):
        if cpu_work:
            (A @ A).sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(100):
            eng.run_bound()
        e1.record()
        if mode == "synchronize":
            torch.cuda.synchronize()
        else:
            while not e1.query():
                pass
        walls.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        devs.append(e0.elapsed_time(e1))
    walls, devs = np.array(walls), np.array(devs)
    late = walls - devs
    print(f"{mode:24s} cpu_work={cpu_work}: {blocks} blocks; device per block median {np.median(devs):.2f} ms max {devs.max():.2f}; wall median {np.median(walls):.2f} max {walls.max():.2f}; "
          f"wall - device: median {np.median(late):.3f} max {late.max():.2f} ms; blocks more than 5 ms late: {int((late > 5).sum())}")
