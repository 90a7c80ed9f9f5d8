#!/usr/bin/env python3
"""host-timed ms per iteration of the one-C-call engine loop at the launch-bound sizes: a rank's share of an 8-GPU run (16 384 interleaved rays,
trainable decoder), 4096 x 4 frames (frozen decoder), 2048 x 1 - 300 iterations each, median of 5 blocks.  `NL_LIB_PATH=ab_libs/x.so python
scripts/rank_share_probe.py` for same-box A/B of library builds (scripts/gpu_ab_probe.sh)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from nerf_loam_amd import dist as D, pipeline as P              # noqa: E402

dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
N = len(w["points"])


def loop(sel, fid, n_frames, train):
    eng = P.SdfEngine(max_rays=len(sel), samples_per_ray_cap=96 if len(sel) <= 16384 and n_frames > 1 else 48, max_frames=max(2, n_frames), device=dev)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel], fid)
    pose = w["pose"]
    eng.set_poses(np.stack([pose + np.array([0.25 * f, -0.1 * f, 0, 0, 0, 0], np.float32) for f in range(n_frames)]), [1] * n_frames)
    eng.begin_call(w["map"], w["dec"])
    eng.bind(w["map"], w["dec"], P.IterConfig(), train_decoder=train, update_decoder=train)
    for _ in range(20):
        eng.run_bound()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60):
            eng.run_bound()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 60 * 1e3)
    assert not eng.call_status()[2]
    return float(np.median(ts))


lo, hi = D.shard_bounds(N, 0, 8)
share = D.interleaved_order(N, 8)[lo:hi]
rs = np.random.default_rng(5)
s4 = np.concatenate([np.sort(rs.choice(N, 4096, replace=False)) for _ in range(4)])
s1 = np.sort(rs.choice(N, 2048, replace=False))
print("%-28s rank share %.4f  4096x4 frozen %.4f  2048x1 %.4f ms / iteration" % (
    os.path.basename(os.environ.get("NL_LIB_PATH", "product")), loop(share, None, 1, True),
    loop(s4, np.repeat(np.arange(4, dtype=np.int32), 4096), 4, False), loop(s1, None, 1, True)))
