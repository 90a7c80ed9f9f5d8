#!/usr/bin/env python
"""GPU time of the per-iteration ray selection (nl_select_rays_batch: two launches for all frames of a call) at the live shapes.
`python scripts/select_probe.py [path/to/lib.so]` on a GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loam_amd import _lib  # noqa: E402
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench  # noqa: E402
from nerf_loam_amd import pipeline as P  # noqa: E402
from nerf_loam_amd.lidar_frame import LidarFrame  # noqa: E402


def main():
    device = torch.device("cuda:0")
    w = bench.build_workload(device)
    pts, cos = torch.from_numpy(w["points"]), torch.from_numpy(w["cos"])
    frames = [LidarFrame(i + 1, pts, cos, np.eye(4)) for i in range(4)]
    out = []
    for n_rays, nf in ((2048, 1), (4096, 4)):
        eng = P.SdfEngine(max_rays=n_rays * nf, samples_per_ray_cap=8, max_frames=max(2, nf), device=device)
        eng.begin_call(w["map"], w["dec"])
        scans = [fr.device_scan(device) for fr in frames[:nf]]
        assert eng.prepare_selection(scans, n_rays)
        for k in range(20):
            eng.reselect(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 400
        e0.record()
        for k in range(reps):
            eng.reselect(100 + k)
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{n_rays} x {nf}: {e0.elapsed_time(e1) / reps * 1e3:.2f} us")
    print(os.path.basename(_lib.LIB_PATH), " | ".join(out))


if __name__ == "__main__":
    main()
