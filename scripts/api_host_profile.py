#!/usr/bin/env python
"""Where the host time of one API call goes (bundle_adjust_frames / track_frame at the reference's live shapes, 20 iterations per
call): cProfile over a few calls, and per call the wall time, the host time of the iteration loop (no synchronisation) and the
time the final read-back waits for the GPU.  `python scripts/api_host_profile.py` on a GPU box."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from argparse import Namespace
    from nerf_loam_amd import render_helpers as RH
    from nerf_loam_amd.criterion import Criterion
    from nerf_loam_amd.decoder import Decoder
    from nerf_loam_amd.lidar_frame import LidarFrame
    device = torch.device("cuda:0")
    w = bench.build_workload(device)
    crit = Criterion(Namespace(criteria=dict(sdf_weight=10000.0, fs_weight=1, eiko_weight=0.1, sdf_truncation=0.30),
                               data_specs=dict(max_depth=50.0)))
    dec_mod = Decoder().to(device)
    dec_mod.load_flat(w["dec"].params.clone())
    m = w["map"]
    emb = m.emb.view(torch.bfloat16)
    map_states = {"voxel_vertex_idx": torch.from_numpy(w["host"]["vertex_idx"]).to(device), "voxel_center_xyz": m.centres,
                  "voxel_structure": m.structure, "voxel_vertex_emb": emb,
                  "voxel_id2embedding_id": torch.from_numpy(w["host"]["id2row"]).to(device)}
    pts, cos = torch.from_numpy(w["points"]), torch.from_numpy(w["cos"])
    frames = []
    for i in range(4):
        P4 = np.eye(4); P4[:3, 3] = [0.25 * i, -0.1 * i, 0.0]
        frames.append(LidarFrame(i + 1, pts, cos, P4))
    kw = dict(truncation=0.3, max_voxel_hit=20, max_distance=50.0)
    iters = 20
    calls = {
        "bundle_adjust 2048x1": lambda: RH.bundle_adjust_frames(frames[:1], emb, map_states, dec_mod, crit, 0.2, 0.1, 2048, iters,
                                                                learning_rate=[0.03, 0.005, 0.001], update_pose=True, update_decoder=True, **kw),
        "bundle_adjust 4096x4 frozen": lambda: RH.bundle_adjust_frames(frames, emb, map_states, dec_mod, crit, 0.2, 0.1, 4096, iters,
                                                                       learning_rate=[0.03, 0.005, 0.001], update_pose=False, update_decoder=False, **kw),
        "track_frame 2048": lambda: RH.track_frame(frames[1].pose, frames[1], map_states, dec_mod, crit, 0.2, 2048, 0.04, iters,
                                                   learning_rate=0.005, **kw),
    }
    # split of a call: patch the engine's loop entry points and the read-back with timers
    from nerf_loam_amd import pipeline as P
    acc = {}

    def wrap(cls, name):
        f = getattr(cls, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        setattr(cls, name, g)
        return f
    names = ("run_bound", "reselect", "call_status_and_poses", "begin_call", "bind", "set_poses", "prepare_selection")
    for name, fn in calls.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"== {name}: call {np.median(ts) * 1e3:.3f} ms = {np.median(ts) / iters * 1e3:.4f} ms per iteration")
        orig = {n: wrap(P.SdfEngine, n) for n in names if hasattr(P.SdfEngine, n)}
        acc.clear()
        reps = 10
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        for n, f in orig.items():
            setattr(P.SdfEngine, n, f)
        print("   host time per call, ms: " + ", ".join(f"{n} {acc.get(n, 0.0) / reps * 1e3:.3f}" for n in orig))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            fn()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
        pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
        print("\n".join(l[:170] for l in s.getvalue().splitlines() if l.strip())[:6000])


if __name__ == "__main__":
    main()
