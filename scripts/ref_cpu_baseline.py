#!/usr/bin/env python3
"""Build container only (needs /root/reference): the REFERENCE's own Python path - bundle_adjust_frames ->
render_rays -> Criterion -> loss.backward() -> torch.optim.Adam.step(), unmodified sources - timed on this container's host
cores on the synthetic 64x2048 scan (SURVEY 8d "CPU baseline").  The harness is tests/golden/make_golden.py's: `.cuda()`
neutralised, the C restatement of the two CUDA kernels injected as its `grid` module, the reference's own C++ octree
(oracle/_ref/svo_ref.so).  The ray subset is a strided subset of the scan; the sampler draws torch uniform_ noise itself.
The number goes into BASELINE.md next to the oracle-port number bench.py measures on the GPU box (the GPU box has no
reference checkout, so bench.py's cpu_baseline is kind "port").

    python scripts/ref_cpu_baseline.py [n_rays ...]      (default 8192 32768)
    python scripts/ref_cpu_baseline.py --json N          (bench.py's cpu_baseline leg when a reference checkout exists: ONE line of JSON
                                                          for N rays on all torch threads; NL_REFERENCE_ROOT overrides /root/reference)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G                                         # noqa: E402  (patches torch / injects `grid` at import)

G._NOISE["ray_ids"] = None                                      # let the reference draw its own uniform_ noise
G.RH.ray_intersect = G._orig_ray_intersect
G.RH.render_rays = G._orig_render


def main():
    as_json = "--json" in sys.argv
    if as_json:
        sys.argv.remove("--json")
        import contextlib
        import io
        import json
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = measure([int(a) for a in sys.argv[1:]] or [8192], threads_list=(torch.get_num_threads(),))
        print(json.dumps(res[0]))
        return
    measure([int(a) for a in sys.argv[1:]] or [8192, 32768])


def measure(sizes, threads_list=None):
    results = []
    sc = G.build_scene(64, 2048, 777)
    M = len(sc["points"])
    print(f"scan {M} returns, {sc['centres'].shape[0]} octree nodes, {sc['emb'].shape[0]} embedding rows, "
          f"{torch.get_num_threads()} torch threads, {os.cpu_count()} host cores")
    for n in sizes:
        sel = np.arange(0, M, max(1, M // n))[:n]

        def sample_rays(self, N_rays, track=False):
            m = np.zeros((self.num_point, 1), bool)
            m[sel] = True
            self.sample_mask = torch.from_numpy(m)
        G.LidarFrame.sample_rays = sample_rays
        for threads in (threads_list or (torch.get_num_threads(), 1)):
            torch.set_num_threads(threads)
            fr = G.make_frame(1, sc["points"], sc["cos"], G.pose4())
            dec = G.make_decoder(777)
            emb = sc["emb"].clone().requires_grad_()
            ms = {"voxel_vertex_idx": sc["features"], "voxel_center_xyz": sc["centres"].detach().clone().requires_grad_(),
                  "voxel_structure": sc["structure"], "voxel_vertex_emb": emb, "voxel_id2embedding_id": sc["id_table"]}
            crit = G.Criterion(G.ARGS)
            run = lambda k: G.RH.bundle_adjust_frames([fr], emb, ms, dec, crit, 0.2, 0.1, len(sel), k, 0.30, 20, 50.0,    # noqa: E731
                                                      learning_rate=[0.03, 0.005, 0.001], update_pose=True, update_decoder=True)
            run(1)
            reps = 3 if threads > 1 else 2
            t0 = time.perf_counter()
            run(reps)
            dt = (time.perf_counter() - t0) / reps
            print(f"reference python path, {len(sel)} rays, {threads} thread(s): {dt * 1e3:.0f} ms/iter = {len(sel) / dt:.0f} rays/s")
            results.append(dict(rays=int(len(sel)), threads=int(threads), ms_per_iter=dt * 1e3, rays_per_s=len(sel) / dt, timed_iterations=reps,
                                octree_nodes=int(sc["centres"].shape[0]), embedding_rows=int(sc["emb"].shape[0])))
        torch.set_num_threads(os.cpu_count())
    return results


if __name__ == "__main__":
    main()
