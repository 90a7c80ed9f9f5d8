"""Per-stage HIP-event times of an iteration on the 150-scan map (bench.build_large_map) at the reference's live shapes. GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P
L.require_gpu()
dev = torch.device("cuda")
w = bench.build_workload(dev)
names = ["intersect", "sample", "gather", "decoder", "wgrad2", "reduce", "scatter", "optim"]
for tag, lm in (("single-scan map", None), ("150-scan map", bench.build_large_map(w, dev))):
    m = w["map"] if lm is None else lm["map"]
    pose = w["pose"] if lm is None else lm["poses"][75]
    for n_rays in (2048, 16384, 131072):
        rs = np.random.default_rng(5)
        sel = np.sort(rs.choice(len(w["points"]), n_rays, replace=False)) if n_rays < len(w["points"]) else np.arange(len(w["points"]))
        eng = P.SdfEngine(max_rays=len(sel), samples_per_ray_cap=96 if n_rays <= 16384 else 48)
        eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(pose[None], [1])
        cfg = P.IterConfig(); eng.begin_call(m, w["dec"])
        ev = [{n: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for n in names} for _ in range(6)]
        for k in range(6):
            eng.timers = ev[k]
            eng.forward_backward(m, w["dec"], cfg, train_decoder=True); eng.optimiser_step(m, w["dec"], cfg)
        torch.cuda.synchronize()
        st = eng.stats()
        ms = {n: float(np.mean([e[n][0].elapsed_time(e[n][1]) for e in ev[1:]])) for n in names}
        print(f"{tag:16s} rays {len(sel):6d} R {st['R']:6d} H {st['H']:2d} S {st['S']:3d} P {st['P']:8d} isect_ovf {int(st['ints'][L.NLC_ISECT_OVF])}  " +
              "  ".join(f"{n} {v * 1e3:7.1f}" for n, v in ms.items()) + f"   sum {sum(ms.values()) * 1e3:8.1f} us")
