"""k_decoder32 (two workgroups per CU): per-phase shader-clock stamps of every workgroup for its first 8 tiles, paired by
the CU the workgroups ran on -- shows whether one workgroup's serial phases overlap the other's GEMM loops. GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P
L.require_gpu()
lib = L.lib(); lib.nl_decoder_set_variant(1)
w = bench.build_workload(torch.device("cuda"))
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
nwg = eng.dec_grid()
dbg = torch.zeros(nwg * 128, dtype=torch.int64, device="cuda")
names = ["A", "B:H1", "C:loop", "C:epi", "D:loss", "E:dH2", "F:loop", "F:epi", "H:dH1", "I:L1bwd", "J:dX"]
for stagger in [int(x) for x in os.environ.get("STAGGERS", "0,29000").split(",")]:
    lib.nl_decoder_set_stagger(stagger)
    for _ in range(2):
        eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
    dbg.zero_()
    lib.nl_decoder_set_debug_buffer(L.ptr(dbg))
    eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
    torch.cuda.synchronize()
    lib.nl_decoder_set_debug_buffer(None)
    d = dbg.cpu().numpy().reshape(nwg, 8, 16)
    key = d[:, 7, 15] if False else d[:, 0, 15]
    st = d[:, :, :12]
    ph = np.diff(st[:, 2:7], axis=2)
    print(f"stagger {stagger}: cycles/phase (mean over all workgroups, tiles 2..6)")
    for n, v in zip(names, ph.mean((0, 1))):
        print(f"  {n:8s} {v:9.0f}")
    print("  total/tile", (st[:, 3:8, 0] - st[:, 2:7, 0]).mean())
    uk, cnt = np.unique(key, return_counts=True)
    print("  distinct CUs", len(uk), "workgroups/CU histogram", np.bincount(cnt))
    # overlap on shared CUs: fraction of workgroup A's GEMM-loop time (C:loop, F:loop) during which B is also in a GEMM loop
    fr = []
    for k in uk[cnt == 2][:64]:
        a_, b_ = np.nonzero(key == k)[0]
        t0 = max(st[a_, 2, 0], st[b_, 2, 0]); t1 = min(st[a_, 6, 11], st[b_, 6, 11])
        if t1 <= t0: continue
        def gemm_iv(x):
            return [(st[x, t, 2], st[x, t, 3]) for t in range(8)] + [(st[x, t, 6], st[x, t, 7]) for t in range(8)]
        ia, ib = gemm_iv(a_), gemm_iv(b_)
        tot = sum(max(0, min(e, t1) - max(s_, t0)) for s_, e in ia)
        both = sum(max(0, min(e, e2, t1) - max(s_, s2, t0)) for s_, e in ia for s2, e2 in ib)
        fr.append((tot / (t1 - t0), both / max(tot, 1), (st[b_, 0, 0] - st[a_, 0, 0])))
    fr = np.array(fr)
    if len(fr):
        print("  per shared CU: A-in-GEMM fraction %.3f, of which B also in GEMM %.3f, start offset B-A mean %.0f (abs mean %.0f)" %
              (fr[:, 0].mean(), fr[:, 1].mean(), fr[:, 2].mean(), np.abs(fr[:, 2]).mean()))
