"""Same-box A/B of the fused decoder kernel's two workgroup layouts (NL_KERNEL_LAYOUT: 1 = one 8-wave workgroup per CU, 2 = two 4-wave
workgroups) on the bench workload: the kernel alone (HIP events around repeated launches on resident inputs), the whole one-C-call iteration,
and the phase stamps of workgroup 0.  GPU only.  Usage: decoder_layout_probe.py [rays ...]   (default: the full 131 072-ray scan)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, ops, pipeline as P
L.require_gpu()
dev = torch.device("cuda")
w = bench.build_workload(dev)
sizes = [int(a) for a in sys.argv[1:] if a != "stamps"] or ([] if "stamps" in sys.argv else [len(w["points"])])


def ev_time(fn, n=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for n_rays in sizes:
    sel = slice(None) if n_rays >= len(w["points"]) else np.sort(np.random.default_rng(1).choice(len(w["points"]), n_rays, replace=False))
    for train in (True, False):
        for layout in (1, 2, 1, 2):
            eng = P.SdfEngine(max_rays=n_rays if n_rays < len(w["points"]) else len(w["points"]), samples_per_ray_cap=48, dec_layout=layout)
            eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
            cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
            eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train)
            torch.cuda.synchronize()
            Pn = eng.stats()["P"]
            dec = w["dec"]
            c = eng.counters

            def kern():
                ops.decoder_fwd_bwd(eng.loss_scalars, eng.X, dec.params, dec.W2T, eng.s_ray, eng.s_depth, eng.cos_gt if hasattr(eng, "cos_gt") else eng.cos, eng.gt_dist,
                                    eng.sdf, eng.dsdf, eng.dX, eng.partials, eng.relu2_mask, eng.n_slabs, int(train), c, eng.kernel_modes)
            try:
                t_k = ev_time(kern)
            except Exception as e:                       # attribute names differ: fall back to the whole stage
                t_k = float("nan"); print("kernel-alone timing unavailable:", e)
            eng.bind(w["map"], w["dec"], cfg, train_decoder=train, want_emb_grad=train, update_decoder=train, update_emb=train)
            t_it = ev_time(eng.run_bound, n=40, warm=10)
            print(f"rays {n_rays:7d} samples {Pn:8d} train {int(train)} layout {layout}: kernel {t_k:.4f} ms   iteration {t_it:.4f} ms", flush=True)
            del eng

# phase stamps of workgroup 0 under layout 2 (layout 1: scripts/phase_probe.py); NL_N_SLABS=<CUs> runs ONE 4-wave workgroup per CU (what a workgroup does with the CU to itself)
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48, dec_layout=2)
print("slabs / workgroups of the stamped launch:", eng.n_slabs)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
dbg = torch.zeros(256 + 8 * 1024, dtype=torch.int64, device="cuda")     # 256 phase stamps + one record per workgroup (k_decoder2<.., STAMPS>)
for train in (True, False):
    for _ in range(2):
        eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train)
    L.lib().nl_decoder_set_debug_buffer(L.ptr(dbg))
    eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train)
    torch.cuda.synchronize()
    L.lib().nl_decoder_set_debug_buffer(None)
    dd = dbg.cpu().numpy()[:256].reshape(16, 16)
    e = dd[2:10]
    print("  inside D/E (stamps 5, 11-14, 6): U planes %.0f, dgrad prefetch %.0f, sums+bits j=0 %.0f, mask stores j=0 + sums j=1 %.0f, mask stores j=1 + barrier %.0f" % tuple(
        np.diff(e[:, [5, 11, 12, 13, 14, 6]], axis=1).mean(0)))
    d = dd[:, :11]
    names = ["-", "B:H1", "C:loop", "C:epi", "D:loss", "E:dH2", "F:loop", "-", "H/I", "dX+stage"]
    ph = np.diff(d[2:10], axis=1)
    print("layout 2", "train" if train else "frozen", "cycles/phase of ONE of the CU's two workgroups (mean over tiles 2..9):")
    for n, v in zip(names, ph.mean(0)):
        print(f"  {n:10s} {v:10.0f}")
    print("  total/tile (this workgroup; the CU finishes two tiles in that time)", (d[3:10, 0] - d[2:9, 0]).mean())
    rec = dbg.cpu().numpy()[256:].reshape(-1, 8)[:eng.n_slabs]
    wall = (rec[:, 1] - rec[:, 0]) / 100.0                      # us (100 MHz)
    cyc = rec[:, 3] - rec[:, 2]
    t0 = rec[:, 0].min()
    cu = ((rec[:, 5] & 0xF) << 16) | (((rec[:, 4] >> 13) & 7) << 12) | (((rec[:, 4] >> 12) & 1) << 8) | ((rec[:, 4] >> 8) & 0xF)     # (xcc, se, sh, cu)
    uniq, cnt = np.unique(cu, return_counts=True)
    print(f"  workgroups {len(rec)}: span {((rec[:, 1].max() - t0) / 100.0):.1f} us; per workgroup {wall.mean():.1f} us (min {wall.min():.1f}, max {wall.max():.1f}); "
          f"start offsets max {((rec[:, 0].max() - t0) / 100.0):.1f} us; clock {np.median(cyc / wall):.0f} MHz; tiles/workgroup {rec[:, 6].min()}..{rec[:, 6].max()}")
    print(f"  distinct CUs {len(uniq)}; workgroups per CU: " + ", ".join(f"{k}: {int((cnt == k).sum())} CUs" for k in sorted(set(cnt))))
    xcc = rec[:, 5] & 0xF
    print("  per XCC (workgroups, mean / min / max us): " + "  ".join(f"{x}: {int((xcc == x).sum())} {wall[xcc == x].mean():.0f}/{wall[xcc == x].min():.0f}/{wall[xcc == x].max():.0f}" for x in sorted(set(xcc))))
    se = (rec[:, 4] >> 13) & 7
    print("  per SE  (workgroups, mean us): " + "  ".join(f"{x}: {int((se == x).sum())} {wall[se == x].mean():.0f}" for x in sorted(set(se))))
    order = np.argsort(wall)
    print("  fastest 8 (block, xcc, se, cu, us):", [(int(i), int(xcc[i]), int(se[i]), int((rec[i, 4] >> 8) & 0xF), round(float(wall[i]), 0)) for i in order[:8]])
    print("  slowest 8 (block, xcc, se, cu, us):", [(int(i), int(xcc[i]), int(se[i]), int((rec[i, 4] >> 8) & 0xF), round(float(wall[i]), 0)) for i in order[-8:]])
    pair = {}
    for i in range(len(rec)): pair.setdefault(int(cu[i]), []).append(float(wall[i]))
    d2 = np.array([abs(v[0] - v[1]) for v in pair.values() if len(v) == 2])
    print(f"  |difference| between the two workgroups of a CU: mean {d2.mean():.1f} us, max {d2.max():.1f} us;  blocks of a CU: {[ [int(i) for i in np.nonzero(cu == uniq[0])[0]] ]}")
    print("  per-block us by block index (every 32nd):", [round(float(wall[i]), 0) for i in range(0, len(rec), 32)])
    late = rec[:, 0] - t0 > 0.2 * (rec[:, 1].max() - t0)
    print(f"  workgroups that started after 20 % of the span: {int(late.sum())}")
