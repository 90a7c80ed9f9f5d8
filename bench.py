#!/usr/bin/env python3
"""bench.py -- LiDAR rays/s per SDF iteration on the synthetic 64x2048 scan (BASELINE.json metric M1).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full mapping iteration of the reference's hot loop
(/root/reference/src/variations/render_helpers.py:356-423) over ALL 131 072 rays of one synthetic
scan: ray set-up, octree intersect, inverse-CDF sampling, embedding gather, decoder forward, SDF
loss, backward (decoder + embeddings + SE3 pose), Adam step - inputs resident in HBM.  With N GPUs
the scan's rays are sharded (strong scaling, total work fixed) with the three RCCL exchanges of
nerf_loam_amd/dist.py.  Rank 0 prints ONE JSON line.

Extra objects: "roofline" for the dominant kernel (fused decoder fwd+bwd on the matrix cores), timed
live with HIP events on the launch stream; "cpu_baseline": the oracle port timed on a bounded ray
sample on this box's host cores (a reported baseline, not the target).

Roofline accounting: `achieved` = ALGORITHMIC fp32 flops per launch / launch time.
The decoder's two 256-deep GEMMs and dW2 run on the bf16 matrix cores as exact-product splits (3 or 9
bf16 MFMAs per fp32 product, fp32 accumulation), the K = 16 layers on the fp32 matrix cores, so
`peak` is the matrix-pipe bound for THAT instruction mix: algorithmic flops / (executed fp32-MFMA
flops / 157.3 TF + executed bf16-MFMA flops / 2500 TF).  `frac` = achieved / peak = the fraction of the
launch during which the matrix pipes would be busy if nothing else limited the kernel.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
G = 2 * 256 * 256                      # one 256x256 GEMM per sample: 131 072 flops
L1 = 2 * 16 * 256                      # one 16x256 layer per sample: 8 192 flops
L3 = 2 * 256                           # the 256 -> 1 layer (VALU, not on the matrix pipes)
FLOPS_PER_SAMPLE_DECODER = 2 * (L1 + G + L3) + (L1 + L3)       # fwd + dgrad + (dW1, dW3): 288 256 (trainable decoder)
FLOPS_PER_SAMPLE_DECODER_FROZEN = 2 * (L1 + G + L3)            # 279 552
FLOPS_PER_SAMPLE_WGRAD2 = G                                    # dW2: 131 072


def matrix_pipe_model(kernel, gemm_mode, wgrad2_mode, train):
    """(algorithmic flops, executed fp32-MFMA flops, executed bf16-MFMA flops) per sample of `kernel`."""
    if kernel == "decoder":
        alg = FLOPS_PER_SAMPLE_DECODER if train else FLOPS_PER_SAMPLE_DECODER_FROZEN
        small = L1 * (3 if train else 2)                        # layer-1 forward, dX, (dW1)
        if gemm_mode in (1, 2):                                 # forward: 3x3 split (mode 2: six of the nine products),
            return alg, small, (9 if gemm_mode == 1 else 6) * G + 3 * G     # dgrad: {0,1} mask x 3-term split
        return alg, small + 2 * G, 0
    if wgrad2_mode == 1:
        return FLOPS_PER_SAMPLE_WGRAD2, L1, 3 * G               # H1 rebuilt on fp32 MFMA; mask x 3-term split
    return FLOPS_PER_SAMPLE_WGRAD2, L1 + G, 0


def roofline_entry(name, kernel, ms, P_local, gemm_mode, wgrad2_mode, train):
    alg, f32, b16 = matrix_pipe_model(kernel, gemm_mode, wgrad2_mode, train)
    t_bound = f32 / (PEAK_FP32_MFMA_TFLOPS * 1e12) + b16 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    ach = P_local * alg / (ms * 1e-3) / 1e12
    peak = alg / t_bound / 1e12
    return {"kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "avg_launch_ms": ms,
            "flops_per_launch": P_local * alg, "mfma_flops_executed_per_launch": {"f32": P_local * f32, "bf16": P_local * b16},
            "matrix_pipe_bound_ms": P_local * t_bound * 1e3}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS command
    (profiles/r*_pmc_summary.json: (2 x FETCH_SIZE + WRITE_SIZE) KB, the x2 being the gfx950 FETCH_SIZE
    correction of MI355X_MICROARCH.md).  None when no PMC summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))[kernel]["hbm_bytes_per_launch"])
    except Exception:
        return None


def build_workload(device, seed=777):
    from nerf_loam_amd import pipeline as P, synthetic as S
    from nerf_loam_amd.svo import Octree
    pts, cos = S.synthetic_scan(64, 2048, seed)
    pose = S.scan_pose()
    oc = Octree()
    oc.init(256 * 256 * 4, 16, 0.2)
    oc.insert(S.voxel_coords(pts, np.eye(3, dtype=np.float32), pose[:3], 0.2))
    centres, structure, vertex_idx = oc.export_device_layout()
    # embedding rows: one per vertex, like nerf_loam_amd.mapping.Mapping.get_embeddings (the reference allocates one per
    # OCCURRENCE, mapping.py:293-317: ~3x the rows, the duplicates are never read - SURVEY B7)
    flat = np.unique(vertex_idx[vertex_idx >= 0])
    id2row = -np.ones(len(centres), np.int32)
    id2row[flat] = np.arange(len(flat), dtype=np.int32)
    E = len(flat)
    rng = np.random.default_rng(seed)
    emb = rng.normal(0, 0.01, (E, 16)).astype(np.float32)
    emb_bits = (emb.view(np.uint32) >> 16).astype(np.uint16)             # truncation is fine for a random init
    k1, k2 = 1 / np.sqrt(16), 1 / np.sqrt(256)
    W1 = rng.uniform(-k1, k1, (256, 16)); b1 = rng.uniform(-k1, k1, 256)
    W2 = rng.uniform(-k2, k2, (256, 256)); b2 = rng.uniform(-k2, k2, 256)
    W3 = rng.uniform(-k2, k2, (1, 256)); b3 = rng.uniform(-k2, k2, 1)
    m = P.MapDevice(centres, structure, vertex_idx, id2row, emb_bits, 0.2, device=device)
    dec = P.DecoderDevice(W1, b1, W2, b2, W3, b3, device=device)
    return dict(points=pts, cos=cos, dirs=S.unit_dirs(pts), pose=pose, map=m, dec=dec, n_nodes=len(centres), n_rows=E,
                host=dict(centres=centres, structure=structure, vertex_idx=vertex_idx, id2row=id2row, emb_bits=emb_bits,
                          dec=(W1, b1, W2, b2, W3, b3)))


def cpu_baseline(w, n_rays=8192, seed=1):
    """Oracle port (oracle/oracle.py + nl_oracle.c) of the same iteration on a bounded ray sample."""
    from oracle import oracle as O
    h = w["host"]
    ms = O.MapState(h["centres"], h["structure"], h["vertex_idx"], h["id2row"], h["emb_bits"].copy(), 0.2)
    dec = O.DecoderParams(*[np.asarray(a, np.float32) for a in h["dec"]])
    rng = np.random.default_rng(seed)
    sel = np.sort(rng.choice(len(w["points"]), n_rays, replace=False))
    fr = O.Frame(w["dirs"][sel], w["points"][sel], w["cos"][sel], w["pose"].copy())
    st = O.AdamState()
    cfg = O.IterCfg()
    out = O.render_and_grad(ms, dec, [fr], cfg)                      # warm-up (library load, BLAS threads)
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        out = O.render_and_grad(ms, dec, [fr], cfg)
        O.optimiser_step(ms, dec, [fr], out, st, [0.03, 0.005, 0.001])
    dt = (time.perf_counter() - t0) / reps
    return dict(value=n_rays / dt, unit="rays/s", cores=int(torch.get_num_threads()), kind="port",
                sample=f"{n_rays} rays of the same 64x2048 scan, 1 mapping iteration incl. Adam, mean of {reps} "
                       f"({dt * 1e3:.0f} ms/iter; numpy/C oracle, GEMMs on torch-CPU threads)")


def pose_refine_bench(w, device, steps=200):
    """M2: ms per pose-refine step (track_frame iteration, render_helpers.py:452-512): 2048 rays, step 0.2*voxel,
    decoder + embeddings frozen, 6-dof pose Adam; rays resident, the launch sequence replayed as a hipGraph."""
    from nerf_loam_amd import pipeline as P
    rng = np.random.default_rng(3)
    sel = np.sort(rng.choice(len(w["points"]), 2048, replace=False))
    eng = P.SdfEngine(max_rays=2048, samples_per_ray_cap=96, device=device)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel])
    pose = w["pose"].copy(); pose[:3] += np.array([0.03, -0.02, 0.01], np.float32)
    eng.set_poses(pose[None], [1])
    cfg = P.IterConfig(step_size=0.04)
    eng.begin_call(w["map"], None)
    flags = dict(train_decoder=False, want_emb_grad=False, want_pose_grad=True, update_emb=False, update_decoder=False,
                 update_pose=True, lr_pose=0.005 / 3)
    out = {}
    for mode in ("eager", "graph"):
        if mode == "graph":
            try:
                eng.capture_iteration(w["map"], w["dec"], cfg, **flags)
            except Exception as e:                               # noqa: BLE001 - report, do not fail the bench
                out["graph_error"] = repr(e)[:200]
                break
        def one():
            if mode == "graph":
                eng.replay()
            else:
                eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=False, want_emb_grad=False, want_pose_grad=True)
                eng.optimiser_step(w["map"], w["dec"], cfg, update_emb=False, update_decoder=False, update_pose=True, lr_pose=0.005 / 3)
        for _ in range(10):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        out[f"ms_per_step_{mode}"] = (time.perf_counter() - t0) / steps * 1e3
    # the reference's track_frame re-draws its 2048 rays every iteration (LidarFrame.sample_rays on the CPU + H2D copy):
    # same step with the rays re-drawn on the device from the resident scan (nl_select_rays)
    scan = dict(dirs=torch.from_numpy(np.ascontiguousarray(w["dirs"])).to(device), points=torch.from_numpy(np.ascontiguousarray(w["points"])).to(device),
                cos=torch.from_numpy(np.ascontiguousarray(w["cos"])).to(device))
    def one_sel(k):
        eng.select_rays([scan], 2048, k)
        eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=False, want_emb_grad=False, want_pose_grad=True)
        eng.optimiser_step(w["map"], w["dec"], cfg, update_emb=False, update_decoder=False, update_pose=True, lr_pose=0.005 / 3)
    for k in range(10):
        one_sel(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one_sel(100 + k)
    torch.cuda.synchronize()
    out["ms_per_step_eager_with_device_ray_selection"] = (time.perf_counter() - t0) / steps * 1e3
    st = eng.stats()
    out.update(rays=2048, valid_samples=st["P"], step_size_m=0.04)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frozen-decoder", action="store_true", help="mapping with update_decoder=False (after freeze_frame)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    from nerf_loam_amd import _lib, pipeline as P, dist as D
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as tdist
        tdist.init_process_group("nccl", device_id=device)

    w = build_workload(device)
    N = len(w["points"])
    lo, hi = D.shard_bounds(N, rank, world)
    eng = P.SdfEngine(max_rays=hi - lo, samples_per_ray_cap=48, device=device)
    if world > 1:
        D.RayShardedExchange(eng)
    # balanced shards: the scan is beam-major and beams differ several-fold in voxels/samples per ray, so each rank takes every
    # world-th return instead of a block of whole beams (identity for one GPU; scripts/shard_probe.py, profiles/r01_k_shard_probe.txt)
    order = D.interleaved_order(N, world)
    sel = order[lo:hi]
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel])
    eng.set_poses(w["pose"][None], [1])
    cfg = P.IterConfig()
    train_dec = not args.frozen_decoder
    eng.begin_call(w["map"], w["dec"])

    def step():
        eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train_dec, ray_id_base=lo)
        eng.optimiser_step(w["map"], w["dec"], cfg, update_decoder=train_dec)

    def barrier():
        if world > 1:
            import torch.distributed as tdist
            tdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # per-kernel events for the roofline object (same stream as the launches)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        eng.timers = {"decoder": (ev[k][0], ev[k][1]), "wgrad2": (ev[k][1], ev[k][2])}
        step()
    barrier()
    dt = time.perf_counter() - t0
    eng.timers = None
    if world > 1:
        import torch.distributed as tdist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    st = eng.stats()
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev]))
    wg_ms = float(np.mean([b.elapsed_time(c) for _, b, c in ev])) if train_dec else 0.0
    P_local = st["P"]
    if rank == 0:
        gm, wm = _lib.lib().nl_decoder_get_gemm_mode(), _lib.lib().nl_decoder_get_wgrad2_mode()
        rf = roofline_entry("k_decoder<train>" if train_dec else "k_decoder<frozen>", "decoder", dec_ms, P_local, gm, wm, train_dec)
        rf = {"bound": "mfma", **rf,
              "traffic": pmc_traffic(("k_decoder<true, %s>" if train_dec else "k_decoder<false, %s>") % ("true" if gm >= 1 else "false")) if gm != 2 else None,
              "peak_note": ("matrix-pipe bound of the kernel's instruction mix: "
                            + (f"256-deep GEMMs as {'exact-product ' if gm == 1 else ''}bf16 splits ({9 if gm == 1 else 6} + 3 MFMAs per fp32 product, 2500 TF pipe), "
                               "K=16 layers on the fp32 pipe (157.3 TF)" if gm >= 1 else "all GEMMs on the fp32 pipe (157.3 TF)")),
              "second_kernel": (roofline_entry("k_decoder_wgrad2_x" if wm == 1 else "k_decoder_wgrad2", "wgrad2", wg_ms, P_local, gm, wm, True)
                                if train_dec else None)}
        out = {
            "metric": "LiDAR rays/sec per SDF iter (64x2048 scan)",
            "value": N * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": ("fp32 values and fp32 accumulation throughout; the decoder's 256-deep GEMMs are evaluated on the bf16 "
                           "matrix cores as exact-product splits (each fp32 operand = 3 bf16 terms exactly; ReLU masks are {0,1}), "
                           "NL_GEMM_MODE=0 / NL_WGRAD2_MODE=0 select the plain fp32-MFMA kernels" if (gm >= 1 or wm == 1)
                           else "fp32 MFMA kernels (NL_GEMM_MODE=0, NL_WGRAD2_MODE=0)"),
            "config": {"workload": "synthetic 64x2048 scan (131072 rays), 1 mapping iteration/step: intersect+sample+gather+"
                                   "decoder fwd/bwd+SDF loss+emb/decoder/pose grads+Adam; voxel 0.2 m, step 0.1 m, "
                                   + ("decoder trainable" if train_dec else "decoder frozen"),
                       "rays": N, "octree_nodes": w["n_nodes"], "embedding_rows": w["n_rows"], "hit_rays": st["R"],
                       "valid_samples_rank0": P_local, "max_samples_per_ray": st["S"], "parallelism": f"ray-shard x{world}" + (" (interleaved returns)" if world > 1 else "")},
            "roofline": rf,
        }
        if world == 1:
            out["pose_refine"] = pose_refine_bench(w, device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
