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

Roofline accounting (DESIGN.md section 4.0 - three numbers, one definition each):
  * `achieved` = ALGORITHMIC fp32 flops per launch (288 256 per valid sample with a trainable decoder: what the reference's three fp32 GEMMs + autograd
    need) / the kernel's launch time by HIP events on the launch stream;
  * `peak` = the matrix-pipe bound of the kernel's OWN instruction mix: every contraction runs on the 16-bit matrix cores (2500 TF dense) on split
    operands - by default fp16 PAIRS (gemm mode 4: 3 forward + 2 dgrad matrix instructions per fp32 product of the 256-deep GEMMs, 4 + 3 + 6 of the
    K = 16 ones = 761 856 EXECUTED flops per sample), under `exact_products` three-term bf16 splits (8 + 3) -, so
    peak = algorithmic flops / (executed 16-bit flops / 2500 TF + executed fp32-MFMA flops / 157.3 TF);
  * `frac` = achieved / peak = matrix-pipe bound time / launch time: the fraction of the launch during which the matrix pipes would be busy if nothing
    else limited the kernel (the SQ MFMA-busy counter of profiles/r*_pmc_summary.json agrees in kind).  `frac_of_dense_16bit_peak` prices the
    algorithmic flops against the plain 2500 TF figure.
"""
import argparse
import contextlib
import gc
import json
import os
import sys
import time


def host_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), else the visible core count"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return float(os.cpu_count() or 1)


# The host thread pools (numpy's OpenBLAS, torch's OpenMP) are capped BELOW the container's CPU quota before the libraries start.  On the GPU
# boxes of this pool 256 cores are visible and the quota is 16: OpenBLAS starts 64 threads, they spin after every call, the cgroup runs out of
# quota and the kernel throttles the WHOLE process for up to ~80 ms of a 100 ms period - the launching thread included.  That was the ~60 ms
# pause in one 100-step block of every tracker leg (an oracle parity check runs just before them) and the one 3.57 ms default run
# (scripts/sync_probe.py, profiles/r05_sync_probe.txt: 8-17 of 40 blocks late with the default pools, none with <= 16 threads).
HOST_CPU_QUOTA = host_cpu_quota()
HOST_POOL_THREADS = max(1, min(16, (int(HOST_CPU_QUOTA) - 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))     # (the ranks of a node share the quota)
_POOL_VARS_SET_HERE = [_v for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS") if _v not in os.environ]
for _v in _POOL_VARS_SET_HERE:
    os.environ[_v] = str(HOST_POOL_THREADS)

import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402

try:                                                           # (a caller imported numpy first: its pool already exists - limit it now)
    import threadpoolctl                                       # noqa: E402
    threadpoolctl.threadpool_limits(limits=int(os.environ["OPENBLAS_NUM_THREADS"]), user_api="blas")
except Exception:                                              # noqa: BLE001 - no threadpoolctl: the environment variables above did it for a fresh process
    pass
torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))


@contextlib.contextmanager
def no_gc():
    """host-timed regions run with the cyclic garbage collector off (what `timeit` does)"""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

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
    """(algorithmic flops, executed fp32-MFMA flops, executed 16-bit (bf16 or fp16: same rate) MFMA flops) per sample of `kernel`."""
    if kernel == "decoder":
        alg = FLOPS_PER_SAMPLE_DECODER if train else FLOPS_PER_SAMPLE_DECODER_FROZEN
        small = L1 * (3 if train else 2)                        # layer-1 forward, dX, (dW1)
        if gemm_mode in (1, 2, 3):                              # forward: 3x3 split (9, 8 or 6 of the nine products), dgrad: {0,1} mask x
            return alg, small - L1, {1: 9, 3: 8, 2: 6}[gemm_mode] * G + 3 * G + 9 * L1   # 3-term split; layer-1 forward: 3x3 split too
        if gemm_mode in (4, 5):                                 # fp16 pairs: forward 3 or 4 of the four products, dgrad mask x 2 terms, layer 1 all four,
            # dX three products (16x16x32), dW1 / db1 three products on 32x32x16 with 17 of 32 output columns used (2 x 3 x L1 executed): nothing on the fp32 pipe
            return alg, 0, {4: 3, 5: 4}[gemm_mode] * G + 2 * G + 4 * L1 + 3 * L1 + (6 * L1 if train else 0)
        return alg, small + 2 * G, 0
    if wgrad2_mode == 2:
        return FLOPS_PER_SAMPLE_WGRAD2, 0, 2 * G + 4 * L1       # H1 rebuilt as 2x2 fp16 products; mask x fp16 pair
    if wgrad2_mode == 1:
        return FLOPS_PER_SAMPLE_WGRAD2, 0, 3 * G + 9 * L1       # H1 rebuilt as 3x3 bf16 products; mask x 3-term split
    return FLOPS_PER_SAMPLE_WGRAD2, L1 + G, 0


def roofline_entry(name, kernel, ms, P_local, gemm_mode, wgrad2_mode, train):
    alg, f32, b16 = matrix_pipe_model(kernel, gemm_mode, wgrad2_mode, train)
    t_bound = f32 / (PEAK_FP32_MFMA_TFLOPS * 1e12) + b16 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    ach = P_local * alg / (ms * 1e-3) / 1e12
    peak = alg / t_bound / 1e12
    return {"kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "avg_launch_ms": ms,
            "flops_per_launch": P_local * alg, "mfma_flops_executed_per_launch": {"f32": P_local * f32, "16bit": P_local * b16},
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
        table = json.load(open(files[-1]))
        key = next(k for k in table if k == kernel or k.startswith(kernel.rstrip(">") + ","))      # (template arguments may follow)
        return float(table[key]["hbm_bytes_per_launch"])
    except Exception:
        return None


def pmc_traffic_in_run(kernel_prefix, timeout=240):
    """HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE - they do not fit one
    pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"; --kernel-trace + --pmc only) over a child `bench.py --pmc-child` (the same
    workload, three iterations), counters of the full-scan launches averaged, (2 x FETCH_SIZE + WRITE_SIZE) KB with the guide's gfx950
    FETCH_SIZE correction.  None when rocprofv3 is not on PATH, NL_BENCH_PMC=0, or a pass fails (the caller falls back to the committed
    summary and says so)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("NL_BENCH_PMC", "1") == "0":
        return None
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="nl_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                            os.path.abspath(__file__), "--pmc-child"], cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            path = next((os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")), None)
            got = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                   if r["Counter_Name"] == counter and r["Kernel_Name"].replace("void ", "").startswith(kernel_prefix)]
            top = max(got)
            keep = [v for v in got if 2 * v >= top]                  # (the full-scan launches)
            vals[counter] = sum(keep) / len(keep)
        except Exception as e:                                       # noqa: BLE001
            print(f"in-run PMC pass {counter} failed: {e!r}", file=sys.stderr)
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def committed_traffic_source():
    import glob
    import subprocess
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return "no PMC summary available"
    rel = os.path.relpath(files[-1], ROOT)
    try:
        h = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", rel], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:                                                # noqa: BLE001
        h = ""
    return f"committed rocprofv3 PMC passes of this command ({rel}" + (f" @ {h}" if h else "") + "), NOT measured in this run"


def build_workload(device, seed=777, voxel=0.2):
    from nerf_loam_amd import pipeline as P, synthetic as S
    from nerf_loam_amd.svo import Octree
    pts, cos = S.synthetic_scan(64, 2048, seed)
    pose = S.scan_pose()
    oc = Octree()
    oc.init(256 * 256 * 4, 16, voxel)
    oc.insert(S.voxel_coords(pts, np.eye(3, dtype=np.float32), pose[:3], voxel))
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
    m = P.MapDevice(centres, structure, vertex_idx, id2row, emb_bits, voxel, device=device)
    dec = P.DecoderDevice(W1, b1, W2, b2, W3, b3, device=device)
    # unit directions (LidarFrame.get_rays, lidarFrame.py:47-52) by the device kernel from the resident points; the oracle-side
    # restatement S.unit_dirs is what the CPU legs consume - parity_check holds the two against each other, bit for bit
    from nerf_loam_amd import ops
    pts_dev = torch.from_numpy(np.ascontiguousarray(pts)).to(device)
    dirs_dev = torch.empty_like(pts_dev)
    ops.unit_dirs(pts_dev, dirs_dev)
    return dict(points=pts, cos=cos, dirs=dirs_dev.cpu().numpy(), dirs_host=S.unit_dirs(pts), pose=pose, map=m, dec=dec, n_nodes=len(centres), n_rows=E, voxel=voxel,
                host=dict(centres=centres, structure=structure, vertex_idx=vertex_idx, id2row=id2row, emb_bits=emb_bits,
                          dec=(W1, b1, W2, b2, W3, b3)))


REF_OFFBOX = ("the reference's own unmodified Python path (scripts/ref_cpu_baseline.py) needs its checkout, which a GPU box does not have: timed in the "
              "build container on 8 cores it does 12.5-13.9 k rays/s (BASELINE.md section 5: 9.4 s per 131 072-ray iteration)")


def reference_cpu_baseline(n_rays=8192, timeout=600):
    """kind "reference": the reference's OWN bundle_adjust_frames -> render_rays -> Criterion -> backward -> Adam on this box's host cores
    (scripts/ref_cpu_baseline.py in a child process - its harness patches torch's .cuda()), when a checkout is present
    (NL_REFERENCE_ROOT or /root/reference) and oracle/_ref holds its octree build.  None otherwise."""
    import subprocess
    root = os.environ.get("NL_REFERENCE_ROOT", "/root/reference")
    if not (os.path.isdir(os.path.join(root, "src")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "svo_ref.so"))):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ref_cpu_baseline.py"), "--json", str(n_rays)], capture_output=True,
                           text=True, timeout=timeout, env={**os.environ, "NL_REFERENCE_ROOT": root, "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                       # noqa: BLE001 - report, fall back to the port
        print(f"reference CPU baseline failed ({e!r}); falling back to the oracle port", file=sys.stderr)
        return None
    return dict(value=d["rays_per_s"], unit="rays/s", cores=d["threads"], kind="reference",
                sample=f"{d['rays']} rays (every {max(1, 131072 // d['rays'])}th return of the same 64x2048 scan), the reference's unmodified bundle_adjust_frames "
                       f"(render_rays + Criterion + autograd backward + torch.optim.Adam, {root}/src) on {d['threads']} torch-CPU threads of {os.cpu_count()} host "
                       f"cores, 1 warm-up + {d['timed_iterations']} timed iterations, {d['ms_per_iter']:.0f} ms/iter; its two CUDA kernels replaced by their C "
                       f"restatement (oracle/nl_oracle.c); {d['octree_nodes']} octree nodes, {d['embedding_rows']} embedding rows as the reference allocates them")


def reference_recorded():
    """The reference's OWN CPU path, as last timed where its checkout exists (the build container: profiles/r05_reference_cpu_baseline.json, written by
    scripts/ref_cpu_baseline.py --json 131072 with its provenance).  A GPU box has no /root/reference, so the bench line there carries this RECORD
    next to the same-box port: one glance gives reference-path rays/s, its core count, and the port on the same thread count."""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "r05_reference_cpu_baseline.json")))
    except Exception:                                            # noqa: BLE001
        return None
    return dict(value=r["rays_per_s"], unit="rays/s", cores=r["threads"], kind="reference", measured_in_this_run=False, ms_per_iter=r["ms_per_iter"], rays=r["rays"],
                torch=r["torch"], where=r["where"], reference=r["reference"], command=r["command"],
                note="the reference's unmodified bundle_adjust_frames on the CPU, all 131 072 rays of the same synthetic scan, timed in the build container (the only place its "
                     "checkout exists); compare with by_threads['%d'] of the port measured in this run" % r["threads"])


def cpu_baseline(w, n_rays_probe=16384, n_rays_1t=8192, reps_1t=2, seed=1):
    """The SAME iteration on this box's host cores.  With a reference checkout on the box: the reference's own Python path (kind
    "reference").  Otherwise the oracle port (oracle/oracle.py + nl_oracle.c: numpy / C with closed-form gradients, the GEMMs and the
    decoder's element-wise stages on the torch-CPU threads) - kind "port": a short probe picks the better of {all cores, 16 threads}, then ONE
    timed mapping iteration incl. Adam over ALL 131 072 rays of the scan at that thread count (after one warm-up on the probe subset), and the
    single-thread figure on a strided subset.  The two C stages (intersect, sampler: independent per ray) are 2 % of the port's iteration
    (cProfile: 0.09 s of 4.8 s at 32 768 rays on 8 cores) - parallelising them would not move the number; what limits the port is numpy's
    single-threaded element-wise passes over [P,256] arrays."""
    ref = reference_cpu_baseline()
    if ref is not None:
        return ref
    from oracle import oracle as O
    h = w["host"]
    N = len(w["points"])

    def run(n, n_warm, n_rep):
        ms = O.MapState(h["centres"], h["structure"], h["vertex_idx"], h["id2row"], h["emb_bits"].copy(), 0.2)
        dec = O.DecoderParams(*[np.asarray(a, np.float32) for a in h["dec"]])
        sel = np.arange(0, N, max(1, N // n))[:n]
        fr = O.Frame(w["dirs_host"][sel], w["points"][sel], w["cos"][sel], w["pose"].copy())
        st = O.AdamState()
        cfg = O.IterCfg()
        ts = []
        for k in range(n_warm + n_rep):
            t0 = time.perf_counter()
            out = O.render_and_grad(ms, dec, [fr], cfg)
            O.optimiser_step(ms, dec, [fr], out, st, [0.03, 0.005, 0.001])
            if k >= n_warm:
                ts.append(time.perf_counter() - t0)
        return len(sel), float(np.median(ts))

    cores = int(torch.get_num_threads())
    quota = max(1, int(HOST_CPU_QUOTA))                                      # (threads beyond the container's CPU quota only get the process throttled)
    res = {}
    try:
        # (8 threads: the thread count of the recorded reference-path figure - the two are then comparable on equal cores)
        for thr in dict.fromkeys((quota, min(16, quota), min(8, quota))):       # probe: which thread count serves the port best on this box
            torch.set_num_threads(thr)
            res[thr] = run(n_rays_probe, 1, 1) + (1, 1)
        best = max(res, key=lambda t: res[t][0] / res[t][1])
        torch.set_num_threads(best)
        n_b, t_b = run(N, 0, 1)                                           # the whole scan, one timed iteration (the probe was its warm-up)
        torch.set_num_threads(1)
        n_1, t_1 = run(n_rays_1t, 1, reps_1t)
    finally:
        torch.set_num_threads(cores)
    rec = reference_recorded()
    sample = (f"all {n_b} rays of the same 64x2048 scan, 1 mapping iteration incl. Adam, 1 timed iteration after a warm-up on {n_rays_probe} rays: "
              f"{t_b * 1e3:.0f} ms/iter; numpy/C oracle port, GEMMs + decoder element-wise stages on {best} torch-CPU threads ({os.cpu_count()} host cores visible, container CPU quota {HOST_CPU_QUOTA:g}) "
              f"(the other numpy stages are single-threaded).  NOT the reference's own code: ")
    sample += (f"reference_recorded carries the reference path's figure ({rec['value']:.0f} rays/s on {rec['cores']} threads in the build container) with its provenance"
               if rec else REF_OFFBOX)
    return dict(value=n_b / t_b, unit="rays/s", cores=best, kind="port", host_cpu_quota=HOST_CPU_QUOTA, reference_recorded=rec, sample=sample,
                by_threads={str(t): dict(value=res[t][0] / res[t][1], rays=res[t][0], ms_per_iter=res[t][1] * 1e3, warmup=res[t][2], timed=res[t][3])
                            for t in res},
                single_thread=dict(value=n_1 / t_1, unit="rays/s", cores=1,
                                   sample=f"{n_1} rays, 1 warm-up + {reps_1t} timed, median {t_1 * 1e3:.0f} ms/iter, torch.set_num_threads(1)"))


def parity_check(eng, w, cfg, train_dec, every=8):
    """parity on the same inputs in the same run.  One more forward+backward of the timed engine (current, i.e.
    Adam-updated, embeddings / decoder / pose) against the oracle on ALL 131 072 rays for the geometry (hit lists, sample
    layout, depths: bit for bit, C restatement of the two CUDA kernels) and on every `every`-th ray's samples for the field /
    decoder outputs (sdf, dL/dsdf, dL/dX).  Bars: the tests' (tests/test_gpu_parity.py compare_iteration)."""
    from oracle import oracle as O
    h = w["host"]
    m, dec = w["map"], w["dec"]
    N = eng.N
    eng.forward_backward(m, dec, cfg, train_decoder=train_dec)
    torch.cuda.synchronize()
    st = eng.stats()
    P_ = st["P"]
    emb_bits = m.emb.cpu().numpy().view(np.uint16).copy()
    dn = dec.numpy()
    pose = eng.pose6[0].cpu().numpy().copy()
    got = dict(hit_count=eng.hit_count[:N].cpu().numpy(), hit_idx=eng.hit_idx[:N].cpu().numpy(), hit_t0=eng.hit_t0[:N].cpu().numpy(),
               hit_t1=eng.hit_t1[:N].cpu().numpy(), samp_off=eng.samp_off[:N].cpu().numpy(), samp_count=eng.samp_count[:N].cpu().numpy(),
               vox=eng.s_vox[:P_].cpu().numpy(), depth=eng.s_depth[:P_].cpu().numpy(), sdf=eng.sdf[:P_].cpu().numpy(),
               dsdf=eng.dsdf[:P_].cpu().numpy(), X=eng.X[:P_].cpu().numpy(), dX=eng.dX[:P_].cpu().numpy())
    if eng.g_emb is not None:
        eng.g_emb.zero_()
    eng.g_pose.zero_()
    t0 = time.perf_counter()
    ms = O.MapState(h["centres"], h["structure"], h["vertex_idx"], h["id2row"], emb_bits, w.get("voxel", 0.2))
    dp = O.DecoderParams(dn["W1"], dn["b1"], dn["W2"], dn["b2"], dn["W3"], dn["b3"])
    fr = O.Frame(w["dirs_host"], w["points"], w["cos"], pose)      # the oracle's own directions (host restatement of lidarFrame.py:47-52)
    sub = np.zeros(N, bool); sub[::every] = True
    out = O.render_and_grad(ms, dp, [fr], O.IterCfg(step_size=cfg.step_size, noise_seed=cfg.noise_seed), want_emb_grad=False,
                            want_dec_grad=False, eval_rays=sub)
    H_ = out["hit_idx"].shape[1]
    live = np.arange(H_)[None, :] < got["hit_count"][:, None]
    md = np.float32(cfg.max_distance)
    hits_equal = bool(np.array_equal(got["hit_count"] > 0, out["hits"]) and st["H"] == H_
                      and np.array_equal(np.where(live, got["hit_idx"][:, :H_], -1), out["hit_idx"])
                      and np.array_equal(np.where(live, got["hit_t0"][:, :H_], md), out["hit_t0"])
                      and np.array_equal(np.where(live, got["hit_t1"][:, :H_], md), out["hit_t1"]))
    hr = np.nonzero(out["hits"])[0]
    rr, ss = np.nonzero(out["valid"])
    samples_equal = bool(P_ == out["n_samples"] and st["S"] == out["valid"].shape[1]
                         and np.array_equal(got["samp_count"][hr], out["valid"].sum(1))
                         and np.array_equal(got["depth"], out["z_vals"][rr, ss]) and np.array_equal(got["vox"], out["s_idx"][rr, ss]))
    dirs_equal = bool(np.array_equal(w["dirs"].view(np.uint32), w["dirs_host"].view(np.uint32)))      # nl_unit_dirs == lidarFrame.py:47-52 on the host
    res = dict(rays=int(N), unit_dirs_equal=dirs_equal, hits_equal=hits_equal, samples_equal=samples_equal, valid_samples=int(P_), subset_rays=int(sub.sum()))
    if samples_equal:
        idx = got["samp_off"][out["sample_ray"]] + out["sample_slot"]            # engine sample index of the subset's samples
        ref_sdf = out["sdf"][np.searchsorted(hr, out["sample_ray"]), out["sample_slot"]]
        ref_ds = out["dsdf"][np.searchsorted(hr, out["sample_ray"]), out["sample_slot"]]
        dxe, dxr = got["dX"][idx].astype(np.float64), out["dfeat"].astype(np.float64)
        res.update(subset_samples=int(len(idx)),
                   sdf_max_abs_err=float(np.abs(got["sdf"][idx] - ref_sdf).max()), sdf_mean_abs_err=float(np.abs(got["sdf"][idx] - ref_sdf).mean()),
                   X_max_abs_err=float(np.abs(got["X"][idx] - out["feats"]).max()),
                   dsdf_max_err_rel_to_max=float(np.abs(got["dsdf"][idx] - ref_ds).max() / max(np.abs(ref_ds).max(), 1e-30)),
                   dX_rel_l2=float(np.linalg.norm(dxe - dxr) / max(np.linalg.norm(dxr), 1e-30)),
                   # samples whose dX row is off by more than 1e-3 of the largest entry: ReLU flips (a hidden unit whose pre-activation is ~1e-8 lands on
                   # either side of zero under another summation order), counted so that the norm bar is not mistaken for an element-wise one
                   dX_samples_off=float((np.abs(dxe - dxr).max(1) > 1e-3 * max(np.abs(dxr).max(), 1e-30)).mean()))
        res["ok"] = bool(dirs_equal and hits_equal and res["sdf_max_abs_err"] < 1e-4 and res["dsdf_max_err_rel_to_max"] < 1e-3 and res["dX_rel_l2"] < 1e-3)
    else:
        res["ok"] = False
    res["bars"] = {"unit_dirs/hits/samples": "bit-exact", "sdf_max_abs_err": 1e-4, "dsdf_max_err_rel_to_max": 1e-3, "dX_rel_l2": 1e-3}
    res["oracle_seconds"] = round(time.perf_counter() - t0, 2)
    return res


STEADY_STEPS = 200                     # informational sustained run after the timed region (steady_state)
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured float4 copy)


def stage_rooflines(eng, w, cfg, train_dec, steps=5):
    """HIP-event times of every stage of the iteration (a separate short pass after the headline loop, all stage timers on)
    and the bandwidth-bound kernels' achieved GB/s on their ALGORITHMIC bytes (unavoidable traffic: every input read once,
    every output written once; DESIGN.md section 7 lists the per-unit figures)."""
    names = ["intersect", "sample", "gather", "decoder", "wgrad2", "reduce", "scatter", "optim"]
    ev = [{n: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for n in names} for _ in range(steps)]
    for k in range(steps):
        eng.timers = ev[k]
        eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train_dec)
        eng.optimiser_step(w["map"], w["dec"], cfg, update_decoder=train_dec)
    torch.cuda.synchronize()
    eng.timers = None
    ms = {n: float(np.mean([e[n][0].elapsed_time(e[n][1]) for e in ev])) for n in names if train_dec or n not in ("wgrad2", "reduce")}
    st = eng.stats()
    N, P_, E = eng.N, st["P"], w["n_rows"]
    nhits = int(eng.hit_count[:N].sum().item())
    touched = int((eng.emb_m != 0).any(1).sum().item())                    # embedding rows the call has touched (they carry moments)
    n_vox = int(torch.unique(eng.s_vox[:P_]).numel())                      # distinct voxels the samples fall in
    by = {
        "intersect": N * (28 + 20) + nhits * 12,                            # dir, gt point, cos in; world dir, gt dist, count + hit list out
        "sample": 2 * (nhits * 12 + N * 16) + P_ * 16 + N * 8,              # both passes read the hit lists; (voxel, depth, dist, ray) out
        # every distinct datum once: the 8 row ids of a voxel and a touched embedding row are read once however many samples share
        # them (they do: ~8 samples per ray run through 3-4 voxels); what scales with the samples is the record, X and dX
        "gather": P_ * (12 + 64) + n_vox * 32 + touched * 32,               # sample record in, X out; row ids per voxel, bf16 rows
        "scatter": P_ * (12 + 64) + n_vox * 32 + touched * (32 + 64 * 2),   # + dX in; every touched accumulator row read-modify-written once
        "optim": touched * 16 * 20 + (E - touched) * 16 * 8 + (70401 * 28 + 1835008 if train_dec else 0),
        "reduce": (eng.n_slabs + 1) * 70401 * 4,
    }
    out = []
    for n, b in by.items():
        if n in ms:
            gbs = b / (ms[n] * 1e-3) / 1e9
            out.append({"stage": n, "algorithmic_bytes_per_launch": int(b), "avg_ms": ms[n], "achieved": gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
    return ms, by, out


def api_path_bench(w, device, iters=20):
    """The path users call: bundle_adjust_frames / track_frame of nerf_loam_amd.render_helpers at the reference's live
    shapes (configs/maicity/maicity.yaml:18-41, src/mapping.py:172-202): steady-state mapping 2048 rays x 1 frame, post-processing
    BA 4096 rays x 4 key-scans, tracking 2048 rays; `iters` iterations per call, ms per iteration of the whole call (ray
    selection, kernels, the call's set-up and its one read-back included), next to the bare engine loop on the same shapes."""
    from argparse import Namespace
    from nerf_loam_amd import pipeline as P, render_helpers as RH
    from nerf_loam_amd.criterion import Criterion
    from nerf_loam_amd.decoder import Decoder
    from nerf_loam_amd.lidar_frame import LidarFrame
    args = Namespace(criteria=dict(sdf_weight=10000.0, fs_weight=1, eiko_weight=0.1, sdf_truncation=0.30), data_specs=dict(max_depth=50.0))
    crit = Criterion(args)
    dec_mod = Decoder().to(device)
    dec_mod.load_flat(w["dec"].params.clone())
    m = w["map"]
    emb = m.emb.view(torch.bfloat16)
    table = torch.from_numpy(w["host"]["id2row"]).to(device)
    map_states = {"voxel_vertex_idx": torch.from_numpy(w["host"]["vertex_idx"]).to(device), "voxel_center_xyz": m.centres,
                  "voxel_structure": m.structure, "voxel_vertex_emb": emb, "voxel_id2embedding_id": table}
    pts, cos = torch.from_numpy(w["points"]), torch.from_numpy(w["cos"])
    frames = []
    for i in range(4):
        P4 = np.eye(4); P4[:3, 3] = [0.25 * i, -0.1 * i, 0.0]
        frames.append(LidarFrame(i + 1, pts, cos, P4))
    out = {"iterations_per_call": iters, "ray_selection": RH.RAY_SELECTION}

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) / iters * 1e3

    kw = dict(truncation=0.3, max_voxel_hit=20, max_distance=50.0)
    out["bundle_adjust_2048x1_ms_per_iter"] = timed(lambda: RH.bundle_adjust_frames(
        frames[:1], emb, map_states, dec_mod, crit, 0.2, 0.1, 2048, iters, learning_rate=[0.03, 0.005, 0.001], update_pose=True,
        update_decoder=True, **kw))
    out["bundle_adjust_4096x4_frozen_decoder_ms_per_iter"] = timed(lambda: RH.bundle_adjust_frames(
        frames, emb, map_states, dec_mod, crit, 0.2, 0.1, 4096, iters, learning_rate=[0.03, 0.005, 0.001], update_pose=False,
        update_decoder=False, **kw))
    out["track_frame_2048_ms_per_iter"] = timed(lambda: RH.track_frame(
        frames[1].pose, frames[1], map_states, dec_mod, crit, 0.2, 2048, 0.04, iters, learning_rate=0.005, **kw))

    # the bare engine on the same shapes (resident rays, no selection): what the API path is measured against
    def engine_loop(n_rays, n_frames, step, train, emb_grad, pose_grad):
        eng = P.SdfEngine(max_rays=n_rays * n_frames, samples_per_ray_cap=96, max_frames=max(2, n_frames), device=device)
        rng = np.random.default_rng(5)
        sel = np.concatenate([np.sort(rng.choice(len(w["points"]), n_rays, replace=False)) for _ in range(n_frames)])
        fid = np.repeat(np.arange(n_frames, dtype=np.int32), n_rays)
        eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel], fid)
        eng.set_poses(np.stack([fr.pose.data.detach().numpy() for fr in frames[:n_frames]]), [1] * n_frames)
        cfg = P.IterConfig(step_size=step)
        eng.begin_call(m, w["dec"])

        eng.bind(m, w["dec"], cfg, train_decoder=train, want_emb_grad=emb_grad, want_pose_grad=pose_grad, update_emb=emb_grad,
                 update_decoder=train, update_pose=pose_grad)

        def loop():
            for _ in range(iters):
                eng.run_bound()                                  # one C call per iteration (nl_iteration), like the API path
        return timed(loop)
    out["engine_2048x1_ms_per_iter"] = engine_loop(2048, 1, 0.1, True, True, True)
    out["engine_4096x4_frozen_decoder_ms_per_iter"] = engine_loop(4096, 4, 0.1, False, True, False)
    out["engine_track_2048_ms_per_iter"] = engine_loop(2048, 1, 0.04, False, False, True)
    return out


def shard_probe_bench(w, device, full_ms, steps=12):
    """What one rank of a K-GPU run does, measured on this GPU: the whole iteration (no exchanges) on rank 0's interleaved share of
    the scan (nerf_loam_amd.dist.interleaved_order, the order bench.py shards by).  T(1) / T(share) bounds the strong-scaling
    efficiency before the exchanges (scripts/shard_probe.py measures every rank)."""
    from nerf_loam_amd import pipeline as P, dist as D
    N = len(w["points"])
    out = {"full_scan_ms": full_ms}
    for K in (2, 4, 8):
        lo, hi = D.shard_bounds(N, 0, K)
        sel = D.interleaved_order(N, K)[lo:hi]
        eng = P.SdfEngine(max_rays=len(sel), samples_per_ray_cap=48, device=device)
        eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
        cfg = P.IterConfig()
        eng.begin_call(w["map"], w["dec"])
        eng.bind(w["map"], w["dec"], cfg, train_decoder=True)
        for _ in range(4):
            eng.run_bound()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_bound()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps * 1e3
        out[f"rank_share_1_of_{K}"] = {"rays": int(len(sel)), "ms_per_step": t, "t1_over_t": full_ms / t}
        del eng
    return out


def build_large_map(w, device, n_scans=150, spacing=3.0, voxel=0.2):
    """the synthetic scan inserted at `n_scans` poses `spacing` m apart along x (host octree, nerf_loam_amd.svo), N(0, 0.01^2) embeddings"""
    from nerf_loam_amd import pipeline as P, synthetic as S
    from nerf_loam_amd.svo import Octree
    t0 = time.perf_counter()
    oc = Octree()
    oc.init(256 * 256 * 4, 16, voxel)
    poses = []
    for i in range(n_scans):
        pose = S.scan_pose(tx=spacing * i)
        poses.append(pose)
        oc.insert(S.voxel_coords(w["points"], np.eye(3, dtype=np.float32), pose[:3], voxel))
    centres, structure, vertex_idx = oc.export_device_layout()
    flat = np.unique(vertex_idx[vertex_idx >= 0])
    id2row = -np.ones(len(centres), np.int32)
    id2row[flat] = np.arange(len(flat), dtype=np.int32)
    E = len(flat)
    emb = np.random.default_rng(778).normal(0, 0.01, (E, 16)).astype(np.float32)
    emb_bits = (emb.view(np.uint32) >> 16).astype(np.uint16)
    build_s = time.perf_counter() - t0
    m = P.MapDevice(centres, structure, vertex_idx, id2row, emb_bits, voxel, device=device)
    return dict(centres=centres, structure=structure, vertex_idx=vertex_idx, id2row=id2row, E=E, map=m, poses=poses, build_s=build_s, voxel=voxel)


# the tracker's operating points of the shipped configs: step = tracker_specs.step_size x voxel_size (src/tracking.py:36) -
# configs/maicity/maicity.yaml:22,29 (0.2 x 0.2), configs/kitti/kitti.yaml:22,29 (0.2 x 0.3), configs/ncd/ncd.yaml:22,29 (0.1 x 0.2)
TRACKER_SETTINGS = {"maicity": dict(voxel=0.2, step=0.04, lr=0.005), "kitti": dict(voxel=0.3, step=0.06, lr=0.005), "ncd": dict(voxel=0.2, step=0.02, lr=0.005)}


def tracker_step_on_map(w, lm, device, step, lr=0.005, n_rays=2048, steps=100, with_parity=True, samples_per_ray_cap=None):
    """M2 at the reference's real operating point: the track_frame-shaped iteration (render_helpers.py:452-512: decoder and embeddings frozen,
    pose gradient, 6-dof Adam with the tracker's lr / 3 rule) on `n_rays` returns of the scan in the middle of an ACCUMULATED map `lm`
    (build_large_map), from a pose 3-4 cm off.  The sample workspace is sized by pipeline.samples_per_ray_bound, like the API sizes it.
    -> ms per step (one C call per iteration), samples per hit ray, and the parity of one iteration against the oracle: hit lists, sample
    layout and depths bit for bit, sdf, dL/dsdf, dL/dX, the 6-dof pose gradient."""
    from nerf_loam_amd import pipeline as P
    voxel = lm["voxel"]
    m, poses = lm["map"], lm["poses"]
    dec = P.DecoderDevice(*[np.asarray(a, np.float32) for a in w["host"]["dec"]], device=device)
    sel = np.sort(np.random.default_rng(5).choice(len(w["points"]), n_rays, replace=False))
    cap = P.samples_per_ray_bound(voxel, step) if samples_per_ray_cap is None else samples_per_ray_cap
    eng = P.SdfEngine(max_rays=n_rays, samples_per_ray_cap=cap, max_frames=2, device=device)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel])
    pose = poses[len(poses) // 2].copy(); pose[:3] += np.array([0.03, -0.02, 0.01], np.float32)
    eng.set_poses(pose[None], [1])
    cfg = P.IterConfig(voxel_size=voxel, step_size=step)
    flags = dict(train_decoder=False, want_emb_grad=False, want_pose_grad=True, update_emb=False, update_decoder=False, update_pose=True, lr_pose=lr / 3)
    eng.begin_call(m, None, emb_state=False)
    out = dict(voxel_size_m=voxel, step_size_m=step, rays=n_rays, samples_per_ray_capacity=cap, octree_nodes=int(len(lm["centres"])), embedding_rows=int(lm["E"]))
    if with_parity:
        from oracle import oracle as O
        eng.forward_backward(m, dec, cfg, train_decoder=False, want_emb_grad=False, want_pose_grad=True)
        eng.optimiser_step(m, dec, cfg, update_emb=False, update_decoder=False, update_pose=False)       # pose_grad6 of this iteration, no step
        torch.cuda.synchronize()
        st = eng.stats()
        P_ = st["P"]
        out.update(valid_samples=int(P_), hit_rays=int(st["R"]), max_hits=int(st["H"]), max_samples_per_ray=int(st["S"]), overflow=int(st["overflow"]),
                   samples_per_hit_ray=float(P_) / max(st["R"], 1))
        ms_o = O.MapState(lm["centres"], lm["structure"], lm["vertex_idx"], lm["id2row"], m.emb.cpu().numpy().view(np.uint16).copy(), voxel)
        dn = dec.numpy()
        dp = O.DecoderParams(dn["W1"], dn["b1"], dn["W2"], dn["b2"], dn["W3"], dn["b3"])
        ref = O.render_and_grad(ms_o, dp, [O.Frame(w["dirs_host"][sel], w["points"][sel], w["cos"][sel], pose.copy())], O.IterCfg(step_size=step),
                                want_emb_grad=False, want_dec_grad=False)
        rr, ss = np.nonzero(ref["valid"])
        hc = eng.hit_count[:n_rays].cpu().numpy()
        H_ = ref["hit_idx"].shape[1]
        live = np.arange(H_)[None, :] < hc[:, None]
        geom = bool(not st["overflow"] and P_ == ref["n_samples"] and np.array_equal(hc > 0, ref["hits"])
                    and np.array_equal(np.where(live, eng.hit_idx[:n_rays, :H_].cpu().numpy(), -1), ref["hit_idx"])
                    and np.array_equal(eng.s_depth[:P_].cpu().numpy(), ref["z_vals"][rr, ss]) and np.array_equal(eng.s_vox[:P_].cpu().numpy(), ref["s_idx"][rr, ss]))
        par = dict(geometry_bit_exact=geom, rays_at_the_20_hit_cap=float((hc == 20).mean()))
        if geom:
            g6, r6 = eng.pose_grad6[0].cpu().numpy().astype(np.float64), ref["grad_pose"][0].astype(np.float64)
            dx = eng.dX[:P_].cpu().numpy()
            par.update(sdf_max_abs_err=float(np.abs(eng.sdf[:P_].cpu().numpy() - ref["sdf"][rr, ss]).max()),
                       dsdf_max_err_rel_to_max=float(np.abs(eng.dsdf[:P_].cpu().numpy() - ref["dsdf"][rr, ss]).max() / max(np.abs(ref["dsdf"]).max(), 1e-30)),
                       dX_rel_l2=float(np.linalg.norm((dx - ref["dfeat"]).astype(np.float64)) / max(np.linalg.norm(ref["dfeat"].astype(np.float64)), 1e-30)),
                       # (a hidden unit whose pre-activation is ~1e-8 lands on either side of zero under another summation order: the whole difference of a
                       #  flipped sample's dX row; counted, so that a bar on the norm is not mistaken for a bar on every element)
                       dX_samples_off=float((np.abs(dx - ref["dfeat"]).max(1) > 1e-3 * max(np.abs(ref["dfeat"]).max(), 1e-30)).mean()),
                       pose_grad_max_err_rel_to_max=float(np.abs(g6 - r6).max() / max(np.abs(r6).max(), 1e-30)))
            par["ok"] = bool(par["sdf_max_abs_err"] < 1e-4 and par["dsdf_max_err_rel_to_max"] < 1e-3 and par["dX_rel_l2"] < 2e-3 and par["pose_grad_max_err_rel_to_max"] < 1e-3)
        else:
            par["ok"] = False
        out["parity_vs_oracle"] = par
        eng.begin_call(m, None, emb_state=False)
        eng.set_poses(pose[None], [1])
    if steps:
        eng.bind(m, dec, cfg, skip_mode=2, **flags)
        for _ in range(10):
            eng.run_bound()
        blocks = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                eng.run_bound()
            torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / steps * 1e3)
        st = eng.stats()
        steps_done, skipped, overflow = eng.call_status()
        out.update(ms_per_step=float(min(blocks)), ms_per_step_blocks=blocks, steps_taken=int(steps_done), steps_skipped=int(skipped), call_overflow=bool(overflow),
                   valid_samples_last_step=int(st["P"]), pose_moved_m=float(np.abs(eng.pose6[0, :3].cpu().numpy() - pose[:3]).max()))
    return out, eng


def large_map_bench(w, device, n_scans=150, spacing=3.0, iters=20):
    """The regime BASELINE configs 2 - 5 live in: a map accumulated over a trajectory (here `n_scans` synthetic scans `spacing` m apart
    along the street canyon: >= 1e6 embedding rows, a deeper octree), not the single-scan map of the headline number.  On it: ms per
    iteration of the one-C-call engine loop at the reference's live shapes (2048 rays x 1 frame trainable decoder, 4096 x 4 frozen) and
    on the full scan, with the rays of the scan(s) in the middle of the trajectory; what begin_call and the optimiser step cost with the
    touched-rows bookkeeping and with the dense one (sparse_adam=False: E-sized memset per call, sweep over all rows per iteration -
    the reference's torch.optim.Adam); and the parity of one mapping iteration on this map against the oracle."""
    from nerf_loam_amd import pipeline as P
    pts, cos, dirs = w["points"], w["cos"], w["dirs"]
    lm = build_large_map(w, device, n_scans, spacing)
    centres, structure, vertex_idx, id2row, E, m, poses, build_s = (lm[k] for k in ("centres", "structure", "vertex_idx", "id2row", "E", "map", "poses", "build_s"))
    dec = P.DecoderDevice(*[np.asarray(a, np.float32) for a in w["host"]["dec"]], device=device)
    mid = n_scans // 2
    out = {"scans": n_scans, "spacing_m": spacing, "octree_nodes": int(len(centres)), "embedding_rows": int(E), "root_side_voxels": int(m.root_side),
           "host_build_s": round(build_s, 2), "single_scan_map_rows": w["n_rows"]}

    def loop(n_rays, n_frames, train, sparse, full=False, copies=1):
        rs = np.random.default_rng(5)
        if full:
            sel, fid = np.arange(len(pts)), np.zeros(len(pts), np.int32)
        else:
            sel = np.concatenate([np.sort(rs.choice(len(pts), n_rays, replace=False)) for _ in range(n_frames)])
            fid = np.repeat(np.arange(n_frames, dtype=np.int32), n_rays)
        # (emb_grad_copies > 1: replicated gradient accumulators - the near-sensor rows of an accumulated map take a contribution from every ray and their
        #  same-address atomics queue; measured in the *_16_accumulator_copies legs: the scatter gains less than the optimiser's fold over the copies costs, so the
        #  default stays ONE array)
        eng = P.SdfEngine(max_rays=len(sel), samples_per_ray_cap=48 if full else 96, max_frames=max(2, n_frames), device=device, sparse_adam=sparse,
                          emb_grad_copies=copies if sparse else 1)
        eng.set_rays(dirs[sel], pts[sel], cos[sel], fid)
        eng.set_poses(np.stack([poses[mid + f] for f in range(n_frames)]), [1] * n_frames)
        cfg = P.IterConfig()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tb, to, tc = [], [], []
        for rep in range(4):
            ev[0].record(); eng.begin_call(m, dec); ev[1].record()
            eng.bind(m, dec, cfg, train_decoder=train, update_decoder=train, want_pose_grad=not full or True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(iters - 1):
                eng.run_bound()
            eng.run_bound(1)
            ev[2].record(); eng.run_bound(2); ev[3].record()
            torch.cuda.synchronize()
            if rep:
                tc.append((time.perf_counter() - t1) / iters * 1e3); tb.append(ev[0].elapsed_time(ev[1])); to.append(ev[2].elapsed_time(ev[3]))
        st = eng.stats()
        assert not st["overflow"] and not eng.call_status()[2]
        r = {"ms_per_iter": float(np.median(tc)), "begin_call_ms": float(np.median(tb)), "optimiser_ms": float(np.median(to)), "valid_samples": int(st["P"]),
             "hit_rays": int(st["R"]), "max_hits": int(st["H"])}
        if sparse:
            r["touched_rows"] = int(eng._touched[1].item())
            r["emb_grad_copies"] = int(eng.emb_grad_copies)
        return r, eng

    out["mapping_2048x1"], eng_small = loop(2048, 1, True, True)
    out["mapping_2048x1_16_accumulator_copies"] = loop(2048, 1, True, True, copies=16)[0]
    out["mapping_2048x1_dense_bookkeeping"] = loop(2048, 1, True, False)[0]
    out["ba_4096x4_frozen_decoder"] = loop(4096, 4, False, True)[0]
    out["ba_4096x4_frozen_decoder_16_accumulator_copies"] = loop(4096, 4, False, True, copies=16)[0]
    out["ba_4096x4_frozen_decoder_dense_bookkeeping"] = loop(4096, 4, False, False)[0]
    out["full_scan_131072"] = loop(0, 1, True, True, full=True)[0]
    # the pose-refine step (M2) at the shipped tracker steps on this map (and on the same trajectory at kitti's 0.3 m voxels)
    trk = {}
    for name, ts in TRACKER_SETTINGS.items():
        lm_t = lm if ts["voxel"] == 0.2 else build_large_map(w, device, n_scans, spacing, voxel=ts["voxel"])
        trk[name] = tracker_step_on_map(w, lm_t, device, ts["step"], ts["lr"])[0]
        if lm_t is not lm:
            del lm_t
            torch.cuda.empty_cache()
    out["track_2048"] = trk
    # parity of one mapping iteration on this map against the oracle (2048 rays of the middle scan): geometry bit for bit, sdf / dsdf / dX
    from oracle import oracle as O
    eng = eng_small
    rs = np.random.default_rng(5)
    sel = np.sort(rs.choice(len(pts), 2048, replace=False))
    emb_now = m.emb.cpu().numpy().view(np.uint16).copy()
    dn = dec.numpy()
    pose_now = eng.pose6[0].cpu().numpy().copy()
    eng.forward_backward(m, dec, P.IterConfig(), train_decoder=True)
    torch.cuda.synchronize()
    st = eng.stats()
    P_ = st["P"]
    got = dict(hit_count=eng.hit_count[:2048].cpu().numpy(), depth=eng.s_depth[:P_].cpu().numpy(), vox=eng.s_vox[:P_].cpu().numpy(),
               sdf=eng.sdf[:P_].cpu().numpy(), dsdf=eng.dsdf[:P_].cpu().numpy(), dX=eng.dX[:P_].cpu().numpy())
    ms_o = O.MapState(centres, structure, vertex_idx, id2row, emb_now, 0.2)
    dp = O.DecoderParams(dn["W1"], dn["b1"], dn["W2"], dn["b2"], dn["W3"], dn["b3"])
    fr = O.Frame(dirs[sel], pts[sel], cos[sel], pose_now)
    ref = O.render_and_grad(ms_o, dp, [fr], O.IterCfg(), want_emb_grad=False, want_dec_grad=False)
    rr, ss = np.nonzero(ref["valid"])
    geom = bool(P_ == ref["n_samples"] and np.array_equal(got["hit_count"] > 0, ref["hits"]) and np.array_equal(got["depth"], ref["z_vals"][rr, ss])
                and np.array_equal(got["vox"], ref["s_idx"][rr, ss]))
    par = {"rays": 2048, "valid_samples": int(P_), "geometry_bit_exact": geom}
    if geom:
        par.update(sdf_max_abs_err=float(np.abs(got["sdf"] - ref["sdf"][rr, ss]).max()),
                   dsdf_max_err_rel_to_max=float(np.abs(got["dsdf"] - ref["dsdf"][rr, ss]).max() / max(np.abs(ref["dsdf"]).max(), 1e-30)),
                   dX_rel_l2=float(np.linalg.norm((got["dX"] - ref["dfeat"]).astype(np.float64)) / max(np.linalg.norm(ref["dfeat"].astype(np.float64)), 1e-30)))
        par["ok"] = bool(par["sdf_max_abs_err"] < 1e-4 and par["dsdf_max_err_rel_to_max"] < 1e-3 and par["dX_rel_l2"] < 1e-3)
    else:
        par["ok"] = False
    out["parity_vs_oracle"] = par
    return out


# the reference's other shipped settings (configs/kitti/kitti.yaml: voxel 0.3, mapper step 0.5 x 0.3; configs/ncd/ncd.yaml: voxel 0.2, mapper
# step 0.2 x 0.2 = 0.04 m: up to ~60 samples per ray) on the same 64 x 2048 scan
SETTINGS = {"kitti": dict(voxel=0.3, step=0.15, lrs=(0.01, 0.005, 0.001)), "ncd": dict(voxel=0.2, step=0.04, lrs=(0.002, 0.005, 0.001))}


def settings_bench(device, iters=4, with_parity=True):
    """one full-scan mapping iteration under the kitti and ncd settings: ms per iteration (one C call per iteration), samples per ray, and the
    in-run oracle check of parity_check on that configuration - so that a regression at 58 samples per ray shows in the bench line"""
    from nerf_loam_amd import pipeline as P
    out = {}
    for name, sp in SETTINGS.items():
        w = build_workload(device, voxel=sp["voxel"])
        N = len(w["points"])
        eng = P.SdfEngine(max_rays=N, samples_per_ray_cap=64, device=device)
        eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
        cfg = P.IterConfig(voxel_size=sp["voxel"], step_size=sp["step"], lr_emb=sp["lrs"][0], lr_dec=sp["lrs"][1], lr_pose=sp["lrs"][2])
        eng.begin_call(w["map"], w["dec"])
        eng.bind(w["map"], w["dec"], cfg, train_decoder=True)
        for _ in range(2):
            eng.run_bound()
        blocks = []
        for _ in range(3):                                       # the best of three blocks: one block of one run measured 13.5 ms instead of 4.0 (host-side
            torch.cuda.synchronize(); t0 = time.perf_counter()   # interference right after an oracle check; the leg is there to show kernel regressions)
            for _ in range(iters):
                eng.run_bound()
            torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / iters)
        dt = min(blocks)
        st = eng.stats()
        if st["overflow"] or st["guard"] or eng.call_status()[2]:
            raise SystemExit(f"bench invalid ({name} settings): overflow={st['overflow']} guard={st['guard']}")
        r = dict(voxel_size_m=sp["voxel"], step_size_m=sp["step"], octree_nodes=w["n_nodes"], embedding_rows=w["n_rows"], ms_per_iter=dt * 1e3,
                 ms_per_iter_blocks=[b * 1e3 for b in blocks], rays_per_s=N / dt, valid_samples=int(st["P"]), samples_per_hit_ray=float(st["P"]) / max(st["R"], 1), max_samples_per_ray=int(st["S"]),
                 max_hits_per_ray=int(st["H"]))
        if with_parity:
            r["parity"] = parity_check(eng, w, cfg, True, every=16)
        out[name] = r
        del eng, w
        torch.cuda.empty_cache()
    return out


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
    eng.graph = None
    eng.bind(w["map"], w["dec"], cfg, **flags)
    for _ in range(10):
        eng.run_bound()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run_bound()
    torch.cuda.synchronize()
    out["ms_per_step_one_c_call"] = (time.perf_counter() - t0) / steps * 1e3
    # the same one-C-call iteration (7 fused launches) as a hipGraph, one iteration and a whole 25-iteration track_frame loop per graph launch.
    # (ms_per_step_graph above replays the STAGE-WISE sequence - 12 launches, profiles/r05_z_timeline_latency_bound_steps.txt sections 2 / 4 - which is why it
    #  loses to the one-C-call path: the graph holds five more ~4.7 us kernels, not a slower launch mechanism)
    try:
        for name, iters in (("ms_per_step_graph_one_c_call", 1), ("ms_per_step_graph_25_iterations", 25)):
            torch.cuda.synchronize()
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                for _ in range(iters):
                    eng.run_bound()
            for _ in range(3):
                g_.replay()
            torch.cuda.synchronize()
            n_rep = max(1, steps // iters)
            t0 = time.perf_counter()
            for _ in range(n_rep):
                g_.replay()
            torch.cuda.synchronize()
            out[name] = (time.perf_counter() - t0) / (n_rep * iters) * 1e3
            del g_
    except Exception as e:                                       # noqa: BLE001 - report, do not fail the bench
        out["graph_one_c_call_error"] = repr(e)[:200]
    # the reference's track_frame re-draws its 2048 rays every iteration (LidarFrame.sample_rays on the CPU + H2D copy):
    # same step with the rays re-drawn on the device from the resident scan (nl_select_rays)
    scan = dict(dirs=None, points=torch.from_numpy(np.ascontiguousarray(w["points"])).to(device),      # dirs None: derived from the points in the selection kernel
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


def frame_loop_bench(device, n_frames=6):
    """What a FRAME costs through the mirrored call sites (reference src/mapping.py:112-170, src/tracking.py:92-147) at the maicity settings: a full
    64 x 2048 scan per frame (the synthetic scene seen from a standing sensor, fresh range noise per frame); per frame `Tracking.do_tracking` (20 iterations x 2048 rays, pose only), `Mapping.create_voxels`
    (host octree insert + incremental export + device update) and `Mapping.do_mapping` (20 iterations x 2048 rays: embeddings + decoder + pose).  Medians over the
    frames after the first two (the first tracked frame runs 5 x the iterations, the first insert builds the whole tree)."""
    import queue
    from argparse import Namespace
    from nerf_loam_amd import synthetic as S
    from nerf_loam_amd.lidar_frame import LidarFrame
    from nerf_loam_amd.mapping import Mapping
    from nerf_loam_amd.tracking import Tracking
    args = Namespace(criteria=dict(sdf_weight=10000.0, fs_weight=1, eiko_weight=0.1, sdf_truncation=0.30), data_specs=dict(max_depth=50.0, min_depth=1.5),
                     decoder_specs=dict(depth=2, width=256, in_dim=16, skips=[], embedder="none", multires=0),
                     tracker_specs=dict(N_rays=2048, learning_rate=0.005, step_size=0.2, max_voxel_hit=20, num_iterations=20),
                     mapper_specs=dict(N_rays_each=2048, use_local_coord=False, voxel_size=0.2, step_size=0.5, window_size=4, num_iterations=20, max_voxel_hit=20,
                                       final_iter=True, mesh_res=8, learning_rate_emb=0.03, learning_rate_decorder=0.005, learning_rate_pose=0.001, freeze_frame=20,
                                       keyframe_gap=8, remove_back=False, key_distance=12),
                     debug_args=dict(verbose=False, mesh_freq=100))
    torch.manual_seed(777)
    mapper, tracker = Mapping(args), Tracking(args)
    share = Namespace(decoder=None, states=None)
    kf = queue.Queue()
    t_track, t_vox, t_map, nodes = [], [], [], []

    def clock(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    last = None
    for i in range(n_frames):
        pts, cos = S.synthetic_scan(seed=777 + i, range_noise=0.01)
        fr = LidarFrame(i, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
        if i == 0:
            _, tv = clock(lambda: mapper.create_voxels(fr))
            _, tm = clock(lambda: mapper.do_mapping(share, fr, selection_method="current"))
            tracker.last_frame = fr
        else:
            _, tt = clock(lambda: tracker.do_tracking(share, fr, kf))
            _, tv = clock(lambda: mapper.create_voxels(fr))
            _, tm = clock(lambda: mapper.do_mapping(share, fr, selection_method="current"))
            t_track.append(tt)
        t_vox.append(tv); t_map.append(tm); nodes.append(int(mapper.svo.count_nodes()))
        last = fr
    med = lambda a: float(np.median(a[2:])) if len(a) > 2 else float(np.median(a))
    frame_ms = med(t_track) + med(t_vox) + med(t_map)
    err = float(np.abs(last.pose.data.detach().cpu().numpy()[:3] - 2000.0).max())
    return {"frames": n_frames, "points_per_frame": 131072, "ms_per_frame": frame_ms, "frames_per_s": 1e3 / frame_ms,
            "track_ms": med(t_track), "create_voxels_ms": med(t_vox), "mapping_ms": med(t_map), "first_frame_create_voxels_ms": t_vox[0], "octree_nodes": nodes,
            "tracked_translation_error_m": err,
            "note": "mirrored Mapping / Tracking call sites at the maicity settings (20 + 20 iterations x 2048 rays per frame), a standing sensor with fresh 1 cm range noise per scan (the tracked pose should stay put: tracked_translation_error_m); create_voxels = host C++ octree insert "
                    "of all 131 072 returns + incremental export + device-side growth of the map tensors (no re-upload of the tree or the embedding table); the reference's process loop, "
                    "logging and queues are not part of this number"}


def get_scores_bench(w, device, res=8, reps=5):
    """f3: the mesher's dense SDF grid - render_helpers.get_scores, reference render_helpers.py:96-153 - over the SURFACE voxels of the bench map:
    res^3 points per voxel, gather (points generated in the kernel) + forward-only decoder on the matrix cores.  voxels / s, points / s and the forward kernel's
    matrix-pipe fraction (139 776 algorithmic flops per point; executed under the fp16-pair arithmetic: 3 products of the 256-deep GEMM + 4 of layer 1)."""
    from nerf_loam_amd import render_helpers as RH, _lib
    from nerf_loam_amd.decoder import Decoder
    h = w["host"]
    surf = np.nonzero(h["vertex_idx"][:, 0] >= 0)[0]
    dec = Decoder().to(device)
    dec.load_flat(torch.from_numpy(np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in h["dec"]])).to(device))
    emb = torch.from_numpy((h["emb_bits"].astype(np.uint32) << 16).view(np.float32)).to(torch.bfloat16).to(device)
    states = {"voxel_vertex_idx": torch.from_numpy(h["vertex_idx"][surf]), "voxel_center_xyz": torch.from_numpy(h["centres"][surf]),
              "voxel_structure": torch.from_numpy(h["structure"][surf]), "voxel_vertex_emb": emb, "voxel_id2embedding_id": torch.from_numpy(h["id2row"])}
    RH.get_scores(dec, states, w["voxel"], bits=res, device_out=True)          # warm-up: resident map tensors, decoder block, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g = RH.get_scores(dec, states, w["voxel"], bits=res, device_out=True)
    torch.cuda.synchronize()
    dt_dev = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    g_host = RH.get_scores(dec, states, w["voxel"], bits=res)
    dt_host = time.perf_counter() - t0
    n_pts = len(surf) * res ** 3
    # marching cubes on the grid (csrc/nl_mesh.hip; reference mesh_util.py:145-169: skimage on the CPU per voxel).  A random-init decoder's field does not change
    # sign inside the voxels, so the zero level is moved to the grid's median - the extraction's work depends on how many voxels the surface crosses, not on where
    from nerf_loam_amd import ops
    gs = (g.view(-1, res, res, res) - g.median()).contiguous()
    cdev = states["voxel_center_xyz"].to(device)
    mv, mf = ops.marching_cubes(gs, cdev, w["voxel"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        mv, mf = ops.marching_cubes(gs, cdev, w["voxel"])
    torch.cuda.synchronize()
    dt_mc = (time.perf_counter() - t0) / reps
    mc_bytes = 2 * n_pts * 4 + len(surf) * (12 + 4 * 4) + mv.numel() * 4 + mf.numel() * 4        # the grid read by both passes, counts / offsets, the mesh written once
    mc = {"ms_per_call": dt_mc * 1e3, "vertices": int(mv.shape[0]), "triangles": int(mf.shape[0]), "voxels_per_s": len(surf) / dt_mc, "triangles_per_s": mf.shape[0] / dt_mc,
          "roofline": {"bound": "hbm", "achieved": mc_bytes / dt_mc / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mc_bytes / dt_mc / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes": int(mc_bytes)},
          "note": "count pass + two prefix scans + one read-back of the totals + emit pass (host-timed whole call, mesh left on the device); zero level at the grid's median; "
                  "tests/test_gpu_mesh.py: bit-exact against oracle/mc_oracle.py (triangulation unpinned to scikit-image, which is absent here - see the oracle's header)"}
    gm = _lib.lib().nl_decoder_get_gemm_mode()
    fwd_alg = FLOPS_PER_SAMPLE_DECODER_FROZEN / 2                           # forward only: 2 (16 x 256 + 256 x 256 + 256)
    ex16 = {4: 3, 5: 4}.get(gm, 8) * G + (4 if gm >= 4 else 9) * L1
    bound_s = n_pts * ex16 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    return {"voxels": int(len(surf)), "res": res, "points": int(n_pts), "ms_per_call_device_resident_grid": dt_dev * 1e3, "ms_per_call_grid_on_host": dt_host * 1e3,
            "voxels_per_s": len(surf) / dt_dev, "points_per_s": n_pts / dt_dev, "algorithmic_tflops": n_pts * fwd_alg / dt_dev / 1e12,
            "matrix_pipe_bound_ms": bound_s * 1e3, "matrix_pipe_frac_of_the_call": bound_s / dt_dev,
            "hbm_bytes_algorithmic": int(n_pts * (64 * 2 + 4) + len(surf) * (12 + 32)), "sdf_range": [float(g_host.min()), float(g_host.max())], "marching_cubes": mc,
            "note": "whole call: map tensors -> device (cached per map), per chunk of <= 16.8 M points one nl_gather_grid + one nl_decoder_forward launch, one device-to-host copy "
                    "of the grid (ms_per_call_grid_on_host; the reference copies every 10 000 voxels and synchronises on each); tests/test_gpu_api_mirror.py holds the call "
                    "against the reference's own get_scores output (tests/golden/scores_res4.npz) at 5e-6"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle parity check")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 PMC passes behind roofline.traffic (the committed summary is quoted instead)")
    ap.add_argument("--no-api-path", action="store_true", help="skip the bundle_adjust_frames / track_frame timings")
    ap.add_argument("--no-large-map", action="store_true", help="skip the 150-scan / 1e6-row map leg")
    ap.add_argument("--no-settings", action="store_true", help="skip the kitti / ncd settings legs")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the informational 200 further steps after the timed region (the profiling scripts "
                                                                   "pass it: the profiled command's per-kernel averages then cover the timed steps only)")
    ap.add_argument("--frozen-decoder", action="store_true", help="mapping with update_decoder=False (after freeze_frame)")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) the child run rocprofv3 profiles for roofline.traffic: three iterations, no output")
    ap.add_argument("--rccl-world1", action="store_true", help="run the ray-sharded code path (RCCL communicator, exchanges inside nl_iteration) "
                                                                "on ONE GPU with a world-size-1 process group: what a 1-GPU box can check of --gpus N")
    args = ap.parse_args()
    # the cyclic garbage collector stays off for the whole run (every leg below is host-timed; no_gc() collects at the start of the main ones): its
    # generation-2 passes showed up as one ~60 ms pause per few hundred steps in every run's per-block timings, and once inside the 20 timed steps
    gc.disable()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run, the
        # same launch the driver uses); rank 0 of that job prints the line
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} devices, this box has {n_dev}")
        import socket
        import subprocess
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = {k: v for k, v in os.environ.items() if k not in _POOL_VARS_SET_HERE}      # (every rank sizes its own pools: the ranks share the node's quota)
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} processes (or none: bench.py starts them itself)")
    from nerf_loam_amd import _lib, pipeline as P, dist as D
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    shard = world > 1 or args.rccl_world1
    if shard:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        import torch.distributed as tdist
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        tdist.barrier()
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL prints a version banner through C stdio at communicator set-up: out before the JSON line

    w = build_workload(device)
    N = len(w["points"])
    lo, hi = D.shard_bounds(N, rank, world)
    eng = P.SdfEngine(max_rays=hi - lo, samples_per_ray_cap=48, device=device)
    # backend "rccl": the exchanges are issued from C inside nl_iteration; the timed iterations are hipGraph replays of that one C call
    # (BASELINE config 5), so the [pose | embedding] all-reduce rides a side stream under dW2 + slab reduction at no cost (a graph edge)
    use_graph = shard and os.environ.get("NL_BENCH_SHARD_GRAPH", "1") != "0"
    ex = D.RayShardedExchange(eng, overlap=True) if shard else None
    # balanced shards: the scan is beam-major and beams differ several-fold in voxels/samples per ray, so each rank takes every
    # world-th return instead of a block of whole beams (identity for one GPU; scripts/shard_probe.py, profiles/r01_k_shard_probe.txt)
    order = D.interleaved_order(N, world)
    sel = order[lo:hi]
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel])
    eng.set_poses(w["pose"][None], [1])
    cfg = P.IterConfig()
    train_dec = not args.frozen_decoder
    eng.begin_call(w["map"], w["dec"])

    # ONE C call per iteration (nl_iteration; ray-sharded: kernels + the four RCCL collectives on the launch stream).  The host enqueues a whole
    # step in ~50 us - the 20 timed steps are in the queue 1-2 ms after the clock starts, and nothing that happens to the launching thread
    # afterwards (a ~40 ms pause of it once doubled a run's ms_per_step when every stage was a Python call: profiles/r05_v_*) can starve the device
    eng.bind(w["map"], w["dec"], cfg, train_decoder=train_dec, update_decoder=train_dec, ray_id_base=lo)

    graph = [None]

    def step():
        graph[0].replay() if graph[0] is not None else eng.run_bound()

    def barrier():
        if shard:
            import torch.distributed as tdist
            tdist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        return
    # per-kernel events for the roofline object: recorded by nl_iteration itself on the launch stream, in front of the decoder kernel, between it
    # and the dW2 kernel and behind dW2 (NlIterDesc.ev_decoder_begin / ev_decoder_end / ev_wgrad2_end; an event is recorded once here so that its handle exists).
    # Created BEFORE the warm-up steps: nothing but the barrier stands between the W warm-up steps and the K timed ones (creating 3 K events there left the
    # device idle for a few ms in front of the timed region)
    ev = []
    for _ in range(args.steps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        for e in (e0, e1, e2):
            e.record()
        ev.append({"decoder": (e0, e1), "wgrad2": (e1, e2)})
    barrier()
    for _ in range(max(args.warmup, 2) if shard else args.warmup):   # (sharded: the first iteration sizes the row exchange of the call)
        step()
    barrier()
    graph_note = None
    if use_graph:
        try:
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                eng.run_bound()
            graph[0] = g_
            step()
            barrier()
            graph_note = "hipGraph replay of the one-C-call iteration (kernels + RCCL collectives, side-stream fork / join as graph edges)"
        except Exception as e:                                       # noqa: BLE001 - report, fall back to eager launches
            graph[0] = None
            graph_note = "eager launches (hipGraph capture failed: " + repr(e)[:160] + ")"
            torch.cuda.synchronize()
    with no_gc():
        t0 = time.perf_counter()
        for k in range(args.steps):
            eng.timers = None if shard else ev[k]          # (sharded: the scatter runs between the two decoder kernels - a stage-wise pass below times them)
            step()
        barrier()
        dt = time.perf_counter() - t0
    eng.timers = None
    # informational: the same step sustained.  The device needs ~25 ms of load to reach its steady clock (scripts/ramp_probe.py), so a 20-step
    # timed region after a few warm-up steps still sits partly on the ramp; `value` stays what the contract defines (the K steps above)
    steady = None
    if not args.no_steady_state:
        with no_gc():
            t1 = time.perf_counter()
            for _ in range(STEADY_STEPS):
                step()
            barrier()
            steady = (time.perf_counter() - t1) / STEADY_STEPS * 1e3
    # the same step with the decoder's EXACT-product arithmetic (gemm mode 3: three-term bf16 splits, eight of nine products), timed right
    # behind the sustained run (warm clock: compare with steady_state): what the default's fp16 pairs buy, and the line a caller that pins
    # exact fp32 products gets.  Selected per call (kernel_modes) - the process default is not touched.
    exact_leg = None
    if not (shard or args.no_steady_state) and _lib.lib().nl_decoder_get_gemm_mode() in (4, 5):
        km0 = eng.kernel_modes
        eng.kernel_modes = eng._desc.kernel_modes = _lib.kernel_modes(3, 1)
        for _ in range(3):
            step()
        barrier(); t1 = time.perf_counter()
        for _ in range(50):
            step()
        barrier()
        exact_leg = (time.perf_counter() - t1) / 50 * 1e3
        eng.kernel_modes = eng._desc.kernel_modes = km0
        step(); barrier()
    if shard:
        import torch.distributed as tdist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    st = eng.stats()
    if st["overflow"] or st["guard"] or eng.call_status()[2]:   # a truncated sample set would overstate the throughput
        raise SystemExit(f"bench invalid: sample overflow={st['overflow']} guard={st['guard']} call_invalid={eng.call_status()[2]}")
    P_local = st["P"]
    sharded = None
    if shard:
        import torch.distributed as tdist
        # (a) the same iterations without the communicator (every rank on its own share, no exchange): what the collectives cost;
        # (b) a short stage-wise pass with the per-kernel events for the roofline object (the hooks run the same C exchanges)
        graph[0] = None                                           # (the remaining passes launch eagerly)
        d = eng._desc
        comm_ptr, d.comm = d.comm, None
        eng._exchange = None
        for _ in range(2):
            eng.run_bound()
        barrier(); t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.run_bound()
        barrier(); dt_local = time.perf_counter() - t1
        d.comm, eng._exchange = comm_ptr, ex
        ev = [{n: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for n in ("decoder", "wgrad2")} for _ in range(5)]
        for k in range(5):
            eng.timers = ev[k]
            eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=train_dec, ray_id_base=lo)
            eng.optimiser_step(w["map"], w["dec"], cfg, update_decoder=train_dec)
        barrier()
        eng.timers = None
        gathered = [None] * world
        tdist.all_gather_object(gathered, dict(rank=rank, rays=int(hi - lo), valid_samples=int(P_local), hit_rays=int(st["R"])))
        tl = torch.tensor([dt_local], dtype=torch.float64, device=device)
        tdist.all_reduce(tl, op=tdist.ReduceOp.MAX)
        sharded = dict(rccl_world=world, exchange_backend=ex.backend, launch_mode=graph_note or "eager launches", gradient_exchange_overlapped=bool(ex.overlap), embedding_exchange=("dense" if ex._rows_cap == "dense" else f"touched rows, capacity {ex._rows_cap}"),
                       collectives_per_step="3 on the critical path (exchange 1, exchange 2, decoder gradient) + the [pose | embedding] all-reduce on a side stream under dW2", per_rank=gathered, ms_per_step_without_exchanges=float(tl.item()) / args.steps * 1e3,
                       exchange_ms_per_step=(dt - float(tl.item())) / args.steps * 1e3,
                       note="one C call per iteration (nl_iteration): ONE all-gather [counters | per-ray hit counts] after the intersect, counter all-gather "
                            "after the sampler, grouped all-reduce [fp64 pose partials | embedding accumulators] on a side stream right after the scatter "
                            "(under dW2 + slab reduction), decoder-gradient all-reduce after the join; "
                            "exchange_ms_per_step = this run minus the same iterations with the communicator removed (those launched eagerly, one C call each)")
    dec_ms = float(np.mean([e["decoder"][0].elapsed_time(e["decoder"][1]) for e in ev]))
    wg_ms = float(np.mean([e["wgrad2"][0].elapsed_time(e["wgrad2"][1]) for e in ev])) if train_dec else 0.0
    stage_ms, stage_bytes, hbm_entries = stage_rooflines(eng, w, cfg, train_dec) if not shard else ({}, {}, [])
    if rank == 0:
        gm, wm = _lib.lib().nl_decoder_get_gemm_mode(), _lib.lib().nl_decoder_get_wgrad2_mode()
        # which fused decoder kernel ran (include/nerfloam_hip.h NL_KERNEL_LAYOUT): two 4-wave workgroups per CU (k_decoder2) for the fp16-pair arithmetic above
        # 16 384 rays when the engine holds two slabs per CU, else one 8-wave workgroup (k_decoder)
        lay = ((eng.kernel_modes >> 16) & 3) or _lib.lib().nl_decoder_layout_for(int(eng.N)) or _lib.lib().nl_decoder_get_layout()
        split = gm in (4, 5) and lay != 1 and (lay == 2 or eng.n_slabs >= _lib.lib().nl_decoder_grid_hint())
        kname = ("k_decoder2" if split else "k_decoder") + ("<train>" if train_dec else "<frozen>")
        rf = roofline_entry(kname, "decoder", dec_ms, P_local, gm, wm, train_dec)
        rf["workgroups"] = "two independent 4-wave workgroups per CU (k_decoder2)" if split else "one 8-wave workgroup per CU (k_decoder)"
        rf["clock_note"] = ("`peak` prices the matrix pipes at the part's 2.4 GHz; under this kernel the chip sits at its power cap and holds ~1.9-2.0 GHz (rocm-smi: 1330 W of 1400 W, "
                            "profiles/r06_power_probe.txt), where the pipes are busy 57.5 % of the SIMD cycles (profiles/r06_y_pmc_summary.json)")
        rf["frac_of_dense_16bit_peak"] = rf["achieved"] / PEAK_BF16_MFMA_TFLOPS
        kfull = (("k_decoder2<true, %d" if train_dec else "k_decoder2<false, %d") % {4: 3, 5: 4}.get(gm, 3) if split else
                 ("k_decoder<true, %s" if train_dec else "k_decoder<false, %s") % ("true" if gm >= 1 else "false"))
        # (quick runs - --no-cpu-baseline - and runs that are themselves under a profiler skip the two nested rocprofv3 passes)
        under_profiler = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY"))
        traffic = pmc_traffic_in_run(kfull) if not (shard or args.no_pmc or args.no_cpu_baseline or under_profiler) else None
        rf = {"bound": "mfma", **rf,
              "traffic": traffic if traffic is not None else pmc_traffic(kfull + ">"),
              "traffic_source": ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes over a 3-iteration child of this command), "
                                 "(2 x FETCH_SIZE + WRITE_SIZE) KB per launch" if traffic is not None else committed_traffic_source()),
              "peak_note": ("matrix-pipe bound of the kernel's instruction mix: "
                            + (f"256-deep GEMMs as fp16 pairs ({ {4: 3, 5: 4}[gm]} of 4 forward + 2 dgrad MFMAs per fp32 product, 2500 TF pipe), layer-1 forward as four fp16 "
                               "products, dX / dW1 / db1 as three fp16-pair products each (dW1 on 32x32x16 with 17 of 32 output columns used); nothing left on the fp32 pipe" if gm >= 4 else
                               f"256-deep GEMMs as bf16 three-term splits ({ {1: 9, 3: 8, 2: 6}.get(gm, 9)} of 9 forward + 3 dgrad MFMAs per fp32 product, 2500 TF pipe), "
                               "layer-1 forward as nine bf16 products too, dX / dW1 (K=16) on the fp32 pipe (157.3 TF)" if gm >= 1 else "all GEMMs on the fp32 pipe (157.3 TF)")),
              "second_kernel": (roofline_entry({1: "k_decoder_wgrad2_x<bf16 x3>", 2: "k_decoder_wgrad2_x<fp16 pair>"}.get(wm, "k_decoder_wgrad2"), "wgrad2", wg_ms, P_local, gm, wm, True)
                                if train_dec else None)}
        out = {
            "metric": "LiDAR rays/sec per SDF iter (64x2048 scan)",
            "value": N * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": ("fp32 values and fp32 accumulation throughout; the decoder's 256-deep GEMMs run on the 16-bit matrix cores on split operands: "
                           + ("fp16 pairs - every fp32 operand (scaled by a power of two) = hi + lo, two fp16 terms that reproduce it to one fp32 rounding "
                              "(2^-23), each hi/lo product exact in fp32; the forward GEMM forms " + {4: "three of the four partial products (without lo x lo: < 2^-22 of a product)", 5: "all four partial products"}[gm]
                              + ", the dgrad is the {0,1} ReLU mask x two terms.  Measured against the oracle and the reference goldens this is indistinguishable from the exact-product "
                              "mode (exact_products: NL_GEMM_MODE=3, three-term bf16 splits)" if gm >= 4 else
                              "exact-product splits (each fp32 operand = 3 bf16 terms exactly; ReLU masks are {0,1}); the forward GEMM forms "
                              + {1: "all nine", 3: "eight of the nine (without lo x lo: < 2^-30 of a product)", 2: "six of the nine"}.get(gm, "?") + " partial products")
                           + "; NL_GEMM_MODE=0 / NL_WGRAD2_MODE=0 = the plain fp32-MFMA kernels" if (gm >= 1 or wm == 1)
                           else "fp32 MFMA kernels (NL_GEMM_MODE=0, NL_WGRAD2_MODE=0)"),
            "config": {"workload": "synthetic 64x2048 scan (131072 rays), 1 mapping iteration/step: intersect+sample+gather+"
                                   "decoder fwd/bwd+SDF loss+emb/decoder/pose grads+Adam; voxel 0.2 m, step 0.1 m, "
                                   + ("decoder trainable" if train_dec else "decoder frozen"),
                       "rays": N, "octree_nodes": w["n_nodes"], "embedding_rows": w["n_rows"], "hit_rays": st["R"],
                       "valid_samples_rank0": P_local, "max_samples_per_ray": st["S"], "parallelism": f"ray-shard x{world}" + (" (interleaved returns)" if world > 1 else ""),
                       "launch": graph_note or "one C call per step (nl_iteration: the launch sequence SdfEngine.bind / run_bound, Mapping and Tracking use); the decoder kernels' "
                                               "events are recorded by that call itself"},
            "roofline": rf,
        }
        if exact_leg is not None:
            out["exact_products"] = {"gemm_mode": 3, "ms_per_step": exact_leg, "rays_per_s": N / exact_leg * 1e3, "steps": 50,
                                     "note": "the same step with the 256-deep GEMMs as three-term bf16 splits (every product exact, eight of the nine forward products: the "
                                             "default of rounds 3-4), timed right behind steady_state on the warm device - compare with steady_state.ms_per_step; selectable per call "
                                             "(NlIterDesc.kernel_modes / SdfEngine(gemm_mode=3)) or process-wide (NL_GEMM_MODE=3)"}
        if steady is not None:
            out["steady_state"] = {"steps": STEADY_STEPS, "ms_per_step": steady, "rays_per_s": N / steady * 1e3,
                                   "note": "the same step, the next %d launches after the timed region (local time of rank 0): the device reaches its steady clock "
                                           "after ~25 ms of load (scripts/ramp_probe.py); informational - `value` is the K timed steps" % STEADY_STEPS}
        if sharded is not None:
            out["sharded"] = sharded
        if not shard:
            # bandwidth-bound stages + the whole iteration against the sum of its per-kernel bounds
            rf["hbm"] = hbm_entries
            bounds = {"decoder": rf["matrix_pipe_bound_ms"], **{e["stage"]: e["algorithmic_bytes_per_launch"] / (HBM_PEAK_GBS * 1e9) * 1e3 for e in hbm_entries}}
            if train_dec:
                bounds["wgrad2"] = rf["second_kernel"]["matrix_pipe_bound_ms"]
            rf["end_to_end"] = {"sum_of_bounds_ms": float(sum(bounds.values())), "ms_per_step": dt / args.steps * 1e3,
                                "frac": float(sum(bounds.values())) / (dt / args.steps * 1e3), "bounds_ms": bounds, "stage_ms": stage_ms,
                                "note": "decoder kernels: matrix-pipe bound of their instruction mix; every other stage: algorithmic bytes / 8 TB/s"}
            out["pose_refine"] = pose_refine_bench(w, device)      # launch-bound loops first: the oracle's BLAS threads keep spinning
            if not args.no_api_path:                               # for a while after use and slow the launching thread down
                out["shard_probe"] = shard_probe_bench(w, device, dt / args.steps * 1e3)   # (same kernels at other sizes: kept out of the
                out["api_path"] = api_path_bench(w, device)                                #  profiled command's per-kernel averages)
            if not args.no_api_path:
                out["get_scores"] = get_scores_bench(w, device)
                try:
                    out["frame_loop"] = frame_loop_bench(device)
                except Exception as e:                                   # noqa: BLE001 - informational leg: report, do not fail the bench
                    out["frame_loop"] = {"error": repr(e)[:300]}
            if not args.no_large_map:
                out["large_map"] = large_map_bench(w, device)
            if not args.no_parity:
                out["parity"] = parity_check(eng, w, cfg, train_dec)
            if not (args.no_settings or args.no_large_map):      # (--no-large-map: the headline map only - the profiling scripts pass it)
                out["settings"] = settings_bench(device, with_parity=not args.no_parity)
        if not args.no_cpu_baseline and not shard:
            out["cpu_baseline"] = cpu_baseline(w)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if shard:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
