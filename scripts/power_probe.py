"""Is the fused decoder kernel at the chip's power limit?  Loops one kernel (decoder / dW2 / scatter / intersect stage of the bench workload) for a few seconds
and samples rocm-smi (socket power, sclk) from a second thread; prints the samples and the kernel's mean time in that window.  GPU only.
Usage: power_probe.py [seconds per kernel]"""
import os, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, ops, pipeline as P
L.require_gpu()
dev = torch.device("cuda")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
w = bench.build_workload(dev)
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
torch.cuda.synchronize()
dec = w["dec"]


def smi():
    out = {}
    for args in (["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"],):
        try:
            r = subprocess.run(args, capture_output=True, text=True, timeout=5)
            out["raw"] = r.stdout.strip()[:1500] if r.returncode == 0 else ("rc %d: " % r.returncode) + (r.stderr or r.stdout)[:300]
        except Exception as e:      # noqa: BLE001
            out["raw"] = repr(e)[:200]
    return out


def k_decoder():
    ops.decoder_fwd_bwd(eng.loss_scalars, eng.X, dec.params, dec.W2T, eng.s_ray, eng.s_depth, eng.cos_gt, eng.gt_dist, eng.sdf, eng.dsdf, eng.dX, eng.partials, eng.relu2_mask,
                        eng.n_slabs, 1, eng.counters, eng.kernel_modes)


def k_wgrad2():
    ops.decoder_wgrad2(eng.loss_scalars, eng.X, dec.params, eng.dsdf, eng.relu2_mask, eng.partials, eng.n_slabs, eng.kernel_modes)


def k_scatter():
    m = w["map"]
    ops.trilinear_bwd(eng.loss_scalars, eng.s_vox, eng.s_depth, eng.s_ray, eng.rays_d_world, eng.rays_d_sensor, eng.frame_id, eng.poses12, eng.F, m.centres, m.vertex_rows, m.emb,
                      m.voxel_size, eng.dX, eng.g_emb, eng.g_pose, 2 * eng.field_blocks, eng._touched)


eng.bind(w["map"], w["dec"], cfg, train_decoder=True, want_emb_grad=True, update_decoder=True, update_emb=True)
try:
    print("power cap:", subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=5).stdout.strip()[:300], flush=True)
except Exception as e:      # noqa: BLE001
    print("power cap: unavailable", repr(e)[:100])
print("idle:", smi()["raw"][:900], flush=True)
for name, fn in (("k_decoder2<train> alone", k_decoder), ("k_decoder_wgrad2_x alone", k_wgrad2), ("k_trilinear_bwd alone", k_scatter), ("whole iteration", eng.run_bound)):
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), smi()["raw"]))
            time.sleep(0.25)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    print(f"== {name}: {dt / n * 1e3:.4f} ms per launch over {dt:.1f} s, {len(samples)} rocm-smi samples", flush=True)
    for t, raw in samples[:: max(1, len(samples) // 6)]:
        print(f"   t+{t - t0:5.2f}s  {raw[:700]}")
