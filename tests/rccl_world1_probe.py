"""Run by tests/test_gpu_dist_rccl.py in a process of its own (GPU box): the ray-sharded iteration on a REAL RCCL communicator of world
size 1 - torch.distributed's ProcessGroupNCCL communicator, driven from C (backend "rccl" of nerf_loam_amd/dist.py).  With one rank
every collective is the identity, so the sharded iteration must reproduce the plain one bit for bit - which exercises everything the
multi-GPU run needs short of a second device: the RCCL symbols bound from torch's librccl, ncclAllGather / ncclAllReduce / grouped
all-reduces issued on the launch stream between the kernels, one C call per iteration, the touched-rows path, and hipGraph capture +
replay of an iteration that contains collectives.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H                                  # noqa: E402
from oracle import oracle as O                       # noqa: E402  (scene construction only)
from nerf_loam_amd import dist as D, pipeline as P   # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = np.load(os.path.join(ROOT, "tests", "golden", "map_2f_2it_frozen.npz"))
    sc = H.build_oracle_scene(int(g["n_beams"]), int(g["n_azimuth"]), int(g["seed"]))
    sc["ms"].id2row = g["id_table"].copy()
    masks = H.unpack_masks(g["masks"], len(sc["points"]))
    dec_np = O.decoder_init(int(g["seed"]))
    nf = masks.shape[0]
    frames = [O.select_rays(sc["points"], sc["cos"], g["poses0"][f].copy(), masks[f][0], optimize_pose=True) for f in range(nf)]
    rays = np.concatenate([f.rays_d for f in frames]); pts = np.concatenate([f.points for f in frames]); cos = np.concatenate([f.cos for f in frames])
    fid = np.concatenate([np.full(len(f.rays_d), i, np.int32) for i, f in enumerate(frames)])
    poses = np.stack([f.pose for f in frames])
    cfg = P.IterConfig(step_size=float(g["step_size"]), noise_seed=7)
    out = {}

    def run(mode, pad_rows=0, sparse_rows="auto", iters=3, overlap=True):
        ms = sc["ms"]
        emb = ms.emb if not pad_rows else np.concatenate([ms.emb, np.zeros((pad_rows, 16), np.uint16)])
        m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, emb, ms.voxel_size)
        dec = P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
        eng = P.SdfEngine(max_rays=len(rays), samples_per_ray_cap=64, max_frames=max(2, nf))
        ex = D.RayShardedExchange(eng, sparse_rows=sparse_rows, backend="rccl", overlap=overlap) if mode != "plain" else None
        eng.set_rays(rays, pts, cos, fid)
        eng.set_poses(poses, [1] * nf)
        eng.begin_call(m, dec)
        if mode == "graph":
            eng.capture_iteration(m, dec, cfg, train_decoder=True)
            for _ in range(iters):
                eng.replay()
        elif mode == "boundgraph":                               # the ONE-C-CALL iteration (overlapped exchange: side stream, event fork / join) in a hipGraph
            eng.bind(m, dec, cfg, train_decoder=True)
            eng.run_bound()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                eng.run_bound()
            for _ in range(iters - 1):
                gr.replay()
        elif mode == "stagewise":
            for _ in range(iters):
                eng.forward_backward(m, dec, cfg, train_decoder=True)
                eng.optimiser_step(m, dec, cfg)
        else:
            eng.bind(m, dec, cfg, train_decoder=True)
            for _ in range(iters):
                eng.run_bound()
        torch.cuda.synchronize()
        st = eng.call_status()
        res = dict(params=dec.params.cpu().numpy().copy(), emb=m.emb.cpu().numpy().copy(), pose6=eng.pose6[:nf].cpu().numpy().copy(), status=st,
                   rows_cap=None if ex is None else ex._rows_cap, backend=None if ex is None else ex.backend)
        if mode == "boundgraph":
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    gr.replay()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
            res["ms_per_iter"] = float(np.median(ts))
        if mode == "onecall":                                   # what the exchanges cost per iteration on this (one-rank) communicator
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    eng.run_bound()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
            res["ms_per_iter"] = float(np.median(ts))
        return res

    plain = run("plain")
    out["steps_plain"] = plain["status"][0]
    ok = True
    for name, kw in (("onecall", {}), ("onecall_serial", dict(overlap=False)), ("stagewise", {}), ("graph", {}), ("boundgraph", {}),
                     ("onecall_rows", dict(pad_rows=400000)), ("onecall_rows_serial", dict(pad_rows=400000, overlap=False)), ("graph_rows", dict(pad_rows=400000)),
                     ("boundgraph_rows", dict(pad_rows=400000)), ("onecall_dense_forced", dict(sparse_rows=False))):
        mode = name.split("_")[0]
        r = run(mode, **{k: v for k, v in kw.items()})
        base = plain if not kw.get("pad_rows") else run("plain", pad_rows=kw["pad_rows"])
        same = {k: bool(np.array_equal(r[k], base[k])) for k in ("params", "emb", "pose6")}
        out[name] = dict(equal=same, steps=r["status"][0], invalid=bool(r["status"][2]), rows_cap=r["rows_cap"], backend=r["backend"],
                         ms_per_iter=r.get("ms_per_iter"))
        ok = ok and all(same.values()) and r["status"][0] == plain["status"][0] and not r["status"][2] and r["backend"] == "rccl"
    # one C call per iteration without a communicator, for the exchange cost
    ms = sc["ms"]
    m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, ms.emb, ms.voxel_size)
    dec = P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
    eng = P.SdfEngine(max_rays=len(rays), samples_per_ray_cap=64, max_frames=max(2, nf))
    eng.set_rays(rays, pts, cos, fid); eng.set_poses(poses, [1] * nf); eng.begin_call(m, dec); eng.bind(m, dec, cfg, train_decoder=True)
    for _ in range(5):
        eng.run_bound()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            eng.run_bound()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    out["plain_ms_per_iter"] = float(np.median(ts))
    out["ok"] = bool(ok)
    print("RCCL_WORLD1 " + json.dumps(out))
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
