"""CPU, world_size 2, gloo: the three exchange points of the ray-sharded iteration
(nerf_loam_amd/dist.py) on a stand-in engine holding CPU tensors with the real counter-block layout.
Checks the global hit-ray count / rank offset / max hits (exchange 1), the summed loss normalisers and
max samples per ray (exchange 2), the summed gradient buffers (exchange 3), and that the shard
partition covers the ray list exactly once."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_loam_amd import _lib as L
from nerf_loam_amd import dist as D


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = torch.zeros(L.NL_CNT_BYTES // 4, dtype=torch.int32)
        eng = SimpleNamespace(counters=c, g_pose=torch.full((2, 12), float(rank + 1), dtype=torch.float64), F=2, g_emb=torch.full((5, 16), 10.0 * (rank + 1)),
                              hook_after_intersect=None, hook_after_count=None, hook_after_backward=None)
        dist_ops = []
        for name in ("all_reduce", "all_gather_into_tensor"):            # record what travels: (collective, reduce op)
            def rec(*a, _f=getattr(dist, name), _n=name, **k):
                dist_ops.append((_n, str(k.get("op", ""))))
                return _f(*a, **k)
            setattr(D.dist, name, rec)
        dec = SimpleNamespace(grad=torch.arange(7, dtype=torch.float32) * (rank + 1))
        ex = D.RayShardedExchange(eng)
        assert eng.hook_after_intersect is not None
        # exchange 1
        c[L.NLC_R] = 100 + 10 * rank; c[L.NLC_HMAX] = 7 + 3 * rank
        ex.after_intersect(eng)
        r1 = (int(c[L.NLC_R_GLOBAL]), int(c[L.NLC_R_OFFSET]), int(c[L.NLC_HMAX]), int(c[L.NLC_R]))
        # exchange 2
        c[L.NLC_NFS] = 5 + rank; c[L.NLC_NSDF] = 50 + rank; c[L.NLC_INV_SDF_RAYS] = 2; c[L.NLC_INV_SDF_CNT] = 9 * (rank + 1)
        c[L.NLC_SMAX] = 19 - 4 * rank; c[L.NLC_P] = 1000 + rank
        dbl = c[L.NL_CNT_INTS:].view(torch.float64)
        dbl[L.NLD_INV_D2] = 1.5 * (rank + 1); dbl[L.NLD_INV_D2CNT] = 0.25; dbl[L.NLD_FS_SQ] = 3.0 + rank
        ex.after_count(eng)
        r2 = (int(c[L.NLC_NFS]), int(c[L.NLC_NSDF]), int(c[L.NLC_INV_SDF_RAYS]), int(c[L.NLC_INV_SDF_CNT]), int(c[L.NLC_SMAX]),
              int(c[L.NLC_P]), float(dbl[L.NLD_INV_D2]), float(dbl[L.NLD_INV_D2CNT]), float(dbl[L.NLD_FS_SQ]))
        # exchange 3
        ex.after_backward(eng, dec, True, True, True)           # decoder gradient, fp64 pose partials, dense embedding accumulators
        r3 = (dec.grad.tolist(), float(eng.g_pose[0, 0]), float(eng.g_emb[0, 0]))
        ex.reduce_loss_sums()
        r4 = float(dbl[L.NLD_FS_SQ])
        # exchange 3 over touched rows: a 1000-row table, rank 0 touched rows {3, 31, 32, 700}, rank 1 {31, 64, 999}; only all-gather and
        # SUM all-reduce travel (what ProcessGroupNCCL / RCCL implements: no bitwise reduction)
        eng.g_emb = torch.zeros(1000, 16)
        eng.touched_rows = torch.tensor([3, 31, 32, 700] if rank == 0 else [31, 64, 999])
        eng.g_emb[eng.touched_rows] = float(rank + 1)
        ex.after_backward(eng, dec, False, True, False)
        r5 = (sorted(torch.nonzero(eng.g_emb.abs().sum(1))[:, 0].tolist()), eng.g_emb[[3, 31, 32, 64, 700, 999], 0].tolist())
        assert {op for n, op in dist_ops if n == "all_reduce"} <= {"RedOpType.SUM", "ReduceOp.SUM"}, dist_ops
        out_q.put((rank, r1, r2, r3, r4, r5))
    finally:
        dist.destroy_process_group()


def test_exchanges_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, a1, a2, a3, a4, a5), (_, b1, b2, b3, b4, b5) = res
    assert a5 == b5 == ([3, 31, 32, 64, 700, 999], [1.0, 3.0, 1.0, 2.0, 1.0, 2.0])      # union of the rows, summed where both touched
    assert a1 == (210, 0, 10, 100) and b1 == (210, 100, 10, 110)            # global R, rank offsets, global Hmax, local R kept
    assert a2[:5] == b2[:5] == (11, 101, 4, 27, 19)                          # sums / max are global ...
    assert a2[6:8] == b2[6:8] == (4.5, 0.5)
    assert a2[5] == 1000 and b2[5] == 1001                                   # ... P (valid samples) stays local
    assert (a2[8], b2[8]) == (3.0, 4.0)                                      # residual sums reduced only on request
    assert a3 == b3 == ([0.0, 3.0, 6.0, 9.0, 12.0, 15.0, 18.0], 3.0, 30.0)
    assert a4 == b4 == 7.0


def test_shard_bounds_partition():
    for n in (1, 7, 131072, 131073):
        for world in (1, 2, 3, 8):
            segs = [D.shard_bounds(n, r, world) for r in range(world)]
            assert segs[0][0] == 0 and segs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
            assert all(lo <= hi for lo, hi in segs)


def test_interleaved_order_gives_strided_shards():
    for n, world in ((131072, 8), (16, 4), (10, 3), (7, 8), (5, 1)):
        p = D.interleaved_order(n, world)
        assert sorted(p.tolist()) == list(range(n))
        if n % world == 0:
            for r in range(world):
                lo, hi = D.shard_bounds(n, r, world)
                assert np.array_equal(p[lo:hi], np.arange(r, n, world))
