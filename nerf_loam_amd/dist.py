"""Ray-sharded data parallelism of the SDF iteration over the GPUs of one node (RCCL over xGMI).

The reference has no distributed code.  Rays are independent up to three reductions
(SURVEY 8e), which are the three exchange points of SdfEngine.forward_backward:

  1. after intersect : all-gather (R_rank, Hmax_rank)  -> global hit-ray count / this rank's hit-rank
                       offset (the sampler's tail quirk depends on a ray's GLOBAL rank)
                       and the global max hit count.
  2. after counting  : all-reduce SUM of the loss normalisers (front / sdf mask counts, the
                       padded-slot constants) and MAX of S = max samples per ray
                       (criterion.py:84-88 weights and the R*S mean divisor are global quantities).
  3. after backward  : all-reduce SUM of one flat fp32 buffer [decoder grads | pose partials |
                       embedding-gradient accumulators]; every rank then applies the identical
                       optimiser step to its replica.

One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm); on CPU test rigs "gloo".
Exchanges 1 and 2 are a few dozen bytes (latency-bound); exchange 3 is 0.28 MB + 64 B per embedding
row.  Known deviation under sharding: the sampler's tail loop consults the hit list of the first ray
of its batch row (sample_gpu.cu:231); when that ray lives on another rank the own list is used.
"""
import torch
import torch.distributed as dist

from . import _lib as L

_SUM_INTS = [L.NLC_NFS, L.NLC_NSDF, L.NLC_INV_FS_RAYS, L.NLC_INV_FS_CNT, L.NLC_INV_SDF_RAYS, L.NLC_INV_SDF_CNT,
             L.NLC_OVERFLOW, L.NLC_GUARD]


def shard_bounds(n, rank, world):
    """contiguous block of the iteration's ray list for `rank` (SURVEY 8e partition)"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


class RayShardedExchange:
    def __init__(self, engine, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.eng = engine
        dev = engine.counters.device
        self._gather = torch.zeros(self.world, 2, dtype=torch.int32, device=dev)
        self._local2 = torch.zeros(2, dtype=torch.int32, device=dev)
        self._sum_idx = torch.tensor(_SUM_INTS, dtype=torch.long, device=dev)
        engine.hook_after_intersect = self.after_intersect
        engine.hook_after_count = self.after_count
        engine.hook_after_backward = self.after_backward

    # exchange 1
    def after_intersect(self, eng):
        c = eng.counters
        self._local2[0:1].copy_(c[L.NLC_R:L.NLC_R + 1])
        self._local2[1:2].copy_(c[L.NLC_HMAX:L.NLC_HMAX + 1])
        dist.all_gather_into_tensor(self._gather.view(-1), self._local2, group=self.group)
        g = self._gather
        c[L.NLC_R_GLOBAL:L.NLC_R_GLOBAL + 1].copy_(g[:, 0].sum(dtype=torch.int32).view(1))
        c[L.NLC_R_OFFSET:L.NLC_R_OFFSET + 1].copy_(g[:self.rank, 0].sum(dtype=torch.int32).view(1))
        c[L.NLC_HMAX:L.NLC_HMAX + 1].copy_(g[:, 1].max().view(1))

    # exchange 2
    def after_count(self, eng):
        c = eng.counters
        ints = c[self._sum_idx]
        dist.all_reduce(ints, op=dist.ReduceOp.SUM, group=self.group)
        c[self._sum_idx] = ints
        smax = c[L.NLC_SMAX:L.NLC_SMAX + 1]
        dist.all_reduce(smax, op=dist.ReduceOp.MAX, group=self.group)
        dbl = c[L.NL_CNT_INTS:].view(torch.float64)[L.NLD_INV_D2:L.NLD_INV_D2CNT + 1]
        dist.all_reduce(dbl, op=dist.ReduceOp.SUM, group=self.group)

    # exchange 3
    def after_backward(self, eng, dec, train_decoder, want_emb_grad, want_pose_grad):
        """ONE collective for all gradients: pack [decoder | pose partials | embedding accumulators] into a flat fp32
        buffer (two tiny copy kernels cost far less than two extra RCCL launches), all-reduce, unpack."""
        parts = []
        if train_decoder:
            parts.append(dec.grad.view(-1))
        if want_pose_grad:
            parts.append(eng.g_pose.view(-1))
        if want_emb_grad:
            parts.append(eng.g_emb.view(-1))
        if not parts:
            return
        if len(parts) == 1:
            dist.all_reduce(parts[0], op=dist.ReduceOp.SUM, group=self.group)
            return
        flat = torch.cat(parts)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in parts:
            p.copy_(flat[off:off + p.numel()])
            off += p.numel()

    def reduce_loss_sums(self):
        c = self.eng.counters
        dbl = c[L.NL_CNT_INTS:].view(torch.float64)[L.NLD_FS_SQ:L.NLD_SDF_SQ + 1]
        dist.all_reduce(dbl, op=dist.ReduceOp.SUM, group=self.group)
