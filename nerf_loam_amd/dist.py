"""Ray-sharded data parallelism of the SDF iteration over the GPUs of one node (RCCL over xGMI).

The reference has no distributed code.  Rays are independent up to three reductions
(SURVEY 8e), which are the three exchange points of SdfEngine.forward_backward:

  1. after intersect : all-gather (R_rank, Hmax_rank)  -> global hit-ray count / this rank's hit-rank
                       offset (the sampler's tail quirk depends on a ray's GLOBAL rank)
                       and the global max hit count.
  2. after counting  : all-reduce SUM of the loss normalisers (front / sdf mask counts, the
                       padded-slot constants) and MAX of S = max samples per ray
                       (criterion.py:84-88 weights and the R*S mean divisor are global quantities).
  3. after backward  : all-reduce SUM of one flat fp32 buffer [decoder grads | embedding-gradient
                       accumulators] and of the fp64 pose partials [F,12]; every rank then applies the
                       identical optimiser step to its replica.

One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm); on CPU test rigs "gloo".
Exchanges 1 and 2 are latency-bound, so each is ONE collective: every rank all-gathers its whole 96-byte counter block
and a one-block kernel (nl_dist_merge_counters) folds the gathered blocks into the local one (sums, max, rank offset) -
not one collective per quantity plus a dozen tiny torch kernels.  Exchange 3 all-reduces the gradients IN PLACE: the
decoder gradient and the embedding accumulators are re-homed once into one flat buffer [decoder | embeddings], so there is
nothing to pack or unpack (0.28 MB + 64 B per embedding row); the fp64 pose partials (96 B per frame) go in their own
collective, issued first so that it runs under the large one.
Known deviation under sharding: the sampler's tail loop consults the hit list of the first ray
of its batch row (sample_gpu.cu:231); when that ray lives on another rank the own list is used.
"""
import torch
import torch.distributed as dist

from . import _lib as L

assert (L.NLC_NFS, L.NLC_GUARD) == (4, 11)          # the summed counters are the contiguous slots NFS .. GUARD (nl_common.h)


def shard_bounds(n, rank, world):
    """contiguous block of the iteration's ray list for `rank` (SURVEY 8e partition)"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def interleaved_order(n, world):
    """Permutation of a ray list such that the contiguous blocks of shard_bounds() are the strided subsets r, r+world, ...:
    neighbouring LiDAR returns (same beam, adjacent azimuth) cost about the same, whole beams do not (grazing beams cross
    several times more voxels than upward ones), so contiguous blocks of a beam-major scan are badly balanced
    (profiles/r01_k_shard_probe.txt).  Identity for world == 1."""
    import numpy as np
    if world <= 1:
        return np.arange(n)
    per = (n + world - 1) // world
    idx = np.arange(per * world).reshape(per, world).T.reshape(-1)      # rank-major: r, r+world, r+2*world, ...
    return idx[idx < n]


class RayShardedExchange:
    def __init__(self, engine, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.eng = engine
        dev = engine.counters.device
        self._stride = L.NL_CNT_BYTES // 4
        self._gather = torch.zeros(self.world * self._stride, dtype=torch.int32, device=dev)
        self._flat = None
        engine.hook_after_intersect = self.after_intersect
        engine.hook_after_count = self.after_count
        engine.hook_after_backward = self.after_backward

    def _merge(self, c, stage):
        """fold the gathered counter blocks into the local block `c`"""
        if c.is_cuda:
            L.check(L.lib().nl_dist_merge_counters(L.ptr(self._gather), self.world, self.rank, stage, L.ptr(c),
                                                   torch.cuda.current_stream().cuda_stream), "nl_dist_merge_counters")
            return
        # host tensors: the gloo test rig of tests/test_dist_gloo.py (no GPU in the loop) - same arithmetic in torch
        g = self._gather.view(self.world, self._stride)
        if stage == 1:
            c[L.NLC_R_GLOBAL] = g[:, L.NLC_R].sum()
            c[L.NLC_R_OFFSET] = g[:self.rank, L.NLC_R].sum()
            c[L.NLC_HMAX] = g[:, L.NLC_HMAX].max()
        else:
            c[L.NLC_NFS:L.NLC_GUARD + 1] = g[:, L.NLC_NFS:L.NLC_GUARD + 1].sum(0)
            c[L.NLC_SMAX] = g[:, L.NLC_SMAX].max()
            gd = g[:, L.NL_CNT_INTS:].contiguous().view(torch.float64)
            c[L.NL_CNT_INTS:].view(torch.float64)[L.NLD_INV_D2:L.NLD_INV_D2CNT + 1] = gd[:, L.NLD_INV_D2:L.NLD_INV_D2CNT + 1].sum(0)

    # exchange 1
    def after_intersect(self, eng):
        dist.all_gather_into_tensor(self._gather, eng.counters, group=self.group)
        self._merge(eng.counters, 1)

    # exchange 2
    def after_count(self, eng):
        dist.all_gather_into_tensor(self._gather, eng.counters, group=self.group)
        self._merge(eng.counters, 2)

    def _adopt(self, eng, dec):
        """re-home dec.grad, eng.g_emb into ONE flat buffer (views keep shape and contents); redone when the
        engine reallocated one of them (new map size)"""
        parts = [dec.grad, eng.g_emb]
        f = self._flat
        if f is not None:
            off, ok = 0, True
            for p in parts:
                ok = ok and p.data_ptr() == f.data_ptr() + 4 * off and p.dtype == torch.float32
                off += p.numel()
            if ok and off == f.numel():
                return
        f = torch.empty(sum(p.numel() for p in parts), dtype=torch.float32, device=parts[0].device)
        off, views = 0, []
        for p in parts:
            v = f[off:off + p.numel()].view(p.shape)
            v.copy_(p)
            views.append(v)
            off += p.numel()
        dec.grad, eng.g_emb = views
        self._flat = f

    # exchange 3
    def after_backward(self, eng, dec, train_decoder, want_emb_grad, want_pose_grad):
        """the fp64 pose partials, then ONE in-place collective over the part of [decoder | embeddings] that was computed"""
        if want_pose_grad:
            dist.all_reduce(eng.g_pose[:eng.F], op=dist.ReduceOp.SUM, group=self.group)
        if not (train_decoder or want_emb_grad):
            return
        self._adopt(eng, dec)
        nd = dec.grad.numel()
        lo = 0 if train_decoder else nd
        hi = self._flat.numel() if want_emb_grad else nd
        dist.all_reduce(self._flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)

    def reduce_loss_sums(self):
        c = self.eng.counters
        dbl = c[L.NL_CNT_INTS:].view(torch.float64)[L.NLD_FS_SQ:L.NLD_SDF_SQ + 1]
        dist.all_reduce(dbl, op=dist.ReduceOp.SUM, group=self.group)
