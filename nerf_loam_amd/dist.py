"""Ray-sharded data parallelism of the SDF iteration over the GPUs of one node (RCCL over xGMI).

The reference has no distributed code.  Rays are independent up to three reductions
(SURVEY 8e), which are the three exchange points of SdfEngine.forward_backward:

  1. after intersect : all-gather (R_rank, Hmax_rank)  -> global hit-ray count / this rank's hit-rank
                       offset (the sampler's tail quirk depends on a ray's GLOBAL rank)
                       and the global max hit count.
  2. after counting  : all-reduce SUM of the loss normalisers (front / sdf mask counts, the
                       padded-slot constants) and MAX of S = max samples per ray
                       (criterion.py:84-88 weights and the R*S mean divisor are global quantities).
  3. after backward  : all-reduce SUM of the decoder gradient (started on a side stream as soon as the slabs are
                       reduced: it runs under the embedding scatter kernel), of the fp64 pose partials [F,12] and of
                       the embedding-gradient accumulators - dense ([E,16] fp32) on a small map, or only the rows
                       the iteration touches (below); every rank then applies the identical optimiser step to
                       its replica.

One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm); on CPU test rigs "gloo".
Exchanges 1 and 2 are latency-bound, so each is ONE collective: every rank all-gathers its whole 96-byte counter block
and a one-block kernel (nl_dist_merge_counters) folds the gathered blocks into the local one (sums, max, rank offset) -
not one collective per quantity plus a dozen tiny torch kernels.  Exchange 3 all-reduces the gradients IN PLACE: the
decoder gradient and the embedding accumulators are re-homed once into one flat buffer [decoder | embeddings], so there is
nothing to pack or unpack (0.28 MB + 64 B per embedding row); the fp64 pose partials (96 B per frame) go in their own
collective, issued first so that it runs under the large one.
The sampler's tail loop consults the hit list of the first ray of a ray's batch row (sample_gpu.cu:231); under
sharding that ray may live on another rank.  Exchange 1 therefore also all-reduces the 200 x ceil(L / 800) row-first hit lists
(nl_dist_row_first: every rank fills the rows it owns, 84 B each), so the samples of a sharded run are bit-identical to the
unsharded one (tests/test_gpu_parity.py).

Embedding gradients over touched rows (a long sequence has millions of rows, an iteration touches rays x hits x 8 at most):
every rank marks the rows of the voxels its rays hit in a bitmap (E / 8 bytes), the bitmaps are OR-all-reduced, the union's rows
are packed in row order into a [capacity, 16] buffer, SUM-all-reduced and unpacked (csrc/nl_dist.hip).  The capacity is fixed per
call from the first iteration's count (x 1.5; one host read per call) - collective sizes must be known on the host - and a
device flag invalidates the call if a later iteration exceeds it (never silently).  Dense when the union is most of the table.
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
    def __init__(self, engine, group=None, sparse_rows="auto"):
        """sparse_rows: "auto" (touched-rows exchange when it moves less than half of the dense table), True, False"""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.eng = engine
        dev = engine.counters.device
        self._stride = L.NL_CNT_BYTES // 4
        self._gather = torch.zeros(self.world * self._stride, dtype=torch.int32, device=dev)
        self._flat = None
        self.sparse_rows = sparse_rows
        self._row_first = None               # [entries][1 + NL_MAX_HITS] i32
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._dec_work = None
        self._rows = None                    # touched-rows state: bitmap, prefix, buffers, capacity (per map size)
        self._rows_cap = None                # per optimisation call: None = measure at the next iteration
        engine.hook_after_intersect = self.after_intersect
        engine.hook_after_count = self.after_count
        engine.hook_after_decoder_grads = self.after_decoder_grads
        engine.hook_after_backward = self.after_backward
        engine._exchange = self

    def new_call(self):
        """a new optimisation call (new frames / ray counts): re-measure the touched-rows capacity at its first iteration"""
        self._rows_cap = None

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
        if eng.counters.is_cuda and getattr(eng, "hit_idx", None) is not None:
            # the batch rows' first-ray hit lists (the sampler's tail loop reads them): every rank fills the rows it owns
            n_total = eng.N_cap * self.world
            entries = 200 * (((n_total + 199) // 200 + 799) // 800)              # 200 batch rows x ceil(L / 800) chunks
            if self._row_first is None or self._row_first.shape[0] < entries:
                self._row_first = torch.zeros(entries, 1 + L.NL_MAX_HITS, dtype=torch.int32, device=eng.counters.device)
            L.check(L.lib().nl_dist_row_first(L.ptr(eng.counters), L.ptr(eng.hit_idx), L.ptr(eng.hit_count), L.ptr(eng.ray_of_rank),
                                              L.ptr(self._row_first), self._row_first.shape[0], L.stream_ptr()), "nl_dist_row_first")
            dist.all_reduce(self._row_first, op=dist.ReduceOp.SUM, group=self.group)
            eng.row_first = self._row_first

    # exchange 2
    def after_count(self, eng):
        dist.all_gather_into_tensor(self._gather, eng.counters, group=self.group)
        self._merge(eng.counters, 2)

    # exchange 3a: the decoder gradient, as soon as it exists - on a side stream, under the embedding scatter kernel
    def after_decoder_grads(self, eng, dec):
        if self._side is None:
            dist.all_reduce(dec.grad, op=dist.ReduceOp.SUM, group=self.group)
            return
        self._side.wait_stream(torch.cuda.current_stream(dec.grad.device))
        with torch.cuda.stream(self._side):
            dist.all_reduce(dec.grad, op=dist.ReduceOp.SUM, group=self.group)

    def _rows_state(self, eng, E):
        st = self._rows
        if st is None or st["E"] != E:
            dev = eng.g_emb.device
            nw = (E + 31) // 32
            st = dict(E=E, nw=nw, bitmap=torch.zeros(nw, dtype=torch.int32, device=dev), prefix=torch.zeros(nw, dtype=torch.int32, device=dev),
                      total=torch.zeros(1, dtype=torch.int32, device=dev), ws=torch.zeros(nw + (nw + 1023) // 1024 + 8, dtype=torch.int32, device=dev),
                      buf=None)
            self._rows = st
            self._rows_cap = None
        return st

    def _exchange_embedding_rows(self, eng, m):
        """touched rows only; returns False when the dense exchange should be used for this call"""
        E = eng.g_emb.shape[0]
        st = self._rows_state(eng, E)
        if self._rows_cap == "dense":
            return False
        lib, sp = L.lib(), L.stream_ptr()
        st["bitmap"].zero_()
        L.check(lib.nl_dist_mark_rows(eng.N, L.ptr(eng.hit_idx), L.ptr(eng.hit_count), L.ptr(m.vertex_rows), L.ptr(st["bitmap"]), sp), "nl_dist_mark_rows")
        # union of the ranks' bitmaps: all-gather + OR (ProcessGroupNCCL / RCCL has no bitwise reduction: ReduceOp.BOR raises there)
        if st.get("gathered") is None or st["gathered"].numel() != self.world * st["nw"]:
            st["gathered"] = torch.empty(self.world * st["nw"], dtype=torch.int32, device=st["bitmap"].device)
        dist.all_gather_into_tensor(st["gathered"], st["bitmap"], group=self.group)
        g = st["gathered"].view(self.world, st["nw"])
        torch.bitwise_or(g[0], g[1], out=st["bitmap"]) if self.world > 1 else st["bitmap"].copy_(g[0])
        for r in range(2, self.world):
            st["bitmap"].bitwise_or_(g[r])
        L.check(lib.nl_dist_rows_prefix(L.ptr(st["bitmap"]), st["nw"], L.ptr(st["prefix"]), L.ptr(st["total"]), L.ptr(st["ws"]), sp), "nl_dist_rows_prefix")
        if self._rows_cap is None:                       # first iteration of a call: ONE host read, the capacity of the whole call
            u = int(st["total"].item())
            cap = min(E, -(-int(1.5 * u + 1024) // 4096) * 4096)
            if self.sparse_rows is False or (self.sparse_rows == "auto" and 2 * cap > E):
                self._rows_cap = "dense"
                return False
            self._rows_cap = cap
            if st["buf"] is None or st["buf"].shape[0] < cap:
                st["buf"] = torch.zeros(cap, L.NL_C, dtype=torch.float32, device=eng.g_emb.device)
        cap = self._rows_cap
        buf = st["buf"][:cap]
        buf.zero_()
        fail = eng.adam_state[3:4]                       # latched "call invalid" word (SdfEngine.call_status)
        L.check(lib.nl_dist_rows_move(0, L.ptr(st["bitmap"]), L.ptr(st["prefix"]), st["nw"], L.ptr(eng.g_emb), L.ptr(buf), cap, L.ptr(fail), sp),
                "nl_dist_rows_move")
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        L.check(lib.nl_dist_rows_move(1, L.ptr(st["bitmap"]), L.ptr(st["prefix"]), st["nw"], L.ptr(eng.g_emb), L.ptr(buf), cap, L.ptr(fail), sp),
                "nl_dist_rows_move")
        return True

    # exchange 3b
    def after_backward(self, eng, dec, train_decoder, want_emb_grad, want_pose_grad, m=None):
        """the fp64 pose partials, the embedding accumulators (touched rows or dense), and the join with the decoder all-reduce"""
        if want_pose_grad:
            dist.all_reduce(eng.g_pose[:eng.F], op=dist.ReduceOp.SUM, group=self.group)
        if want_emb_grad:
            m = m if m is not None else getattr(eng, "_map_for_exchange", None)
            sparse = (m is not None and eng.g_emb.is_cuda and self.sparse_rows is not False and self._exchange_embedding_rows(eng, m))
            if not sparse:
                dist.all_reduce(eng.g_emb, op=dist.ReduceOp.SUM, group=self.group)
        if train_decoder and self._side is not None:
            torch.cuda.current_stream(dec.grad.device).wait_stream(self._side)

    def reduce_loss_sums(self):
        c = self.eng.counters
        dbl = c[L.NL_CNT_INTS:].view(torch.float64)[L.NLD_FS_SQ:L.NLD_SDF_SQ + 1]
        dist.all_reduce(dbl, op=dist.ReduceOp.SUM, group=self.group)
