"""Ray-sharded data parallelism of the SDF iteration over the GPUs of one node (RCCL over xGMI).

The reference has no distributed code.  Rays are independent up to three reductions:

  1. after intersect : the sampler's tail quirk depends on a ray's GLOBAL hit rank and on the hit COUNT of the first ray of its batch row
                       (sample_gpu.cu:224-237 tests `pts_idx[curr_bin] == -1` on that ray's packed hit list): ONE all-gather of
                       [96-byte counter block | a byte per ray: its hit count] (-> global hit-ray count, this rank's hit-rank offset,
                       global max hits, and - a prefix sum over the gathered bytes - the row-first table, computed by every rank for
                       itself).  With it the samples of a sharded run are bit-identical to the unsharded run.
  2. after sampling  : criterion.py:84-88 weights and the R*S mean divisor are global: all-gather of the counter blocks (loss
                       normalisers SUM, max samples per ray MAX), the touched-rows bitmaps riding along.
  3. gradients       : a grouped SUM all-reduce of the fp64 pose partials [F,12] and the embedding accumulators - dense ([E,16] fp32) on a
                       small map, or only the rows the iteration touches - issued right after the embedding scatter on a SIDE stream, under
                       the dW2 kernel and the slab reduction (event fork / join inside nl_iteration); then the decoder gradient's
                       all-reduce (282 KB).  Every rank then applies the identical optimiser step to its replica.
  Three collectives on an iteration's critical path + one hidden under compute.

Where the exchanges run.  On the GPU they are issued FROM C, on the stream the kernels run on (csrc/nl_exchange.cpp: nl_exchange_* /
inside nl_iteration), through a four-function communicator (NlComm, include/nerfloam_hip.h):

  backend "rccl"  : librccl's ncclAllGather / ncclAllReduce / ncclGroupStart / ncclGroupEnd on the ncclComm_t torch.distributed's
                    ProcessGroupNCCL already holds (`_comm_ptr()`), resolved from the RCCL already loaded in the process.  A sharded
                    iteration is ONE C call with no host work between its launches, and hipGraph-capturable (SdfEngine.capture_iteration).
  backend "torch" : the same four entry points as ctypes callbacks onto torch.distributed collectives over the registered buffers - any
                    process group works (and the virtual-rank tests' in-process stand-in).  Not capturable.

Only all-gather and SUM all-reduce are used: ProcessGroupNCCL / RCCL has no bitwise reductions, the union of the touched-rows bitmaps is
an all-gather + OR kernel.  Host tensors (the gloo test rig of tests/test_dist_gloo.py, no GPU in the loop) take the same exchanges
through torch.distributed with the merge arithmetic in torch.

Embedding gradients over touched rows (a long sequence has millions of rows, an iteration touches rays x hits x 8 at most): every rank
marks the rows of the voxels its rays hit in a bitmap (E / 8 bytes, sent with exchange 2), the union's rows are packed in row order (prefix
sum of the word popcounts: identical on every rank) into a [capacity, 16] buffer, SUM-all-reduced and unpacked.  The capacity is fixed
per call from the first iteration's count (x 1.5; ONE host read per call, between the backward pass and the gradient exchange of the first
iteration: collective sizes must be known on the host) and a device flag invalidates the call if a later iteration exceeds it (never
silently).  Dense when the union is most of the table (the single-scan bench map)."""
import ctypes

import torch
import torch.distributed as dist

from . import _lib as L

assert (L.NLC_NFS, L.NLC_GUARD) == (4, 11)          # the summed counters are the contiguous slots NFS .. GUARD (nl_common.h)
CNT_STRIDE = L.NL_CNT_BYTES // 4                    # ints per counter block


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


def row_first_entries(n_rays_total):
    """table size of nl_dist_x1_merge's row-first table: 200 batch rows x ceil(L / 800) chunks, L = rays per batch row"""
    return 200 * (((n_rays_total + 199) // 200 + 799) // 800)


class _TorchComm:
    """NlComm backend "torch": the four communicator entry points as ctypes callbacks onto torch.distributed collectives.  The C side
    passes raw device pointers; the exchange registers every buffer it hands to the library, a pointer is resolved to the registered
    tensor that contains it."""

    def __init__(self, group, world, rank):
        self.group, self.world, self.rank = group, world, rank
        self.buffers = {}                    # data_ptr -> flat tensor
        self.error = None
        self._cb = (L.NL_COMM_ALL_GATHER(self._all_gather), L.NL_COMM_ALL_REDUCE(self._all_reduce), L.NL_COMM_GROUP(self._group),
                    L.NL_COMM_GROUP(self._group))
        self.struct = L.NlComm(world, rank, None, *self._cb)

    def register(self, *tensors):
        for t in tensors:
            if t is not None:
                self.buffers[t.data_ptr()] = t.detach().reshape(-1)

    def _view(self, ptr, nbytes):
        for base, t in self.buffers.items():
            end = base + t.numel() * t.element_size()
            if base <= ptr and ptr + nbytes <= end:
                off = (ptr - base) // t.element_size()
                return t[off:off + nbytes // t.element_size()]
        raise L.NerfLoamHipError(f"communicator: pointer {ptr:#x} (+{nbytes}) is not inside a registered buffer")

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:            # noqa: BLE001 - a callback must not raise through C; re-raised by RayShardedExchange._check
            self.error = e
            return 2

    @staticmethod
    def _on(stream, t):
        """the collective is enqueued on the stream the C side names (the launch stream, or the side stream of the overlapped gradient
        exchange): torch.distributed orders its work after torch's CURRENT stream"""
        import contextlib
        return torch.cuda.stream(torch.cuda.ExternalStream(int(stream))) if (stream and t.is_cuda) else contextlib.nullcontext()

    def _all_gather(self, ctx, send, recv, nbytes, stream):
        def run():
            s = self._view(send, nbytes)
            r = self._view(recv, nbytes * self.world)
            assert s.dtype == r.dtype
            with self._on(stream, s):
                dist.all_gather_into_tensor(r, s, group=self.group)
        return self._guard(run)

    def _all_reduce(self, ctx, buf, count, dtype, stream):
        def run():
            size = {L.NL_COMM_F32: 4, L.NL_COMM_F64: 8, L.NL_COMM_I32: 4}[dtype]
            t = self._view(buf, count * size)
            assert t.dtype == {L.NL_COMM_F32: torch.float32, L.NL_COMM_F64: torch.float64, L.NL_COMM_I32: torch.int32}[dtype]
            with self._on(stream, t):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return self._guard(run)

    def _group(self, ctx):
        return 0


class RayShardedExchange:
    def __init__(self, engine, group=None, sparse_rows="auto", backend="auto", overlap=True):
        """sparse_rows: "auto" (touched-rows exchange when it moves less than half of the dense table), True, False.
        backend: "rccl" | "torch" | "auto" (rccl when the process group is ProcessGroupNCCL and exposes its communicator).
        overlap: the [pose partials | embedding accumulators] all-reduce leaves on a side stream right after the scatter, under the dW2
        kernel and the slab reduction (False: every exchange on the launch stream - same results bit for bit).  Replayed as a hipGraph
        (bench.py --gpus N) the event fork / join is a graph edge and costs nothing; issued eagerly it costs ~8 us per iteration on a
        one-rank communicator (scripts/timeline_probe.py sections 6-9: 0.449 -> 0.457 ms eager, 0.443 -> 0.440 ms replayed).  The side
        stream has DEFAULT priority: a high-priority one makes every kernel of the launch stream measure 2x slower on this ROCm
        (0.92 ms per iteration, NL_COMM_STREAM_PRIORITY=high reproduces it)."""
        if getattr(engine, "_emb_copies_wanted", 1) != 1:
            raise L.NerfLoamHipError("a ray-sharded engine exchanges ONE embedding-gradient array: build it with emb_grad_copies=1 (the default)")
        self.group = group
        self.overlap = bool(overlap)
        self._overlap_handles = None
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.eng = engine
        self.sparse_rows = sparse_rows
        self._rows = None                    # touched-rows state: bitmap, prefix, buffers (per map size)
        self._rows_cap = None                # per optimisation call: None = measure at the next iteration, "dense", or the capacity
        self.device = engine.counters.is_cuda
        self._gather = torch.zeros(self.world * CNT_STRIDE, dtype=torch.int32, device=engine.counters.device)
        engine.hook_after_intersect = self.after_intersect
        engine.hook_after_count = self.after_count
        engine.hook_after_backward = self.after_backward
        engine._exchange = self
        if self.device:
            self._setup_device(backend)

    # ------------------------------------------------------------------ device side: communicator + descriptor fields
    def _setup_device(self, backend):
        eng = self.eng
        dev = eng.counters.device
        self._torch_comm = None
        requested = backend
        if backend == "auto":
            backend = "rccl" if self._nccl_comm_ptr() else "torch"
        if backend == "rccl":
            ptr = self._nccl_comm_ptr()
            if not ptr:
                raise L.NerfLoamHipError("backend 'rccl' needs a ProcessGroupNCCL with an initialised communicator "
                                         "(init_process_group('nccl', device_id=...)); use backend='torch' otherwise")
            self.comm = L.NlComm()
            rc = L.lib().nl_comm_init_rccl(ctypes.byref(self.comm), ctypes.c_void_p(ptr), self.world, self.rank)
            if rc == 3 and requested == "auto":
                # the library binds only an RCCL that is ALREADY loaded in the process (never a second copy): none found -> the same four
                # entry points over torch.distributed
                backend = "torch"
            else:
                L.check(rc, "nl_comm_init_rccl")
        if backend == "rccl":
            pass
        elif backend == "torch":
            self._torch_comm = _TorchComm(self.group, self.world, self.rank)
            self.comm = self._torch_comm.struct
        else:
            raise ValueError(backend)
        self.backend = backend
        # exchange 1: every rank sends [counter block | a byte per ray]; the all-gather needs ONE block size, so the ranks agree on the
        # largest ray capacity among them (one host collective at set-up)
        cap = torch.tensor([eng.N_cap], dtype=torch.int32, device=dev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=self.group)
        self._x1_rays = (int(cap.item()) + 15) & ~15
        self._x1_stride = CNT_STRIDE * 4 + self._x1_rays
        self._x1_send = torch.zeros(self._x1_stride // 4, dtype=torch.int32, device=dev)
        self._x1_recv = torch.zeros(self.world * self._x1_stride // 4, dtype=torch.int32, device=dev)
        self._entries = row_first_entries(self._x1_rays * self.world)
        self._row_first = torch.zeros(self._entries, 1 + L.NL_MAX_HITS, dtype=torch.int32, device=dev)
        eng.row_first = self._row_first
        if self.overlap:
            h = (ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p())
            L.check(L.lib().nl_overlap_create(ctypes.byref(h[0]), ctypes.byref(h[1]), ctypes.byref(h[2])), "nl_overlap_create")
            self._overlap_handles = h
        self._xg_stride = CNT_STRIDE
        self._xg_send = None
        self._xg_recv = self._gather
        self._fill_desc()

    def __del__(self):
        h, self._overlap_handles = getattr(self, "_overlap_handles", None), None
        if h is not None:
            try:
                L.lib().nl_overlap_destroy(h[0], h[1], h[2])
            except Exception:                # noqa: BLE001 - interpreter shutdown
                pass

    def _nccl_comm_ptr(self):
        try:
            if dist.get_backend(self.group) != "nccl":
                return 0
            from torch.distributed import distributed_c10d as c10d
            pg = self.group if self.group is not None else c10d._get_default_group()
            be = pg._get_backend(self.eng.counters.device)
            return int(be._comm_ptr())
        except Exception:                    # noqa: BLE001 - no RCCL communicator to borrow: the torch backend carries the exchanges
            return 0

    def _fill_desc(self):
        d, eng = self.eng._desc, self.eng
        d.comm = ctypes.cast(ctypes.pointer(self.comm), ctypes.c_void_p)
        d.xg_send = None if self._xg_send is None else self._xg_send.data_ptr()
        d.xg_recv, d.xg_stride = self._xg_recv.data_ptr(), self._xg_stride
        d.row_first, d.row_first_entries = self._row_first.data_ptr(), self._entries
        d.x1_send, d.x1_recv, d.x1_stride_bytes, d.x1_rays = self._x1_send.data_ptr(), self._x1_recv.data_ptr(), self._x1_stride, self._x1_rays
        h = self._overlap_handles
        d.comm_stream, d.ev_fork, d.ev_join = (None, None, None) if h is None else (h[0].value, h[1].value, h[2].value)
        st = self._rows
        if st is None:
            d.rows_mode, d.rows_bitmap, d.rows_prefix, d.rows_total, d.rows_ws, d.rows_buf, d.rows_cap, d.rows_words = 0, None, None, None, None, None, 0, 0
        else:
            d.rows_bitmap, d.rows_prefix, d.rows_total, d.rows_ws = (st[k].data_ptr() for k in ("bitmap", "prefix", "total", "ws"))
            d.rows_words = st["nw"]
            if isinstance(self._rows_cap, int):
                d.rows_mode, d.rows_buf, d.rows_cap = 1, st["buf"].data_ptr(), self._rows_cap
            else:                                                          # undecided: measure (mode 1 computes the union); dense: mode 0
                d.rows_mode, d.rows_buf, d.rows_cap = (1 if self._rows_cap is None else 0), None, 0
        if self._torch_comm is not None:
            tc = self._torch_comm
            tc.buffers.clear()
            tc.register(eng.counters, self._xg_recv, self._xg_send, self._x1_send, self._x1_recv, eng.g_pose, eng.g_emb, None if st is None else st.get("buf"))
            dec = getattr(eng, "_dec_for_exchange", None)
            if dec is not None:
                tc.register(dec.grad)

    def _check(self, rc, what):
        if rc != 0 and self._torch_comm is not None and self._torch_comm.error is not None:
            e, self._torch_comm.error = self._torch_comm.error, None
            raise e
        L.check(rc, what)

    def new_call(self):
        """a new optimisation call (new frames / ray counts): re-measure the touched-rows capacity at its first iteration"""
        self._rows_cap = None

    def prepare(self, m, dec, want_emb_grad):
        """(re)size the touched-rows state for the map's table and point the engine's descriptor at the exchange buffers; called by
        SdfEngine.bind / forward_backward before the first exchange of an iteration"""
        eng = self.eng
        eng._dec_for_exchange = dec
        if not self.device:
            return
        if want_emb_grad and self.sparse_rows is not False and eng.g_emb is not None:
            E = eng.g_emb.shape[0]
            st = self._rows
            if st is None or st["E"] != E:
                dev = eng.g_emb.device
                nw = ((E + 31) // 32 + 1) & ~1                          # even: the doubles of the gathered counter blocks stay aligned
                st = dict(E=E, nw=nw, bitmap=torch.zeros(nw, dtype=torch.int32, device=dev), prefix=torch.zeros(nw, dtype=torch.int32, device=dev),
                          total=torch.zeros(1, dtype=torch.int32, device=dev), ws=torch.zeros(nw + (nw + 1023) // 1024 + 8, dtype=torch.int32, device=dev),
                          buf=None)
                self._rows = st
                self._rows_cap = None
                self._xg_stride = CNT_STRIDE + nw
                self._xg_send = torch.zeros(self._xg_stride, dtype=torch.int32, device=dev)
                self._xg_recv = torch.zeros(self.world * self._xg_stride, dtype=torch.int32, device=dev)
        elif self.sparse_rows is False:
            self._rows_cap = "dense"
        self._fill_desc()

    def rows_undecided(self):
        return bool(self.device and self._rows is not None and self._rows_cap is None and self.eng._desc.want_emb_grad)

    def decide_rows(self):
        """first iteration of a call, after its backward pass: ONE host read of the union's row count fixes the exchange of the whole
        call - dense, or touched rows with capacity 1.5 x the count"""
        st = self._rows
        E = st["E"]
        u = int(st["total"].item())
        cap = min(E, -(-int(1.5 * u + 1024) // 4096) * 4096)
        if self.sparse_rows == "auto" and 2 * cap > E:
            self._rows_cap = "dense"
        else:
            self._rows_cap = cap
            if st["buf"] is None or st["buf"].shape[0] < cap:
                st["buf"] = torch.zeros(cap, L.NL_C, dtype=torch.float32, device=self.eng.g_emb.device)
        self._fill_desc()

    # ------------------------------------------------------------------ host-tensor rig (gloo tests): the merge arithmetic in torch
    def _merge_host(self, c, stage):
        g = self._gather.view(self.world, CNT_STRIDE)
        if stage == 1:
            c[L.NLC_R_GLOBAL] = g[:, L.NLC_R].sum()
            c[L.NLC_R_OFFSET] = g[:self.rank, L.NLC_R].sum()
            c[L.NLC_HMAX] = g[:, L.NLC_HMAX].max()
        else:
            c[L.NLC_NFS:L.NLC_GUARD + 1] = g[:, L.NLC_NFS:L.NLC_GUARD + 1].sum(0)
            c[L.NLC_SMAX] = g[:, L.NLC_SMAX].max()
            gd = g[:, L.NL_CNT_INTS:].contiguous().view(torch.float64)
            c[L.NL_CNT_INTS:].view(torch.float64)[L.NLD_INV_D2:L.NLD_INV_D2CNT + 1] = gd[:, L.NLD_INV_D2:L.NLD_INV_D2CNT + 1].sum(0)

    def _merge(self, c, stage):
        """fold the gathered counter blocks (self._gather) into the block `c`: the kernel on device tensors, torch on host tensors"""
        if c.is_cuda:
            L.check(L.lib().nl_dist_merge_counters(L.ptr(self._gather), self.world, self.rank, stage, L.ptr(c), L.stream_ptr()), "nl_dist_merge_counters")
        else:
            RayShardedExchange._merge_host(self, c, stage)

    def host_touched_rows_exchange(self, g_emb, touched_rows):
        """the touched-rows protocol on host tensors: bitmap of `touched_rows` -> all-gather + OR -> rows of the union packed in row order
        -> SUM all-reduce -> unpacked.  Same steps as nl_dist_mark_rows / nl_dist_rows_union_prefix / nl_dist_rows_move + the collectives
        of nl_exchange.cpp; returns the number of rows exchanged."""
        E = g_emb.shape[0]
        nw = (E + 31) // 32
        bits = torch.zeros(nw * 32, dtype=torch.bool)
        bits[touched_rows] = True
        w = (bits.view(nw, 32).to(torch.int64) << torch.arange(32, dtype=torch.int64)).sum(1)
        w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)                       # the bitmap words, as the int32 the device holds
        gathered = torch.zeros(self.world * nw, dtype=torch.int32)
        dist.all_gather_into_tensor(gathered, w, group=self.group)
        union = gathered.view(self.world, nw)[0].clone()
        for r in range(1, self.world):
            union |= gathered.view(self.world, nw)[r]
        u64 = torch.where(union < 0, union.to(torch.int64) + 2 ** 32, union.to(torch.int64))
        rows = torch.nonzero(((u64[:, None] >> torch.arange(32, dtype=torch.int64)) & 1).reshape(-1))[:, 0]
        rows = rows[rows < E]
        buf = g_emb[rows].contiguous()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        g_emb[rows] = buf
        return int(rows.numel())

    # ------------------------------------------------------------------ the three exchange points (stage-wise path + host rig)
    def after_intersect(self, eng):
        if self.device:
            eng._desc.N = eng.N
            self._check(L.lib().nl_exchange_after_intersect(ctypes.byref(eng._desc), L.stream_ptr()), "nl_exchange_after_intersect")
            return
        dist.all_gather_into_tensor(self._gather, eng.counters, group=self.group)
        self._merge_host(eng.counters, 1)

    def after_count(self, eng):
        if self.device:
            eng._desc.N = eng.N
            self._check(L.lib().nl_exchange_after_sampling(ctypes.byref(eng._desc), L.stream_ptr()), "nl_exchange_after_sampling")
            return
        dist.all_gather_into_tensor(self._gather, eng.counters, group=self.group)
        self._merge_host(eng.counters, 2)

    def after_backward(self, eng, dec, train_decoder, want_emb_grad, want_pose_grad, m=None):
        if self.device:
            if self.rows_undecided():
                self.decide_rows()
                eng._desc_touched()                              # dense exchange -> dense optimiser sweep for this call
            eng._desc.F = eng.F
            self._check(L.lib().nl_exchange_gradients(ctypes.byref(eng._desc), L.stream_ptr()), "nl_exchange_gradients")
            return
        if train_decoder:
            dist.all_reduce(dec.grad, op=dist.ReduceOp.SUM, group=self.group)
        if want_pose_grad:
            dist.all_reduce(eng.g_pose[:eng.F], op=dist.ReduceOp.SUM, group=self.group)
        if want_emb_grad:
            touched = getattr(eng, "touched_rows", None)
            if touched is not None and self.sparse_rows is not False:
                self.host_touched_rows_exchange(eng.g_emb, touched)
            else:
                dist.all_reduce(eng.g_emb, op=dist.ReduceOp.SUM, group=self.group)

    def reduce_loss_sums(self):
        """the squared-residual sums of the loss VALUE (logging only; the gradients never need them)"""
        c = self.eng.counters
        dbl = c[L.NL_CNT_INTS:].view(torch.float64)[L.NLD_FS_SQ:L.NLD_SDF_SQ + 1]
        dist.all_reduce(dbl, op=dist.ReduceOp.SUM, group=self.group)
