"""SdfEngine: the device-resident NeRF-LOAM SDF iteration on MI355X.

One `iteration()` = what the reference does per pass of its hot loops
(/root/reference/src/variations/render_helpers.py:356-425 bundle_adjust_frames and :452-512
track_frame): ray set-up -> octree intersect -> inverse-CDF sampling -> embedding gather ->
decoder forward -> SDF loss -> backward (decoder, embeddings, SE3 pose) -> Adam.

MI355X-first structure: every stage is a hand-written HIP kernel launched through the C ABI
(ops.py); all tensors live in HBM for the whole call; data-dependent sizes (hit rays R, max hits,
max samples S, valid samples P, loss normalisers) stay in a device counter block, so an iteration
runs WITHOUT a host synchronisation and with a fixed launch sequence (hipGraph-capturable).
The only torch ops on the path are a memset and two 4-byte device copies.

Multi-GPU (dist.py): rays are sharded; nl_iteration issues the exchanges itself (communicator in the descriptor), the stage-wise
path reaches the same C exchange functions through the `hook_*` points.
"""
import ctypes
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib as L
from . import ops

I32, F32 = torch.int32, torch.float32


@dataclass
class IterConfig:
    """Hyper-parameters of one optimisation call (reference config keys in brackets)."""
    voxel_size: float = 0.2            # mapper_specs.voxel_size
    step_size: float = 0.1             # metres: mapper 0.5*voxel (mapping.py:61), tracker 0.2*voxel (tracking.py:36)
    max_distance: float = 50.0         # data_specs.max_depth
    truncation: float = 0.30           # criteria.sdf_truncation
    sdf_weight: float = 10000.0        # criteria.sdf_weight
    fs_weight: float = 1.0             # criteria.fs_weight
    lr_emb: float = 0.03               # mapper_specs.learning_rate_emb
    lr_dec: float = 0.005              # mapper_specs.learning_rate_decorder
    lr_pose: float = 0.001             # mapper_specs.learning_rate_pose
    noise_seed: int = 777              # None -> deterministic 0.5 noise (voxel_helpers.py:298-299)
    tail_always: bool = False          # False = reference sampler tail behaviour


EMB_COPIES_BYTES = 4 << 30      # most memory the replicated embedding-gradient accumulators of an engine may take (SdfEngine(emb_grad_copies=))
BYTES_PER_SAMPLE = 184         # per-sample workspace of an engine: voxel, depth, dist, ray, X[16], dX[16], sdf, dsdf, 8 ReLU words


def samples_per_ray_bound(voxel_size, step_size, max_hits=L.NL_MAX_HITS):
    """Most valid samples one ray can produce (reference sampler, voxel_helpers.py:572-577 + sample_gpu.cu:165-238): a ray keeps at most
    `max_hits` voxels, a straight line stays inside a cube of side v for at most sqrt(3) v, the sampler takes
    steps = sum(len) / step_size stratified samples (+1: the ceil) and one closing sample per hit interval.  An engine whose sample
    workspace holds N times this many samples cannot overflow, whatever the map: maicity tracker (0.2 m / 0.04 m) 195, kitti tracker
    (0.3 / 0.06) 195, ncd tracker (0.2 / 0.02) 368, the mappers (step = voxel / 2) 91."""
    import math
    return int(math.ceil(max_hits * math.sqrt(3.0) * float(voxel_size) / float(step_size))) + int(max_hits) + 1




def pack_children_blocks(centres, structure):
    """Children-block traversal layout of nl_ray_intersect (device tensors in, device tensors out).
    Block 0 = pseudo block with the root in slot 0; every node with a listed child owns one block; blocks are
    numbered breadth-first so that the children blocks of a block are consecutive in octant order.
      blk_ids[B,8]   i32: child node ids (-1 = none)
      blk_hdr[B,2]   i32: (first child block or -1, exist mask | own-a-block mask << 8)
    The kernel recomputes node centres from the lattice path, which needs node min-corners to be multiples of their
    side (true for the octree: Morton prefixes) and centres = (xyz + side/2) * voxel_size (mapping.py:322)."""
    dev = centres.device
    n = centres.shape[0]
    child = structure[:, :8].long()                                   # [n,8]
    interior = (child > -1).any(1)
    ids, hdrs = [], []
    # pseudo block 0
    i0 = torch.full((1, 8), -1, dtype=torch.int32, device=dev); i0[0, 0] = 0
    has_root = bool(interior[0])
    hdrs.append(torch.tensor([[1 if has_root else -1, 0x101 if has_root else 1]], dtype=torch.int32, device=dev))
    ids.append(i0)
    frontier = torch.zeros(1, dtype=torch.long, device=dev) if has_root else torch.zeros(0, dtype=torch.long, device=dev)
    next_index = 1
    bits = (1 << torch.arange(8, device=dev)).view(1, 8)
    while frontier.numel() > 0:
        ch = child[frontier]                                           # [F,8]
        exist = ch > -1
        chc = ch.clamp(min=0)
        owns = exist & interior[chc]                                   # children that own a block themselves
        cnt = owns.sum(1)
        first = next_index + frontier.numel() + torch.cumsum(cnt, 0) - cnt      # blocks of this level come first
        ids.append(torch.where(exist, chc, torch.full_like(chc, -1)).to(torch.int32))
        hdrs.append(torch.stack([torch.where(cnt > 0, first, torch.full_like(first, -1)),
                                 (exist * bits).sum(1) + ((owns * bits).sum(1) << 8)], 1).to(torch.int32))
        next_index += frontier.numel()
        frontier = chc[owns]                                           # parent-major, octant-minor: consecutive per parent
    blk_ids, blk_hdr = torch.cat(ids).contiguous(), torch.cat(hdrs).contiguous()
    # Single-child chain under the root: the lattice is 262 144 voxels wide, a map a few hundred metres - the first ~8-10 levels of the tree
    # have ONE child each.  Their slab tests need no memory (the geometry follows from the octants), so the traversal kernel evaluates them
    # in registers and starts its work-list at the chain's end: one round trip per level saved on a pure latency chain.  The chain lives in the
    # pseudo block's unused id slots: [1] = length, [2] / [3] = the octants (3 bits each, level 0 first), [4] = the block the work-list starts
    # with, [5..7] = its lattice position.  (Blocks are numbered breadth-first: the chain's blocks are 1, 2, 3, ...)
    if has_root:
        head = blk_hdr[:40].cpu().numpy()
        cs, b, pos, octs = root_side_of(structure) >> 1, 1, [0, 0, 0], []
        while cs > 1 and b < len(head) and len(octs) < 20:
            has, exist = (int(head[b, 1]) >> 8) & 255, int(head[b, 1]) & 255
            if has == 0 or has & (has - 1) or exist != has:           # no child block, several, or a child without a block of its own: the chain ends here
                break
            u = has.bit_length() - 1
            octs.append(u)
            pos = [pos[0] + (cs if u & 1 else 0), pos[1] + (cs if u & 2 else 0), pos[2] + (cs if u & 4 else 0)]
            b, cs = int(head[b, 0]), cs >> 1
        packed = sum(u << (3 * i) for i, u in enumerate(octs))
        blk_ids[0, 1:8] = torch.tensor([len(octs), packed & 0x3FFFFFFF, packed >> 30, b, pos[0], pos[1], pos[2]], dtype=torch.int32, device=dev)
    return blk_ids, blk_hdr


def root_side_of(structure):
    return int(structure[0, 8])


class MapDevice:
    """Device copy of the reference's `map_states` (mapping.py:319-339).  The 2e9-row CPU id table
    (mapping.py:76) is folded, once per map update, into vertex_rows[n,8] = row of each corner."""

    def __init__(self, centres, structure, vertex_idx, id2row, emb_bf16_bits, voxel_size, device="cuda"):
        self.voxel_size = float(voxel_size)
        self.centres = torch.as_tensor(np.ascontiguousarray(centres, np.float32)).to(device)
        self.structure = torch.as_tensor(np.ascontiguousarray(structure, np.int32)).to(device)
        vi = np.asarray(vertex_idx)
        rows = np.where(vi >= 0, np.asarray(id2row)[np.maximum(vi, 0)], -1).astype(np.int32)
        # voxels that are never sampled (non-SURFACE nodes) keep row 0 so a stray read stays in bounds
        self.vertex_rows = torch.as_tensor(np.maximum(rows, 0)).to(device)
        self.emb = torch.as_tensor(np.ascontiguousarray(emb_bf16_bits).view(np.int16)).to(device)   # bf16 bit patterns
        self.blk_ids, self.blk_hdr = pack_children_blocks(self.centres, self.structure)
        self.root_side = int(self.structure[0, 8])
        self.n_nodes = self.centres.shape[0]
        self.n_rows = self.emb.shape[0]

    def emb_bits(self):
        return self.emb.cpu().numpy().view(np.uint16)

    def isect_lanes_for(self, n_rays):
        """lanes per ray of the intersect's work-list (nl_ray_intersect_lanes): 32 on an accumulated map - a ray crosses many occupied voxels and
        has more than 16 nodes pending per round (at 16 384 rays: 150 scans / 300 k children blocks 120 -> 95 us, 40 scans / 71 k blocks 97 -> 90;
        15 scans / 38 k blocks 68 -> 80, a one-scan map 41 -> 65) -, 0 = by ray count otherwise (32 lanes up to 4096 rays on every map, 16 up to
        16 384, 8 beyond)."""
        blk = getattr(self, "blk_hdr", None)
        return int(L.lib().nl_isect_lanes_for(int(n_rays), int(blk.shape[0]) if blk is not None else 0))     # (the one table: csrc/nl_common.h)

    @classmethod
    def from_tensors(cls, centres, structure, vertex_idx, id2row, emb_bf16, voxel_size, device="cuda", traversal=True, blocks=None):
        """Build from the reference's `map_states` tensors.  `emb_bf16` is the caller's bfloat16 CUDA parameter:
        it is ALIASED (viewed as int16 bit patterns), so the kernels update it in place like the reference's
        optimiser does.  `id2row` is the node-id -> embedding-row table (any [>=n] or [>=n,1] int tensor)."""
        self = cls.__new__(cls)
        self.voxel_size = float(voxel_size)
        dev = torch.device(device)
        self.centres = centres.detach().to(dev, torch.float32).contiguous()
        self.structure = structure.detach().to(dev, torch.int32).contiguous()
        n = self.centres.shape[0]
        vi = vertex_idx.detach().to(dev).long()
        table = id2row.detach().reshape(-1).to(dev).long() if id2row.numel() < (1 << 28) else None
        if table is None:                                   # the reference's 2e9-row host table: gather on its device
            rows = id2row.reshape(-1)[vi.clamp(min=0).cpu()].to(dev).long()
        else:
            rows = table[vi.clamp(min=0)]
        self.vertex_rows = torch.where(vi >= 0, rows, torch.zeros_like(rows)).clamp(min=0).to(torch.int32).contiguous()
        if emb_bf16.dtype != torch.bfloat16 or not emb_bf16.is_cuda:
            raise L.NerfLoamHipError("voxel_vertex_emb must be a CUDA bfloat16 tensor")
        self.emb = emb_bf16.detach().view(torch.int16)
        if traversal:                                      # per-voxel subsets (mesh-time queries) carry no usable tree
            if blocks is not None:                         # packed by the octree itself (svo.Octree.pack_blocks: one C call instead of ~200 launches)
                self.blk_ids = torch.as_tensor(blocks[0]).to(dev, torch.int32).contiguous()
                self.blk_hdr = torch.as_tensor(blocks[1]).to(dev, torch.int32).contiguous()
            else:
                self.blk_ids, self.blk_hdr = pack_children_blocks(self.centres, self.structure)
            self.root_side = int(self.structure[0, 8])
        self.n_nodes, self.n_rows = n, self.emb.shape[0]
        return self


class DecoderDevice:
    """Decoder parameter block (lidar.py:105-107 layers) + transposed W2 + Adam state."""

    def __init__(self, W1, b1, W2, b2, W3, b3, device="cuda"):
        flat = np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in (W1, b1, W2, b2, W3, b3)])
        assert flat.size == L.NL_DEC_PARAMS
        self.params = torch.as_tensor(flat).to(device)
        self.W2T = torch.zeros(L.NL_DEC_WS_FLOATS, dtype=F32, device=device)       # (zeros: the range block's sticky status word starts clear)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.grad = torch.zeros_like(self.params)
        self.refresh()

    @classmethod
    def from_flat(cls, flat):
        """from a device tensor holding the parameter block (W1 b1 W2 b2 W3 b3): no host round trip"""
        self = cls.__new__(cls)
        assert flat.numel() == L.NL_DEC_PARAMS and flat.is_cuda
        self.params = flat.detach().to(F32).contiguous().clone()
        self.W2T = torch.zeros(L.NL_DEC_WS_FLOATS, dtype=F32, device=flat.device)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.grad = torch.zeros_like(self.params)
        self.refresh()
        return self

    def refresh(self):
        ops.decoder_transpose_w2(self.params, self.W2T)

    def reset_state(self):
        self.m.zero_()
        self.v.zero_()
        self.W2T[L.NL_DEC_WS_RANGE_STATUS:L.NL_DEC_WS_RANGE_STATUS + 1].zero_()       # a new call starts with a clear range status

    def range_status(self, clear=False):
        """NL_SAT_* bits (include/nerfloam_hip.h) raised since the last clear by the decoder kernels under the fp16-pair arithmetic: an operand left - or, for the
        two bound-based bits, may have left - the range its scaled fp16 pair holds, and was clipped.  Synchronises; optimisation calls get the same information
        with their one status read-back (SdfEngine.call_status: .saturated)."""
        out = ctypes.c_uint(0)
        L.check(L.lib().nl_decoder_range_status(L.ptr(self.W2T), ctypes.byref(out), int(clear), ops.stream_ptr()), "nl_decoder_range_status")
        return int(out.value)

    def numpy(self):
        f = self.params.cpu().numpy()
        return dict(W1=f[L.OFF_W1:L.OFF_B1].reshape(256, 16), b1=f[L.OFF_B1:L.OFF_W2], W2=f[L.OFF_W2:L.OFF_B2].reshape(256, 256),
                    b2=f[L.OFF_B2:L.OFF_W3], W3=f[L.OFF_W3:L.OFF_B3].reshape(1, 256), b3=f[L.OFF_B3:])

    @staticmethod
    def split(flat):
        f = flat
        return dict(W1=f[L.OFF_W1:L.OFF_B1].reshape(256, 16), b1=f[L.OFF_B1:L.OFF_W2], W2=f[L.OFF_W2:L.OFF_B2].reshape(256, 256),
                    b2=f[L.OFF_B2:L.OFF_W3], W3=f[L.OFF_W3:L.OFF_B3].reshape(1, 256), b3=f[L.OFF_B3:])


class SdfEngine:
    def __init__(self, max_rays, samples_per_ray_cap=48, max_frames=8, device="cuda", gemm_mode=None, wgrad2_mode=None, sparse_adam=True, emb_grad_copies=1,
                 dec_layout=None):
        """gemm_mode / wgrad2_mode: the decoder kernel selection of THIS engine (include/nerfloam_hip.h: 0 fp32 matrix cores,
        1 exact-product bf16 splits, ...); None = the process default (NL_GEMM_MODE / NL_WGRAD2_MODE).  Carried per call
        (NL_KERNEL_MODES), so engines with different selections coexist in one process.  dec_layout (NL_KERNEL_LAYOUT): 1 = one 8-wave
        decoder workgroup per CU, 2 = two independent 4-wave workgroups, None = the library's rule (2 when the slabs allow)."""
        L.require_gpu()
        # sparse_adam: the embedding optimiser sweeps the rows touched since begin_call (bit-identical to the reference's dense sweep,
        # include/nerfloam_hip.h NlTouchedRows) and begin_call clears only those - False = dense sweep + E-sized memset per call
        self.sparse_adam = bool(sparse_adam)
        # emb_grad_copies > 1: replicated gradient accumulators (NlTouchedRows.copies) - the scatter's waves add into different copies, the optimiser's
        # sweep over the touched rows sums them: what same-address atomics on the near-sensor rows of an accumulated map cost is divided by the copies
        # (150-scan map, 16 copies: DESIGN.md 4.7).  Only with the touched-rows optimiser, and not on a ray-sharded engine (the exchange reads one array).
        self.emb_grad_copies = self._emb_copies_wanted = max(1, int(emb_grad_copies)) if self.sparse_adam else 1
        self._touched, self._emb_cap, self._emb_dirty_dense = None, 0, False
        self.saturated = False                              # set by call_status*: the call's decoder kernels flagged an operand outside the fp16-pair range
        self.kernel_modes = L.kernel_modes(gemm_mode, wgrad2_mode, dec_layout)
        self.dev = torch.device(device)
        self.N_cap = int(max_rays)
        self.samples_per_ray_cap, self.samples_clipped = int(samples_per_ray_cap), False     # (samples_per_ray_bound: what no call can exceed)
        self.P_cap = int(max_rays) * int(samples_per_ray_cap)
        self.F_cap = int(max_frames)
        d = self.dev
        N, P = self.N_cap, self.P_cap
        # per-ray inputs
        self.rays_d_sensor = torch.zeros(N, 3, dtype=F32, device=d)
        self.points_gt = torch.zeros(N, 3, dtype=F32, device=d)
        self.cos_gt = torch.zeros(N, dtype=F32, device=d)
        self.frame_id = torch.zeros(N, dtype=I32, device=d)
        # poses
        # pose6 [F_cap,6] and the per-frame optimise flags in ONE buffer: set_poses is one host-to-device copy
        self._pose_io = torch.zeros(self.F_cap * 7, dtype=I32, device=d)
        self.pose6 = self._pose_io[:self.F_cap * 6].view(F32).view(self.F_cap, 6)
        self.poses12 = torch.zeros(self.F_cap, 12, dtype=F32, device=d)
        # the small per-call optimiser state lives in ONE buffer, cleared by one memset per call (begin_call):
        # [g_pose f64 F x 12 | adam_state 28 i32 | pose_m F x 6 | pose_v F x 6 | pose_grad6 F x 6]
        Fc = self.F_cap
        self._call_state = torch.zeros(Fc * 24 + L.NL_ADAM_STATE_BYTES // 4 + 3 * Fc * 6, dtype=I32, device=d)
        o = Fc * 24
        self.adam_state = self._call_state[o:o + L.NL_ADAM_STATE_BYTES // 4]
        o += L.NL_ADAM_STATE_BYTES // 4
        self.pose_m = self._call_state[o:o + Fc * 6].view(F32).view(Fc, 6)
        self.pose_v = self._call_state[o + Fc * 6:o + 2 * Fc * 6].view(F32).view(Fc, 6)
        self.pose_enable = self._pose_io[self.F_cap * 6:]
        self.g_pose = self._call_state[:Fc * 24].view(torch.float64).view(Fc, 12)     # fp64 accumulators (nl_trilinear_bwd)
        self.pose_grad6 = self._call_state[o + 2 * Fc * 6:o + 3 * Fc * 6].view(F32).view(Fc, 6)
        # per-ray workspace
        self.rays_d_world = torch.empty(N, 3, dtype=F32, device=d)
        self.gt_dist = torch.empty(N, dtype=F32, device=d)
        self.hit_idx = torch.empty(N, L.NL_MAX_HITS, dtype=I32, device=d)
        self.hit_t0 = torch.empty(N, L.NL_MAX_HITS, dtype=F32, device=d)
        self.hit_t1 = torch.empty(N, L.NL_MAX_HITS, dtype=F32, device=d)
        self.hit_count = torch.zeros(N, dtype=I32, device=d)
        self.hit_rank = torch.zeros(N, dtype=I32, device=d)
        self.ray_of_rank = torch.zeros(N, dtype=I32, device=d)
        self.samp_count = torch.zeros(N, dtype=I32, device=d)
        self.samp_off = torch.zeros(N, dtype=I32, device=d)
        self.scan_ws = torch.zeros(max(1, (N + 1023) // 1024) + 8, dtype=I32, device=d)
        # per-sample workspace
        self.s_vox = torch.empty(P, dtype=I32, device=d)
        self.s_depth = torch.empty(P, dtype=F32, device=d)
        self.s_dist = torch.empty(P, dtype=F32, device=d)
        self.s_ray = torch.empty(P, dtype=I32, device=d)
        self.X = torch.empty(P, L.NL_C, dtype=F32, device=d)
        self.dX = torch.empty(P, L.NL_C, dtype=F32, device=d)
        self.sdf = torch.empty(P, dtype=F32, device=d)
        self.dsdf = torch.empty(P, dtype=F32, device=d)
        self.relu2_mask = torch.empty(P * 8 + 1024, dtype=I32, device=d)     # [tiles][512] ReLU bit words
        # counters + loss scalars
        self._counters2 = torch.zeros(2 * (L.NL_CNT_BYTES // 4), dtype=I32, device=d)
        self.counters = self._counters2[:L.NL_CNT_BYTES // 4]           # the live block the kernels count into
        self.counters_copy = self._counters2[L.NL_CNT_BYTES // 4:]      # run_bound(): the finished iteration's block (live one cleared)
        self._stats_from_copy = False
        self.sample_state = torch.zeros(1 + max(1, (N + 31) // 32), dtype=torch.int64, device=d)   # launch counter + look-back words of the one-launch sampler
        self.loss_scalars = torch.zeros(L.NL_LOSS_SCALARS_BYTES // 4, dtype=I32, device=d)
        # decoder partial slabs (one per persistent workgroup)
        hint = int(L.lib().nl_decoder_grid_hint())
        self.n_slabs = int(os.environ.get("NL_N_SLABS") or hint)       # (override: scripts/sequence_diag.py, the summation partition of the decoder gradient)
        self.partials = torch.zeros(self.n_slabs, L.NL_DEC_PARAMS, dtype=F32, device=d)
        self.field_blocks = 4 * hint
        self.N = 0
        self.F = 1
        self.g_emb = None
        self.emb_m = None
        self.emb_v = None
        # (adam_state: device step counter + hyper-parameters, a slice of _call_state above)
        self.graph = None
        # multi-GPU hooks (dist.py installs them); identity on one GPU
        self.row_first = None    # multi-GPU: table of the batch rows' first-ray hit lists (dist.py, nl_dist_x1_merge)
        self._exchange = None    # multi-GPU: dist.RayShardedExchange (its communicator rides in the descriptor: nl_iteration exchanges itself)
        self.hook_after_intersect = None
        self.hook_after_count = None
        self.hook_after_backward = None
        self.timers = None       # bench: {stage: (start, end)} torch.cuda.Event pairs, see _mark
        self._desc = L.NlIterDesc()
        self._bound = None

    # ------------------------------------------------------------------ inputs
    def set_rays(self, rays_d_sensor, points_gt, cos_gt, frame_id=None):
        """Upload (or copy on device) the iteration's ray list: the reference's per-iteration
        `frame.rays_d[mask]`, `frame.points[mask]`, `frame.pointsCos[mask]` (render_helpers.py:366-388)."""
        n = rays_d_sensor.shape[0]
        if n > self.N_cap:
            raise L.NerfLoamHipError(f"{n} rays exceed engine capacity {self.N_cap}")
        self.N = n
        self.rays_d_sensor[:n].copy_(torch.as_tensor(rays_d_sensor, dtype=F32), non_blocking=True)
        self.points_gt[:n].copy_(torch.as_tensor(points_gt, dtype=F32), non_blocking=True)
        self.cos_gt[:n].copy_(torch.as_tensor(cos_gt, dtype=F32), non_blocking=True)
        if frame_id is None:
            self.frame_id[:n].zero_()
        else:
            self.frame_id[:n].copy_(torch.as_tensor(frame_id, dtype=I32), non_blocking=True)

    def select_rays(self, scans, n_rays, seed, want_masks=False):
        """On-device ray selection (LidarFrame.sample_rays, lidarFrame.py:55-57): for every frame in `scans`
        (dicts of DEVICE tensors points [M,3], cos [M], dirs [M,3] or None = derived from the points in the kernel, lidarFrame.py:47-52)
        a uniformly random subset of n_rays returns, dataset
        order kept, gathered straight into the engine's ray buffers - no host RNG, no top-k on the CPU, no H2D copy.
        Deterministic in (seed, M, n_rays).  Returns the boolean masks (device) when want_masks."""
        total = 0
        masks = []
        # every frame in the same two launches when the shapes allow it (n << M: the live configurations)
        Ms = [int(sc["points"].shape[0]) for sc in scans]
        ns = [min(int(n_rays), M) for M in Ms]
        if sum(ns) > self.N_cap:
            raise L.NerfLoamHipError(f"{sum(ns)} rays exceed engine capacity {self.N_cap}")
        if len(scans) <= L.NL_SEL_MAX_FRAMES and all((sc.get("dirs") is None or sc["dirs"].is_contiguous()) and sc["points"].is_contiguous() for sc in scans):
            self._selb_frames(len(scans))
            mks = [sc.get("mask_u8") if sc.get("mask_u8") is not None else (torch.empty(M, dtype=torch.uint8, device=self.dev) if want_masks else None)
                   for sc, M in zip(scans, Ms)]
            offs = [sum(ns[:f]) for f in range(len(ns))]
            if ops.select_rays_batch(Ms, ns, [(int(seed) * 1000003 + f) & 0xFFFFFFFF for f in range(len(scans))],
                                     [sc.get("dirs") for sc in scans], [sc["points"] for sc in scans], [sc["cos"] for sc in scans], mks, offs,
                                     self.rays_d_sensor, self.points_gt, self.cos_gt, self.frame_id, self._selb_ws, self._selb_parity,
                                     self.adam_state[3:4]):
                self._selb_parity ^= 1
                self.N = sum(ns)
                return mks if want_masks else None
        for f, sc in enumerate(scans):
            M = int(sc["points"].shape[0])
            n = min(int(n_rays), M)
            if total + n > self.N_cap:
                raise L.NerfLoamHipError(f"{total + n} rays exceed engine capacity {self.N_cap}")
            need = 264 + 2 * M + (M + 1023) // 1024 + 8
            if getattr(self, "_sel_ws", None) is None or self._sel_ws.numel() < need:
                self._sel_ws = torch.empty(need, dtype=I32, device=self.dev)
            mask = sc.get("mask_u8")                        # a caller-owned [M] uint8 buffer receives the boolean sample mask
            if mask is None and want_masks:
                mask = torch.empty(M, dtype=torch.uint8, device=self.dev)
            ops.select_rays(M, n, (int(seed) * 1000003 + f) & 0xFFFFFFFF, sc.get("dirs"), sc["points"], sc["cos"], f,
                            self.rays_d_sensor[total:], self.points_gt[total:], self.cos_gt[total:], self.frame_id[total:],
                            mask, self._sel_ws)
            masks.append(mask)
            total += n
        self.N = total
        return masks if want_masks else None

    def _selb_frames(self, F):
        """workspace of nl_select_rays_batch.  A call clears the OTHER-parity candidate counter only of the frames it includes, so
        when the frame count of this engine changes both counters of every frame are cleared (a frame skipped for an odd number
        of calls would otherwise meet a stale counter)."""
        if getattr(self, "_selb_ws", None) is None:
            self._selb_ws = torch.zeros(L.NL_SEL_MAX_FRAMES * L.NL_SEL_BATCH_WS_INTS_PER_FRAME, dtype=I32, device=self.dev)
            self._selb_parity = 0
        elif getattr(self, "_selb_F", F) != F:
            self._selb_ws.view(L.NL_SEL_MAX_FRAMES, -1)[:, :2].zero_()
        self._selb_F = F

    def prepare_selection(self, scans, n_rays):
        """marshal the frame list of a call ONCE (select_rays re-builds its argument arrays on every call); then reselect(seed) is
        one C call per iteration.  Returns False when the shapes need the per-frame radix path (use select_rays then)."""
        F = len(scans)
        Ms = [int(sc["points"].shape[0]) for sc in scans]
        ns = [min(int(n_rays), M) for M in Ms]
        if F > L.NL_SEL_MAX_FRAMES or sum(ns) > self.N_cap or any(sc.get("mask_u8") is None for sc in scans):
            return False
        self._selb_frames(F)
        I, U, PP = ctypes.c_int * F, ctypes.c_uint * F, ctypes.c_void_p * F
        dptr = lambda sc: None if sc.get("dirs") is None else sc["dirs"].data_ptr()          # noqa: E731  (NULL: directions from the points)
        self._sel_prepared = dict(
            F=F, M=I(*Ms), n=I(*ns), seed=U(*([0] * F)), d=PP(*[dptr(sc) for sc in scans]),
            p=PP(*[sc["points"].data_ptr() for sc in scans]), c=PP(*[sc["cos"].data_ptr() for sc in scans]),
            mk=PP(*[sc["mask_u8"].data_ptr() for sc in scans]), off=I(*[sum(ns[:f]) for f in range(F)]), total=sum(ns), keep=scans)
        return True

    def reselect(self, seed):
        q = self._sel_prepared
        for f in range(q["F"]):
            q["seed"][f] = (int(seed) * 1000003 + f) & 0xFFFFFFFF
        rc = L.lib().nl_select_rays_batch(q["F"], q["M"], q["n"], q["seed"], q["d"], q["p"], q["c"], q["mk"], q["off"],
                                          self.rays_d_sensor.data_ptr(), self.points_gt.data_ptr(), self.cos_gt.data_ptr(),
                                          self.frame_id.data_ptr(), self._selb_ws.data_ptr(), self._selb_parity,
                                          self.adam_state.data_ptr() + 12, L.stream_ptr())
        if rc == 4:                                            # shapes outside the window method's range: nothing was launched
            return False
        L.check(rc, "nl_select_rays_batch")
        self._selb_parity ^= 1
        self.N = q["total"]
        return True

    def predraw(self, scans, n_rays, seeds):
        """Draw the ray subsets of ALL iterations of a call up front: (iteration, frame) pairs go through nl_select_rays_batch_ex eight at a
        time - 2 launches per eight pairs instead of 2 per iteration inside the loop (at 2048 rays x 1 frame the two selection launches
        were 10 % of an iteration).  The same subsets as reselect(seed) iteration by iteration: same keys, same seeds.  use_predrawn(it)
        then points the iteration descriptor at iteration it's slice.  Returns False when the shapes need the per-frame radix path."""
        F, iters = len(scans), len(seeds)
        Ms = [int(sc["points"].shape[0]) for sc in scans]
        ns = [min(int(n_rays), M) for M in Ms]
        tot = sum(ns)
        dptr = lambda sc: None if sc.get("dirs") is None else sc["dirs"].data_ptr()          # noqa: E731  (NULL: directions from the points)
        if tot > self.N_cap or F > L.NL_SEL_MAX_FRAMES or any(n >= M for n, M in zip(ns, Ms)) or \
                not all(sc["points"].is_contiguous() and (sc.get("dirs") is None or sc["dirs"].is_contiguous()) for sc in scans):
            return False
        d = self.dev
        pre = getattr(self, "_pre", None)
        if pre is None or pre["iters"] < iters or pre["tot"] != tot or pre["Ms"] != Ms:
            pre = dict(iters=iters, tot=tot, Ms=Ms, d=torch.empty(iters * tot, 3, dtype=F32, device=d), p=torch.empty(iters * tot, 3, dtype=F32, device=d),
                       c=torch.empty(iters * tot, dtype=F32, device=d), f=torch.empty(iters * tot, dtype=I32, device=d),
                       masks=[torch.empty(iters, M, dtype=torch.uint8, device=d) for M in Ms],
                       ws=torch.zeros(-(-iters * F // L.NL_SEL_MAX_FRAMES) * L.NL_SEL_MAX_FRAMES * L.NL_SEL_BATCH_WS_INTS_PER_FRAME, dtype=I32, device=d),
                       parity=0)
            self._pre = pre
        lib, sp = L.lib(), L.stream_ptr()
        per = L.NL_SEL_BATCH_WS_INTS_PER_FRAME
        key = (iters, tuple(ns), tuple(dptr(sc) for sc in scans), tuple(sc["points"].data_ptr() for sc in scans),
               tuple(sc["cos"].data_ptr() for sc in scans))
        if pre.get("key") != key:
            # the marshalled argument arrays of every chunk of eight (iteration, frame) pairs: built once per configuration, only the
            # seeds change from call to call
            pairs = [(it, f) for it in range(iters) for f in range(F)]
            offs = [sum(ns[:f]) for f in range(F)]
            mbase = [mk.data_ptr() for mk in pre["masks"]]
            chunks = []
            for c0 in range(0, len(pairs), L.NL_SEL_MAX_FRAMES):
                chunk = pairs[c0:c0 + L.NL_SEL_MAX_FRAMES]
                n = len(chunk)
                I, U, PP = ctypes.c_int * n, ctypes.c_uint * n, ctypes.c_void_p * n
                chunks.append(dict(n=n, pairs=chunk, Ms=I(*[Ms[f] for _, f in chunk]), ns=I(*[ns[f] for _, f in chunk]), seeds=U(),
                                   dirs=PP(*[dptr(scans[f]) for _, f in chunk]), points=PP(*[scans[f]["points"].data_ptr() for _, f in chunk]),
                                   cos=PP(*[scans[f]["cos"].data_ptr() for _, f in chunk]), masks=PP(*[mbase[f] + it * Ms[f] for it, f in chunk]),
                                   out_off=I(*[it * tot + offs[f] for it, f in chunk]), fid=I(*[f for _, f in chunk]), ws=pre["ws"].data_ptr() + 4 * c0 * per))
            pre["chunks"], pre["key"] = chunks, key
        # a call clears the other-parity candidate counters only of the (iteration, frame) slots it includes: slots beyond the previous
        # call's count may hold counters of an older call at either parity - clear both before this call uses them
        n_pairs, last = iters * F, pre.get("pairs_last", 0)
        if n_pairs > last > 0:
            pre["ws"].view(-1, per)[last:n_pairs, :2].zero_()
        pre["pairs_last"] = n_pairs
        for ch in pre["chunks"]:
            sd = ch["seeds"]
            for i, (it, f) in enumerate(ch["pairs"]):
                sd[i] = (int(seeds[it]) * 1000003 + f) & 0xFFFFFFFF
            rc = lib.nl_select_rays_batch_ex(ch["n"], ch["Ms"], ch["ns"], sd, ch["dirs"], ch["points"], ch["cos"], ch["masks"], ch["out_off"], ch["fid"],
                                             pre["d"].data_ptr(), pre["p"].data_ptr(), pre["c"].data_ptr(), pre["f"].data_ptr(), ch["ws"], pre["parity"],
                                             self.adam_state.data_ptr() + 12, sp)
            if rc == 4:
                return False
            L.check(rc, "nl_select_rays_batch_ex")
        pre["parity"] ^= 1
        pre["drawn"] = iters
        return True

    def use_predrawn(self, it):
        """iteration `it` of the call runs on the ray subset predraw() drew for it (run_bound: the descriptor's ray pointers)"""
        pre = self._pre
        tot = pre["tot"]
        d = self._desc
        d.rays_d_sensor, d.points_gt = pre["d"].data_ptr() + 12 * it * tot, pre["p"].data_ptr() + 12 * it * tot
        d.cos_gt, d.frame_id = pre["c"].data_ptr() + 4 * it * tot, pre["f"].data_ptr() + 4 * it * tot
        self.N = tot
        self._pre_it = it

    def set_poses(self, pose6, optimise=None):
        """pose6 [F,6] = (t, w) like se3pose.OptimizablePose.data; optimise[f] = pose is in the optimiser."""
        p = np.asarray(pose6, np.float32).reshape(-1, 6)
        self.F = p.shape[0]
        if self.F > self.F_cap:
            raise L.NerfLoamHipError("too many frames")
        io = np.zeros(self.F_cap * 7, np.int32)
        io[:self.F * 6] = p.reshape(-1).view(np.int32)
        io[self.F_cap * 6:self.F_cap * 6 + self.F] = 1 if optimise is None else np.asarray(optimise, np.int32)
        self._pose_io.copy_(torch.from_numpy(io))
        ops.pose_matrices(self.pose6[:self.F], self.poses12)

    def begin_call(self, m: MapDevice, dec: DecoderDevice = None, emb_state=True):
        """A fresh torch.optim.Adam is created per bundle_adjust_frames / track_frame call
        (render_helpers.py:353,448): reset optimiser state.  emb_state=False (tracking, forward-only queries): the call never
        touches the embedding gradient accumulators / moments, so they are neither allocated nor cleared (160 B per row)."""
        if getattr(self, "_exchange", None) is not None:
            self._exchange.new_call()                            # multi-GPU: re-measure the touched-rows capacity
        self._call_state.zero_()                             # adam state, pose moments, g_pose / pose_grad6 (a previous call may have
        self.graph = None                                    # aborted between backward and the optimiser step): one memset
        E = m.n_rows
        if not emb_state:
            pass
        elif self.g_emb is None or E > self._emb_cap or E < self.g_emb.shape[0]:
            # gradient accumulators + moments for `cap` rows, zero-filled ONCE: a growing map (rows are appended every frame) stays
            # inside the allocation, and only rows a call touches are ever dirtied (and cleaned again through the touched-rows list)
            cap = self._emb_cap = max(E, int(1.25 * E) + 4096) if self.sparse_adam else E
            # (copies: the requested number, halved until the accumulators fit EMB_COPIES_BYTES - 64 B per row and copy: 1.4 GB for 16 copies of the 150-scan map)
            K = self._emb_copies_wanted
            while K > 1 and K * cap * L.NL_C * 4 > EMB_COPIES_BYTES:
                K //= 2
            self.emb_grad_copies = K
            self._emb_state = torch.zeros((K + 1) * cap * L.NL_C, dtype=F32, device=self.dev)     # [g_emb f32 x K copies | emb_m bf16 | emb_v bf16]
            self._touched = ((torch.empty(cap, dtype=I32, device=self.dev), torch.zeros(1, dtype=I32, device=self.dev),
                              torch.zeros((cap + 31) // 32, dtype=I32, device=self.dev), K, cap * L.NL_C) if self.sparse_adam else None)
            self._emb_dirty_dense = False
            self._emb_views(E)
        else:
            if self._touched is None or self._emb_dirty_dense:
                # dense bookkeeping (sparse_adam off, or the previous call exchanged dense gradients across GPUs: rows other ranks
                # touched carry moments without being listed here): clear everything once
                self._emb_state.zero_()
                if self._touched is not None:
                    self._touched[1].zero_(); self._touched[2].zero_()
                self._emb_dirty_dense = False
            else:
                # a fresh Adam (render_helpers.py:353): accumulators / moments / flags of the rows the PREVIOUS call touched - cost
                # proportional to those rows, not to the table
                ops.touched_rows_reset(self._touched, self._emb_state[:self._emb_cap * L.NL_C], self._emb_mv[:self._emb_cap * L.NL_C],
                                       self._emb_mv[self._emb_cap * L.NL_C:])           # (every accumulator copy of the listed rows)
            if E != self.g_emb.shape[0]:
                self._emb_views(E)
        if dec is not None:
            dec.reset_state()

    def _emb_views(self, E):
        cap, K = self._emb_cap, self.emb_grad_copies
        self.g_emb = self._emb_state[:cap * L.NL_C].view(cap, L.NL_C)[:E]            # copy 0 (the only one with emb_grad_copies == 1): g_emb_total() sums the copies
        self._emb_mv = self._emb_state[K * cap * L.NL_C:].view(torch.int16)
        self.emb_m = self._emb_mv[:cap * L.NL_C].view(cap, L.NL_C)[:E]
        self.emb_v = self._emb_mv[cap * L.NL_C:].view(cap, L.NL_C)[:E]

    def g_emb_total(self):
        """the embedding-gradient accumulators as ONE [E,16] fp32 tensor: the sum of the copies (emb_grad_copies > 1: what the optimiser's sweep forms row by row;
        tests and probes that look at the gradient before the optimiser step)"""
        if self.emb_grad_copies == 1:
            return self.g_emb
        cap, K, E = self._emb_cap, self.emb_grad_copies, self.g_emb.shape[0]
        return self._emb_state[:K * cap * L.NL_C].view(K, cap, L.NL_C)[:, :E].sum(0)

    def touched_rows(self):
        """(list, count, flags) for the optimiser's sparse sweep, or None when this call sweeps the table densely: sparse_adam off, or a
        multi-GPU call whose embedding gradients are all-reduced densely (rows only other ranks touched are then not listed here, and the
        next begin_call clears the whole state once).  The scatter records touched rows in either case."""
        if self._touched is None:
            return None
        ex = self._exchange
        if ex is not None and ex.device and not isinstance(ex._rows_cap, int):
            if ex._rows_cap == "dense":
                self._emb_dirty_dense = True
            return None                                           # (undecided: re-evaluated once the first iteration sized the exchange)
        return self._touched

    # ------------------------------------------------------------------ one iteration
    def forward_backward(self, m: MapDevice, dec: DecoderDevice, cfg: IterConfig, train_decoder=True, want_emb_grad=True,
                         want_pose_grad=True, ray_id_base=0, fresh_noise=False):
        """fresh_noise: fold the device-side optimiser step counter into the sampler seed, so every iteration of a call draws
        new jitter like the reference's uniform_() (voxel_helpers.py:298-303) - also under hipGraph replay.  Off = the jitter is a
        pure function of (cfg.noise_seed, ray, step): what the oracle-parity tests need."""
        N = self.N
        c = self.counters
        tm = self._mark
        if self._exchange is not None:                           # multi-GPU: the hooks run the C exchanges on the engine's descriptor
            self._fill_desc(m, dec, cfg, train_decoder=train_decoder, want_emb_grad=want_emb_grad, want_pose_grad=want_pose_grad,
                            update_emb=want_emb_grad, ray_id_base=ray_id_base, fresh_noise=fresh_noise)
        c.zero_()
        self._stats_from_copy = False                            # stage-wise iterations count into (and leave) the live block
        self._desc.counters_clean = 0
        tm("intersect", 0)
        ops.ray_intersect(N, self.rays_d_sensor, self.points_gt, self.cos_gt, self.frame_id, self.poses12, m.blk_hdr, m.blk_ids, m.root_side,
                          m.voxel_size, cfg.max_distance, self.rays_d_world, self.gt_dist, self.hit_idx, self.hit_t0, self.hit_t1,
                          self.hit_count, c, self.ray_of_rank, m.isect_lanes_for(N))
        # hit-ray ranks + compaction + R (and R_GLOBAL: overwritten by the multi-GPU hook) in one launch (two beyond 4096 rays)
        ops.scan_hit_rays(self.hit_count, self.hit_rank, self.ray_of_rank, N, c[L.NLC_R:L.NLC_R + 1], c[L.NLC_R_GLOBAL:L.NLC_R_GLOBAL + 1],
                          self.scan_ws)
        tm("intersect", 1)
        if self.hook_after_intersect is not None:
            self.hook_after_intersect(self)                      # fills NLC_R_GLOBAL / NLC_R_OFFSET / global NLC_HMAX
        seed = 0 if cfg.noise_seed is None else cfg.noise_seed
        use_hash = 0 if cfg.noise_seed is None else 1
        args = (N, self.hit_idx, self.hit_t0, self.hit_t1, self.hit_count, self.hit_rank, self.ray_of_rank, self.cos_gt, self.gt_dist,
                cfg.step_size, cfg.truncation, cfg.max_distance, seed, use_hash, int(cfg.tail_always), ray_id_base,
                self.adam_state if fresh_noise else None, self.row_first, c, self.samp_count)
        tm("sample", 0)
        ops.sample_rays(0, *args, None, self.P_cap, None, None, None, None)
        ops.exclusive_scan(self.samp_count, self.samp_off, N, 0, c[L.NLC_P:L.NLC_P + 1], self.scan_ws)
        if self.hook_after_count is not None:
            self.hook_after_count(self)                          # all-reduce of the loss normalisers
        ops.loss_finalize(c, self.loss_scalars, cfg.fs_weight, cfg.sdf_weight, cfg.truncation, cfg.max_distance, self.P_cap)
        ops.sample_rays(1, *args, self.samp_off, self.P_cap, self.s_vox, self.s_depth, self.s_dist, self.s_ray)
        tm("sample", 1)
        tm("gather", 0)
        ops.gather_trilinear(self.loss_scalars, self.s_vox, self.s_depth, self.s_ray, self.rays_d_world, self.frame_id, self.poses12,
                             self.F, m.centres, m.vertex_rows, m.emb, m.voxel_size, self.X, self.field_blocks)
        tm("gather", 1)
        tm("decoder", 0)
        modes = self.kernel_modes                            # decoder workgroup layout by ray count where the engine leaves it open (what nl_iteration does)
        if not (modes >> 16) & 3:
            modes |= int(L.lib().nl_decoder_layout_for(N)) << 16
        ops.decoder_fwd_bwd(self.loss_scalars, self.X, dec.params, dec.W2T, self.s_ray, self.s_depth, self.cos_gt, self.gt_dist,
                            self.sdf, self.dsdf, self.dX, self.partials, self.relu2_mask, self.n_slabs, int(train_decoder), c,
                            modes)
        tm("decoder", 1)
        if train_decoder:
            tm("wgrad2", 0)
            ops.decoder_wgrad2(self.loss_scalars, self.X, dec.params, self.dsdf, self.relu2_mask, self.partials, self.n_slabs, modes)
            tm("wgrad2", 1)
            tm("reduce", 0)
            ops.decoder_reduce(self.partials, self.n_slabs, dec.params, dec.grad, modes)
            tm("reduce", 1)
        tm("scatter", 0)
        ops.trilinear_bwd(self.loss_scalars, self.s_vox, self.s_depth, self.s_ray, self.rays_d_world, self.rays_d_sensor, self.frame_id,
                          self.poses12, self.F, m.centres, m.vertex_rows, m.emb, m.voxel_size, self.dX,
                          self.g_emb if want_emb_grad else None, self.g_pose if want_pose_grad else None, 2 * self.field_blocks,
                          self._touched if want_emb_grad else None)
        tm("scatter", 1)
        if self.hook_after_backward is not None:
            self.hook_after_backward(self, dec, train_decoder, want_emb_grad, want_pose_grad)

    def _mark(self, stage, which):
        """bench / probes: self.timers = {stage: (start, end)} torch.cuda.Event pairs on the launch stream; stages: intersect,
        sample, gather, decoder, wgrad2, reduce, scatter, optim"""
        t = self.timers
        if t is not None and stage in t:
            t[stage][which].record()

    def forward_only(self, m: MapDevice, dec: DecoderDevice, cfg: IterConfig, ray_id_base=0):
        """render_rays without gradients (render_helpers.py:190-318): intersect, sample, gather, decoder forward."""
        N = self.N
        c = self.counters
        c.zero_()
        ops.ray_intersect(N, self.rays_d_sensor, self.points_gt, self.cos_gt, self.frame_id, self.poses12, m.blk_hdr, m.blk_ids, m.root_side,
                          m.voxel_size, cfg.max_distance, self.rays_d_world, self.gt_dist, self.hit_idx, self.hit_t0, self.hit_t1,
                          self.hit_count, c, self.ray_of_rank, m.isect_lanes_for(N))
        ops.scan_hit_rays(self.hit_count, self.hit_rank, self.ray_of_rank, N, c[L.NLC_R:L.NLC_R + 1], c[L.NLC_R_GLOBAL:L.NLC_R_GLOBAL + 1],
                          self.scan_ws)
        seed = 0 if cfg.noise_seed is None else cfg.noise_seed
        args = (N, self.hit_idx, self.hit_t0, self.hit_t1, self.hit_count, self.hit_rank, self.ray_of_rank, self.cos_gt, self.gt_dist,
                cfg.step_size, cfg.truncation, cfg.max_distance, seed, 0 if cfg.noise_seed is None else 1, int(cfg.tail_always),
                ray_id_base, None, None, c, self.samp_count)
        ops.sample_rays(0, *args, None, self.P_cap, None, None, None, None)
        ops.exclusive_scan(self.samp_count, self.samp_off, N, 0, c[L.NLC_P:L.NLC_P + 1], self.scan_ws)
        ops.loss_finalize(c, self.loss_scalars, cfg.fs_weight, cfg.sdf_weight, cfg.truncation, cfg.max_distance, self.P_cap)
        ops.sample_rays(1, *args, self.samp_off, self.P_cap, self.s_vox, self.s_depth, self.s_dist, self.s_ray)
        ops.gather_trilinear(self.loss_scalars, self.s_vox, self.s_depth, self.s_ray, self.rays_d_world, self.frame_id, self.poses12,
                             self.F, m.centres, m.vertex_rows, m.emb, m.voxel_size, self.X, self.field_blocks)
        P = int(c[L.NLC_P].item())                                        # host sync: a forward-only query returns data anyway
        ops.decoder_forward(self.X, dec.params, dec.W2T, min(P, self.P_cap), self.sdf, self.n_slabs, self.kernel_modes)
        return P

    def optimiser_step(self, m: MapDevice, dec: DecoderDevice, cfg: IterConfig, update_emb=True, update_decoder=True, update_pose=True,
                       lr_pose=None, skip_mode=0):
        """optim.step() of render_helpers.py:421-423 / :508-510 on the device-resident parameters.  skip_mode 1 / 2: the kernel itself
        leaves an unusable iteration (no hit ray, sampler guard, sample overflow) without a step - see call_status()."""
        self._mark("optim", 0)
        ops.optimiser_step(self.adam_state, cfg.lr_emb, cfg.lr_dec, cfg.lr_pose if lr_pose is None else lr_pose,
                           (m.emb, self.g_emb, self.emb_m, self.emb_v) if update_emb else None,
                           (dec.params, dec.grad, dec.m, dec.v, dec.W2T) if update_decoder else (None, None, None, None, dec.W2T),   # (the workspace always: range status)
                           (self.pose6[:self.F], self.g_pose, self.pose_m, self.pose_v, self.pose_enable, self.pose_grad6, self.poses12,
                            update_pose), self.counters if skip_mode else None, skip_mode, self.touched_rows() if update_emb else None)
        self._mark("optim", 1)

    def call_status(self):
        """(steps taken, steps skipped as unusable, overflow seen) since begin_call - ONE small read-back per call instead of one
        per iteration"""
        st = self.adam_state[:4].cpu().numpy()
        self.saturated = bool(st[3] & 2)                    # the decoder's range status, latched by the optimiser (DecoderDevice.range_status)
        return int(st[0]), int(st[2]), bool(st[3] & 1)

    def call_status_and_poses(self):
        """call_status() + the frames' current pose6, in ONE device-to-host copy"""
        o = self.F_cap * 24
        both = torch.cat([self._call_state[o:o + 4], self.pose6[:self.F].reshape(-1).view(I32)]).cpu()
        st = both[:4].numpy()
        self.saturated = bool(st[3] & 2)
        return (int(st[0]), int(st[2]), bool(st[3] & 1)), both[4:].view(F32).view(self.F, 6)

    # ------------------------------------------------------------------ one C call per iteration
    def bind(self, m: MapDevice, dec: DecoderDevice, cfg: IterConfig, train_decoder=True, want_emb_grad=True, want_pose_grad=True,
             update_emb=True, update_decoder=True, update_pose=True, lr_pose=None, skip_mode=0, fresh_noise=False, ray_id_base=0):
        """Fill the engine's NlIterDesc (include/nerfloam_hip.h) for a run of iterations on (m, dec, cfg): every device pointer and
        hyper-parameter once, so that run_bound() is ONE ctypes call per iteration instead of ~15 calls with ~250 marshalled
        arguments.  Call again when the map, the decoder, the configuration or the flags change (tensors are looked up here, not in
        the loop).  Not used when multi-GPU hooks or stage timers are installed (forward_backward / optimiser_step then)."""
        if self._exchange is None and any(h is not None for h in (self.hook_after_intersect, self.hook_after_count, self.hook_after_backward)):
            raise L.NerfLoamHipError("bind() / run_bound(): stage hooks are installed on this engine, and they run only on the stage-wise "
                                     "path forward_backward + optimiser_step (a ray-sharded engine - dist.RayShardedExchange - is fine: "
                                     "nl_iteration issues its exchanges itself through the descriptor's communicator)")
        if self._exchange is not None and not self._exchange.device:
            raise L.NerfLoamHipError("bind() / run_bound() need device tensors")
        self._fill_desc(m, dec, cfg, train_decoder, want_emb_grad, want_pose_grad, update_emb, update_decoder, update_pose, lr_pose, skip_mode,
                        fresh_noise, ray_id_base)
        self._bound = (m, dec)                                       # keeps the tensors the descriptor points at alive

    def _fill_desc(self, m, dec, cfg, train_decoder=True, want_emb_grad=True, want_pose_grad=True, update_emb=True, update_decoder=True,
                   update_pose=True, lr_pose=None, skip_mode=0, fresh_noise=False, ray_id_base=0):
        d = self._desc
        d.struct_size = ctypes.sizeof(L.NlIterDesc)
        pt = lambda t: None if t is None else t.data_ptr()          # noqa: E731
        for name in ("rays_d_sensor", "points_gt", "cos_gt", "frame_id", "pose6", "poses12", "pose_m", "pose_v", "pose_enable", "g_pose", "pose_grad6",
                     "rays_d_world", "gt_dist", "hit_idx", "hit_t0", "hit_t1", "hit_count", "hit_rank", "ray_of_rank", "samp_count", "samp_off",
                     "scan_ws", "s_vox", "s_depth", "s_dist", "s_ray", "X", "dX", "sdf", "dsdf", "relu2_mask", "counters", "loss_scalars",
                     "adam_state", "partials", "g_emb", "emb_m", "emb_v"):
            setattr(d, name, pt(getattr(self, name)))
        d.blk_hdr, d.blk_ids, d.root_side, d.voxel_size = pt(m.blk_hdr), pt(m.blk_ids), int(m.root_side), float(m.voxel_size)
        d.isect_lanes = m.isect_lanes_for(self.N)
        d.centres, d.vertex_rows, d.emb, d.n_emb_elems = pt(m.centres), pt(m.vertex_rows), pt(m.emb), int(m.emb.numel())
        d.dec_params, d.dec_ws, d.dec_grad, d.dec_m, d.dec_v = pt(dec.params), pt(dec.W2T), pt(dec.grad), pt(dec.m), pt(dec.v)
        d.P_cap, d.n_slabs, d.field_blocks = self.P_cap, self.n_slabs, self.field_blocks
        d.step_size, d.max_distance, d.truncation = cfg.step_size, cfg.max_distance, cfg.truncation
        d.sdf_weight, d.fs_weight = cfg.sdf_weight, cfg.fs_weight
        d.lr_emb, d.lr_dec, d.lr_pose = cfg.lr_emb, cfg.lr_dec, cfg.lr_pose if lr_pose is None else lr_pose
        d.noise_seed = (0 if cfg.noise_seed is None else int(cfg.noise_seed)) & 0xFFFFFFFF
        d.use_hash_noise, d.tail_always = int(cfg.noise_seed is not None), int(cfg.tail_always)
        d.ray_id_base, d.fresh_noise = int(ray_id_base), int(fresh_noise)
        d.train_decoder, d.want_emb_grad, d.want_pose_grad = int(train_decoder), int(want_emb_grad), int(want_pose_grad)
        d.update_emb, d.update_decoder, d.update_pose, d.skip_mode = int(update_emb), int(update_decoder), int(update_pose), int(skip_mode)
        if want_emb_grad or update_emb:
            assert self.g_emb is not None, "begin_call(emb_state=True) first"
        d.counters_copy, d.counters_clean = pt(self.counters_copy), 0
        d.sample_state = pt(self.sample_state)
        d.kernel_modes = self.kernel_modes
        d.N, d.F = self.N, self.F
        if self._exchange is not None:
            self._exchange.prepare(m, dec, bool(want_emb_grad))
        self._desc_touched()

    def _desc_touched(self):
        d, t = self._desc, self._touched
        d.touched_list, d.touched_count, d.touched_flags = (None, None, None) if t is None else (t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())
        d.touched_copies, d.touched_copy_stride = (1, 0) if t is None else (int(t[3]), int(t[4]))
        d.sparse_sweep = int(self.touched_rows() is not None)

    def run_bound(self, stages=3):
        """one iteration of the bound configuration: stages bit 0 = forward + backward, bit 1 = optimiser step"""
        d = self._desc
        if d.N != self.N:                                           # (the lane choice follows the ray count: 32 / 16 / 8 lanes per ray)
            d.isect_lanes = self._bound[0].isect_lanes_for(self.N) if self._bound is not None else 0
        d.N, d.F = self.N, self.F
        stages = int(stages)
        # bench / probes: self.timers = {"decoder": (e0, e1), "wgrad2": (e1, e2)} - torch.cuda.Event objects that have been recorded once (their
        # handles exist): nl_iteration records them around the decoder kernel and behind dW2 (NlIterDesc.ev_decoder_begin ...)
        t = self.timers
        if t is not None and "decoder" in t:
            d.ev_decoder_begin, d.ev_decoder_end = t["decoder"][0].cuda_event, t["decoder"][1].cuda_event
            d.ev_wgrad2_end = t["wgrad2"][1].cuda_event if "wgrad2" in t else None
            self._desc_timed = True
        elif getattr(self, "_desc_timed", False):
            d.ev_decoder_begin = d.ev_decoder_end = d.ev_wgrad2_end = None
            self._desc_timed = False
        ex = self._exchange
        if ex is None:
            L.check(L.lib().nl_iteration(ctypes.byref(d), stages, L.stream_ptr()), "nl_iteration")
        elif not (stages & 1):
            ex._check(L.lib().nl_iteration(ctypes.byref(d), stages, L.stream_ptr()), "nl_iteration")
        elif ex.rows_undecided():
            # first iteration of a call on a touched-rows map: forward + backward, ONE host read sizes the row exchange of the whole
            # call (collective sizes must be known on the host), then the gradient exchange and the optimiser step
            ex._check(L.lib().nl_iteration(ctypes.byref(d), 1, L.stream_ptr()), "nl_iteration")
            ex.decide_rows()
            self._desc_touched()                                 # (dense exchange -> dense optimiser sweep for this call)
            ex._check(L.lib().nl_iteration(ctypes.byref(d), 4 | (stages & 2), L.stream_ptr()), "nl_iteration")
            stages &= ~2                                         # (two calls: the counter block was not handed over - see below)
        else:
            ex._check(L.lib().nl_iteration(ctypes.byref(d), stages | 4, L.stream_ptr()), "nl_iteration")     # one C call, exchanges included
        # a whole iteration ends with its counter block handed to counters_copy and the live block cleared for the next one
        # (no memset launch then); a forward-only call leaves the block in place
        whole = (stages & 3) == 3
        d.counters_clean = int(whole)
        if stages & 1:
            self._stats_from_copy = whole

    # ------------------------------------------------------------------ hipGraph
    def capture_iteration(self, m: MapDevice, dec: DecoderDevice, cfg: IterConfig, **flags):
        """Capture forward_backward + optimiser_step into a hipGraph (torch.cuda.CUDAGraph).  Possible because the launch
        sequence is fixed, every data-dependent size lives in device memory and the optimiser step counter is on the
        device.  Ray buffers / poses may be rewritten between replays (set_rays / set_poses write in place)."""
        fb = {k: flags[k] for k in ("train_decoder", "want_emb_grad", "want_pose_grad", "ray_id_base", "fresh_noise") if k in flags}
        op = {k: flags[k] for k in ("update_emb", "update_decoder", "update_pose", "lr_pose", "skip_mode") if k in flags}
        state0 = self.adam_state.clone()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):                       # warm-up outside capture (allocator, lazy module load)
            self.forward_backward(m, dec, cfg, **fb)
            if self.g_emb is not None:
                # EVERY accumulator copy (emb_grad_copies > 1: g_emb is copy 0 only; the warm-up's waves added into all of them, and the rows they touched
                # stay listed - the first replay's sweep would otherwise add the stale copies 1..K-1 to its gradient)
                self._emb_state[:self.emb_grad_copies * self._emb_cap * L.NL_C].zero_()
            self.g_pose.zero_()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.forward_backward(m, dec, cfg, **fb)
            self.optimiser_step(m, dec, cfg, **op)
        self.adam_state.copy_(state0)                       # capture does not execute, but keep the counter explicit
        self.graph = g
        return g

    def replay(self):
        self.graph.replay()

    # ------------------------------------------------------------------ read-back (tests, API parity)
    def stats(self):
        """Host copy of the counter block (synchronises)."""
        c = (self.counters_copy if self._stats_from_copy else self.counters).cpu().numpy()
        ints = c[:L.NL_CNT_INTS]
        dbl = c[L.NL_CNT_INTS:].view(np.float64)
        ls = self.loss_scalars.cpu().numpy()
        lf = ls.view(np.float32)
        return dict(R=int(ints[L.NLC_R]), H=int(ints[L.NLC_HMAX]), S=int(ints[L.NLC_SMAX]), P=int(ints[L.NLC_P]),
                    overflow=int(ints[L.NLC_OVERFLOW]), guard=int(ints[L.NLC_GUARD]),
                    w_fs=float(lf[0]), w_sdf=float(lf[1]), ints=ints.copy(), dbl=dbl.copy())

    def loss_value(self, cfg: IterConfig):
        """Scalar loss of criterion.py:43-56 from the device sums (value only; gradients never need it)."""
        st = self.stats()
        ints, dbl = st["ints"], st["dbl"]
        n = float(int(ints[L.NLC_R_GLOBAL])) * float(st["S"])       # global hit-ray count (== R on one GPU; under ray sharding the
        if n == 0:                                                  # residual sums must have been all-reduced: dist.reduce_loss_sums)
            return None
        inv_fs = st["S"] * int(ints[L.NLC_INV_FS_RAYS]) - int(ints[L.NLC_INV_FS_CNT])
        fs = (dbl[L.NLD_FS_SQ] + inv_fs) / n * st["w_fs"]
        sd = (dbl[L.NLD_SDF_SQ] + st["S"] * dbl[L.NLD_INV_D2] - dbl[L.NLD_INV_D2CNT]) / n * st["w_sdf"]
        return dict(loss=cfg.fs_weight * fs + cfg.sdf_weight * sd, fs_loss=fs, sdf_loss=sd, **{k: st[k] for k in ("R", "S", "P", "H")})

    def export_render_device(self):
        """The dict the reference's render_rays returns (render_helpers.py:311-318), rebuilt from the packed device buffers by
        nl_unpack_samples and LEFT ON THE DEVICE: z_vals[R,S], sdf[R,S] (ones where invalid), valid_mask[R,S] bool, ray_mask[N] bool."""
        st = self.stats()
        R, S, N = st["R"], st["S"], self.N
        if R == 0 or S == 0:
            return None
        out_sdf = torch.ones(R, S, dtype=F32, device=self.dev)
        out_z = torch.full((R, S), 80.0, dtype=F32, device=self.dev)
        out_valid = torch.zeros(R, S, dtype=torch.uint8, device=self.dev)
        ops.unpack_samples(self.loss_scalars, self.s_ray, self.samp_off, self.hit_rank, self.sdf, self.s_depth, S, out_sdf, out_z, out_valid)
        return dict(z_vals=out_z, sdf=out_sdf, valid_mask=out_valid.view(torch.bool), ray_mask=self.hit_count[:N] > 0, stats=st)

    def export_render(self):
        """export_render_device() as numpy arrays (tests, probes)"""
        r = self.export_render_device()
        if r is None:
            return None
        return dict(z_vals=r["z_vals"].cpu().numpy(), sdf=r["sdf"].cpu().numpy(), valid_mask=r["valid_mask"].cpu().numpy(),
                    ray_mask=r["ray_mask"].cpu().numpy(), stats=r["stats"])
