"""ctypes binding of libnerfloam_hip.so (include/nerfloam_hip.h).  Fails loudly: there is no CPU or
PyTorch fallback for the hot path - if the HIP library is missing or no GPU is visible, the product
raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NL_LIB_PATH: another build of the same library (same-box A/B builds in ab_libs/, the AddressSanitizer build of scripts/asan_build.py)
LIB_PATH = os.environ.get("NL_LIB_PATH") or os.path.join(_HERE, "libnerfloam_hip.so")

NL_MAX_HITS = 20
NL_CNT_INTS = 16
NL_CNT_DOUBLES = 4
NL_CNT_BYTES = NL_CNT_INTS * 4 + NL_CNT_DOUBLES * 8
NL_LOSS_SCALARS_BYTES = 48
NL_ADAM_STATE_BYTES = 112
NL_DEC_PARAMS = 70401
NL_ABI_VERSION = 6
NL_DEC_WS_RANGE_STATUS = 401408 + 4    # float index of the range block's sticky status word in the decoder weight workspace (csrc/nl_common.h NLR_STATUS)
NL_SAT_X, NL_SAT_H1, NL_SAT_Q, NL_SAT_PLANES = 1, 2, 4, 8
NL_DEC_WS_FLOATS = 401424        # decoder weight workspace: W2^T fp32 + 2 x 3 bf16 + 2 x 2 fp16 operand planes of W2 + 2 x 2 fp16 planes of W1 (include/nerfloam_hip.h; checked against nl_dec_ws_floats() at load)
NL_SEL_BATCH_WS_INTS_PER_FRAME = 8 + 4 * 128 + 2 * 4096     # NL_SELECT_BATCH_WS_INTS(1)
NL_SEL_MAX_FRAMES = 8
NL_MAX_FRAMES = 32               # frames (poses) one field-kernel launch takes (csrc/nl_field.hip)
NL_C = 16
NL_W = 256
OFF_W1, OFF_B1 = 0, 256 * 16
OFF_W2 = OFF_B1 + 256
OFF_B2 = OFF_W2 + 256 * 256
OFF_W3 = OFF_B2 + 256
OFF_B3 = OFF_W3 + 256

# counter indices (nl_common.h)
(NLC_R, NLC_HMAX, NLC_SMAX, NLC_P, NLC_NFS, NLC_NSDF, NLC_INV_FS_RAYS, NLC_INV_FS_CNT, NLC_INV_SDF_RAYS,
 NLC_INV_SDF_CNT, NLC_OVERFLOW, NLC_GUARD, NLC_R_OFFSET, NLC_R_GLOBAL, NLC_ISECT_OVF, NLC_TICKET) = range(16)
NLD_FS_SQ, NLD_SDF_SQ, NLD_INV_D2, NLD_INV_D2CNT = range(4)

_ERR = {1: "invalid argument", 2: "kernel launch failed", 3: "no HIP device", 4: "capacity exceeded"}


class NerfLoamHipError(RuntimeError):
    pass


def _iter_desc_fields():
    P_, I_, F_, D_, LL_, U_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_longlong, ctypes.c_uint
    f = [("struct_size", I_), ("N", I_), ("F", I_)]
    f += [(n, P_) for n in ("rays_d_sensor", "points_gt", "cos_gt", "frame_id")]
    f += [(n, P_) for n in ("pose6", "poses12", "pose_m", "pose_v", "pose_enable", "g_pose", "pose_grad6")]
    f += [("blk_hdr", P_), ("blk_ids", P_), ("root_side", I_), ("voxel_size", F_)]
    f += [("centres", P_), ("vertex_rows", P_), ("emb", P_), ("n_emb_elems", LL_)]
    f += [(n, P_) for n in ("rays_d_world", "gt_dist", "hit_idx", "hit_t0", "hit_t1", "hit_count", "hit_rank", "ray_of_rank", "samp_count",
                            "samp_off", "scan_ws")]
    f += [("P_cap", I_)] + [(n, P_) for n in ("s_vox", "s_depth", "s_dist", "s_ray", "X", "dX", "sdf", "dsdf", "relu2_mask")]
    f += [("counters", P_), ("loss_scalars", P_), ("adam_state", P_)]
    f += [(n, P_) for n in ("dec_params", "dec_ws", "dec_grad", "dec_m", "dec_v", "partials")] + [("n_slabs", I_), ("field_blocks", I_)]
    f += [("g_emb", P_), ("emb_m", P_), ("emb_v", P_)]
    f += [(n, F_) for n in ("step_size", "max_distance", "truncation", "sdf_weight", "fs_weight")]
    f += [("lr_emb", D_), ("lr_dec", D_), ("lr_pose", D_)]
    f += [("noise_seed", U_)] + [(n, I_) for n in ("use_hash_noise", "tail_always", "ray_id_base", "fresh_noise")]
    f += [(n, I_) for n in ("train_decoder", "want_emb_grad", "want_pose_grad", "update_emb", "update_decoder", "update_pose", "skip_mode")]
    f += [("counters_copy", P_), ("counters_clean", I_)]
    f += [("sample_state", P_)]
    f += [("kernel_modes", I_)]
    # ray-sharded multi-GPU iteration (NULL comm: one GPU)
    f += [("comm", P_), ("xg_send", P_), ("xg_recv", P_), ("xg_stride", I_), ("row_first", P_), ("row_first_entries", I_)]
    f += [("rows_mode", I_), ("rows_bitmap", P_), ("rows_prefix", P_), ("rows_total", P_), ("rows_ws", P_), ("rows_buf", P_), ("rows_cap", I_), ("rows_words", I_)]
    f += [("touched_list", P_), ("touched_count", P_), ("touched_flags", P_), ("sparse_sweep", I_)]
    f += [("touched_copies", I_), ("touched_copy_stride", LL_)]
    f += [("x1_send", P_), ("x1_recv", P_), ("x1_stride_bytes", I_), ("x1_rays", I_)]
    f += [("comm_stream", P_), ("ev_fork", P_), ("ev_join", P_)]
    f += [("isect_lanes", I_)]
    f += [("ev_decoder_begin", P_), ("ev_decoder_end", P_), ("ev_wgrad2_end", P_)]
    return f


class NlIterDesc(ctypes.Structure):
    """ctypes mirror of NlIterDesc (include/nerfloam_hip.h): field order and types must match (tests/test_c_abi_exports.py)"""
    _fields_ = _iter_desc_fields()


class NlTouchedRows(ctypes.Structure):
    """rows of the embedding table touched since the optimiser of a call was created (include/nerfloam_hip.h)"""
    _fields_ = [("struct_size", ctypes.c_int), ("list", ctypes.c_void_p), ("count", ctypes.c_void_p), ("flags", ctypes.c_void_p), ("copies", ctypes.c_int), ("copy_stride", ctypes.c_longlong)]


# communicator of the ray-sharded iteration (NlComm, include/nerfloam_hip.h)
NL_COMM_F32, NL_COMM_F64, NL_COMM_I32 = 0, 1, 2
NL_COMM_ALL_GATHER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p)
NL_COMM_ALL_REDUCE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p)
NL_COMM_GROUP = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)


class NlComm(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int), ("rank", ctypes.c_int), ("ctx", ctypes.c_void_p), ("all_gather", NL_COMM_ALL_GATHER),
                ("all_reduce_sum", NL_COMM_ALL_REDUCE), ("group_begin", NL_COMM_GROUP), ("group_end", NL_COMM_GROUP)]


_lib = None

_P, _I, _F, _D, _LL, _U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_longlong, ctypes.c_uint

_SIGS = {
    "nl_version": ([], _I),
    "nl_device_count": ([], _I),
    "nl_decoder_grid_hint": ([], _I),
    "nl_svo_intersect": ([_P] * 4 + [_I, _I, _I, _F, _I] + [_P] * 4, _I),
    "nl_inverse_cdf_sampling": ([_P] * 6 + [_I] * 4 + [_F] + [_P] * 4, _I),
    "nl_ray_intersect": ([_I] + [_P] * 7 + [_I, _F, _F] + [_P] * 9, _I),
    "nl_ray_intersect_lanes": ([_I] + [_P] * 7 + [_I, _F, _F] + [_P] * 8 + [_I, _P], _I),
    "nl_exclusive_scan_i32": ([_P, _P, _I, _I, _P, _P, _P], _I),
    "nl_compact_hit_rays": ([_I, _P, _P, _P, _P], _I),
    "nl_scan_hit_rays": ([_P, _P, _P, _I, _P, _P, _P, _P], _I),
    "nl_sample_rays": ([_I, _I] + [_P] * 8 + [_F, _F, _F, _U, _I, _I, _I] + [_P, _P, _P, _P, _P, _I] + [_P] * 5, _I),
    "nl_dist_merge_finalize": ([_P, _I, _I, _P, _P, _F, _F, _F, _F, _I, _P], _I),
    "nl_dist_x1_pack": ([_P, _P, _I, _I, _P, _P], _I),
    "nl_dist_x1_merge": ([_P, _I, _I, _I, _I, _P, _P, _I, _P], _I),
    "nl_exchange_emb_pose": ([_P, _P], _I),
    "nl_exchange_decoder": ([_P, _P], _I),
    "nl_overlap_create": ([_P, _P, _P], _I),
    "nl_overlap_destroy": ([_P, _P, _P], _I),
    "nl_loss_finalize": ([_P, _P, _F, _F, _F, _F, _I, _P], _I),
    "nl_criterion_forward": ([_I, _I] + [_P] * 6 + [_F] * 4 + [_P] * 3, _I),
    "nl_criterion_backward": ([_I, _I] + [_P] * 6 + [_F] * 4 + [_P] * 4, _I),
    "nl_scan_samples_finalize": ([_P, _P, _I, _P, _P, _F, _F, _F, _F, _I, _P, _P], _I),
    "nl_sample_rays_fused": ([_I] + [_P] * 8 + [_F, _F, _F, _U, _I, _I, _I] + [_P] * 4 + [_I] + [_P] * 5 + [_F, _F, _P, _P, _P], _I),
    "nl_ray_intersect_scan": ([_I] + [_P] * 7 + [_I, _F, _F] + [_P] * 13, _I),
    "nl_ray_intersect_scan_lanes": ([_I] + [_P] * 7 + [_I, _F, _F] + [_P] * 12 + [_I, _P], _I),
    "nl_isect_lanes_for": ([_I, _I], _I),
    "nl_ray_intersect_scan_x1": ([_I] + [_P] * 7 + [_I, _F, _F] + [_P] * 12 + [_I, _P, _I, _P], _I),
    "nl_gather_trilinear": ([_P] * 7 + [_I] + [_P] * 3 + [_F, _P, _I, _P], _I),
    "nl_gather_points": ([_I] + [_P] * 5 + [_F, _P, _P], _I),
    "nl_gather_grid": ([_I, _I, _I, _P, _P, _P, _P, _F, _P, _P], _I),
    "nl_mc_count": ([_P, _I, _I, _P, _P, _P], _I),
    "nl_mc_emit": ([_P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P], _I),
    "nl_decoder_fwd_bwd": ([_P] * 13 + [_I, _I, _P, _P], _I),
    "nl_decoder_wgrad2": ([_P] * 6 + [_I, _P], _I),
    "nl_decoder_forward": ([_P, _P, _P, _I, _P, _I, _P], _I),
    "nl_decoder_fwd_bwd_m": ([_P] * 13 + [_I, _I, _P, _I, _P], _I),
    "nl_decoder_wgrad2_m": ([_P] * 6 + [_I, _I, _P], _I),
    "nl_decoder_forward_m": ([_P, _P, _P, _I, _P, _I, _I, _P], _I),
    "nl_decoder_reduce_m": ([_P, _I, _P, _P, _I, _P], _I),
    "nl_field_set_debug_buffer": ([_P], _I),
    "nl_field_set_one_round": ([_I], _I),
    "nl_field_set_probes": ([_I], _I),
    "nl_field_set_midspan_flush": ([_I], _I),
    "nl_geometry_set_debug_buffer": ([_P], _I),
    "nl_geometry_set_lanes_per_ray": ([_I], _I),
    "nl_geometry_set_intersect_prune": ([_I], _I),
    "nl_geometry_set_scan_single": ([_I], _I),
    "nl_geometry_set_sampler_mode": ([_I], _I),
    "nl_dist_merge_counters": ([_P, _I, _I, _I, _P, _P], _I),
    "nl_unit_dirs": ([_I, _P, _P, _P, _P], _I),
    "nl_select_rays": ([_I, _I, ctypes.c_uint, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P], _I),
    "nl_select_rays_batch": ([_I] + [_P] * 13 + [_I, _P, _P], _I),
    "nl_select_rays_batch_ex": ([_I] + [_P] * 14 + [_I, _P, _P], _I),
    "nl_decoder_set_gemm_mode": ([_I], _I),
    "nl_decoder_get_gemm_mode": ([], _I),
    "nl_decoder_set_wgrad2_mode": ([_I], _I),
    "nl_decoder_get_wgrad2_mode": ([], _I),
    "nl_decoder_set_layout": ([_I], _I),
    "nl_decoder_get_layout": ([], _I),
    "nl_decoder_layout_for": ([_I], _I),
    "nl_dec_ws_floats": ([], _I),
    "nl_decoder_range_status": ([_P, _P, _I, _P], _I),
    "nl_abi_version": ([], _I),
    "nl_reduce_partials": ([_P, _I, _I, _P, _P], _I),
    "nl_decoder_reduce": ([_P, _I, _P, _P, _P], _I),
    "nl_decoder_transpose_w2": ([_P, _P, _P], _I),
    "nl_trilinear_bwd": ([_P] * 8 + [_I] + [_P] * 3 + [_F] + [_P] * 3 + [_I, _P], _I),
    "nl_unpack_samples": ([_P] * 6 + [_I] + [_P] * 4, _I),
    "nl_adam_prepare": ([_P, _D, _D, _D, _P], _I),
    "nl_adam_embeddings": ([_P, _P, _P, _P, _LL, _P, _P], _I),
    "nl_embedding_grad_bf16": ([_P, _P, _LL, _P], _I),
    "nl_adam_f32": ([_P, _P, _P, _P, _I, _P, _I, _P], _I),
    "nl_pose_matrices": ([_P, _P, _I, _P], _I),
    "nl_pose_step": ([_P] * 7 + [_I, _P, _I, _P], _I),
    "nl_optimiser_step": ([_P, _D, _D, _D] + [_P] * 4 + [_LL] + [_P] * 5 + [_P] * 7 + [_I, _I, _P, _I, _P], _I),
    "nl_optimiser_step_ex": ([_P, _D, _D, _D] + [_P] * 4 + [_LL] + [_P] * 5 + [_P] * 7 + [_I, _I, _P, _I, _P, _P, _P], _I),
    "nl_iteration": ([_P, _I, _P], _I),
    "nl_trilinear_bwd_t": ([_P] * 8 + [_I] + [_P] * 3 + [_F] + [_P] * 3 + [_I, _P, _P], _I),
    "nl_touched_rows_reset": ([_P, _P, _P, _P, _P], _I),
    "nl_optimiser_step_t": ([_P, _D, _D, _D] + [_P] * 4 + [_LL] + [_P] * 5 + [_P] * 7 + [_I, _I, _P, _I, _P, _P, _P, _P], _I),
    "nl_dist_rows_move_t": ([_I, _P, _P, _I, _P, _P, _I, _P, _P, _P], _I),
    "nl_comm_init_rccl": ([_P, _P, _I, _I], _I),
    "nl_exchange_after_intersect": ([_P, _P], _I),
    "nl_exchange_after_intersect_packed": ([_P, _P], _I),
    "nl_exchange_after_sampling": ([_P, _P], _I),
    "nl_exchange_gradients": ([_P, _P], _I),
    "nl_dist_merge_counters_strided": ([_P, _I, _I, _I, _I, _P, _P], _I),
    "nl_dist_rows_union_prefix": ([_P, _I, _I, _I, _P, _I, _P, _P, _P, _P], _I),
    "nl_dist_mark_rows": ([_I, _P, _P, _P, _P, _P], _I),
    "nl_dist_rows_prefix": ([_P, _I, _P, _P, _P, _P], _I),
    "nl_dist_rows_move": ([_I, _P, _P, _I, _P, _P, _I, _P, _P], _I),
    "nl_octree_create": ([_LL], _P),
    "nl_octree_destroy": ([_P], None),
    "nl_octree_insert": ([_P, _P, _LL], _I),
    "nl_octree_count_nodes": ([_P], _LL),
    "nl_octree_count_leaf_nodes": ([_P], _LL),
    "nl_octree_has_voxel": ([_P, _I, _I, _I], _I),
    "nl_octree_voxels_dfs": ([_P, _P], _I),
    "nl_octree_leaf_voxels": ([_P, _P], _LL),
    "nl_octree_try_insert": ([_P, _P, _LL], ctypes.c_double),
    "nl_octree_export": ([_P, _P, _P, _P], _I),
    "nl_octree_export_device_layout": ([_P, _F, _P, _P, _P], _I),
    "nl_octree_delta_count": ([_P], _LL),
    "nl_octree_export_delta": ([_P, _F, _P, _P, _P, _P], _I),
    "nl_octree_pack_blocks": ([_P, _P, _P, _LL], _LL),
    "nl_mfma_selftest": ([_P] * 7, _I),
    "nl_decoder_set_debug_buffer": ([_P], _I),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def lib():
    """Load libnerfloam_hip.so (built by nerf_loam_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        # torch first: its wheel bundles its own libamdhip64; loading it before our library makes both
        # use ONE HIP runtime in the process (the other order leaves torch without devices)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise NerfLoamHipError(f"{LIB_PATH} not found - run `python -m nerf_loam_amd.build` (needs hipcc); "
                                   "the SDF hot path has no CPU/PyTorch fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        # the caller-allocated workspaces and the descriptor structs of this binding are sized from constants compiled in HERE: a library
        # of another revision would overrun them - refuse it at load
        if L.nl_abi_version() != NL_ABI_VERSION or L.nl_dec_ws_floats() != NL_DEC_WS_FLOATS:
            raise NerfLoamHipError(f"{LIB_PATH}: ABI version {L.nl_abi_version()} / decoder workspace {L.nl_dec_ws_floats()} floats, this binding "
                                   f"expects {NL_ABI_VERSION} / {NL_DEC_WS_FLOATS} - rebuild with `python -m nerf_loam_amd.build --force`")
        if os.environ.get("NL_DEC_LAYOUT"):                 # A/B switch: 1 = one 8-wave decoder workgroup per CU (rounds 1-5), 2 = two 4-wave workgroups
            if L.nl_decoder_set_layout(int(os.environ["NL_DEC_LAYOUT"])) != 0:
                raise NerfLoamHipError("NL_DEC_LAYOUT must be 0, 1 or 2")
        if os.environ.get("NL_GEMM_MODE"):
            if L.nl_decoder_set_gemm_mode(int(os.environ["NL_GEMM_MODE"])) != 0:
                raise NerfLoamHipError("NL_GEMM_MODE must be 0 .. 5")
        if os.environ.get("NL_SAMPLER_MODE"):
            L.nl_geometry_set_sampler_mode(int(os.environ["NL_SAMPLER_MODE"]))
        if os.environ.get("NL_SCAN_SINGLE"):                # A/B switch: 0 = mid-size prefix scans as two launches
            L.nl_geometry_set_scan_single(int(os.environ["NL_SCAN_SINGLE"]))
        if os.environ.get("NL_LANES_PER_RAY"):              # A/B switch for measurements: lanes per ray of the work-list intersect
            L.nl_geometry_set_lanes_per_ray(int(os.environ["NL_LANES_PER_RAY"]))
        if os.environ.get("NL_FIELD_MIDSPAN_FLUSH"):        # A/B switch for measurements (nl_field_set_midspan_flush)
            L.nl_field_set_midspan_flush(int(os.environ["NL_FIELD_MIDSPAN_FLUSH"]))
        if os.environ.get("NL_FIELD_PROBES"):               # A/B switch for measurements (nl_field_set_probes)
            L.nl_field_set_probes(int(os.environ["NL_FIELD_PROBES"]))
        if os.environ.get("NL_FIELD_ONE_ROUND"):            # A/B switch for measurements: the scatter's one-round rule (nl_field_set_one_round)
            L.nl_field_set_one_round(int(os.environ["NL_FIELD_ONE_ROUND"]))
        if os.environ.get("NL_WGRAD2_MODE"):                # A/B switch for measurements (default: the library's own default)
            if L.nl_decoder_set_wgrad2_mode(int(os.environ["NL_WGRAD2_MODE"])) != 0:
                raise NerfLoamHipError("NL_WGRAD2_MODE must be 0, 1 or 2")
        _lib = L
    return _lib


def require_gpu():
    if lib().nl_device_count() <= 0:
        raise NerfLoamHipError("no HIP device visible: the NeRF-LOAM SDF hot path runs only on the GPU (no CPU fallback)")


def check(rc, what=""):
    if rc != 0:
        raise NerfLoamHipError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    """device/host pointer of a torch tensor (must be contiguous) or None"""
    if t is None:
        return None
    assert t.is_contiguous(), "nerfloam_hip needs contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def kernel_modes(gemm_mode=None, wgrad2_mode=None, dec_layout=None):
    """NL_KERNEL_MODES | NL_KERNEL_LAYOUT of include/nerfloam_hip.h: the decoder kernel selection of one call / one NlIterDesc; None = the
    process default (NL_GEMM_MODE / NL_WGRAD2_MODE / NL_DEC_LAYOUT, nl_decoder_set_*)"""
    g = -1 if gemm_mode is None else int(gemm_mode)
    w = -1 if wgrad2_mode is None else int(wgrad2_mode)
    l = 0 if dec_layout is None else int(dec_layout)
    if not (-1 <= g <= 5 and -1 <= w <= 2 and 0 <= l <= 2):
        raise ValueError(f"gemm_mode {gemm_mode} / wgrad2_mode {wgrad2_mode} / dec_layout {dec_layout}: 0..5 / 0..2 / 1..2 or None")
    return ((g + 1) & 0xFF) | (((w + 1) & 0xFF) << 8) | (l << 16)


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
