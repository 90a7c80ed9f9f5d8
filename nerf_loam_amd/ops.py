"""Typed thin wrappers: torch device tensors -> C ABI of libnerfloam_hip.so (include/nerfloam_hip.h).

torch is plumbing here (device memory + streams); every function launches hand-written HIP kernels
on the current torch stream and returns nothing (outputs are caller-allocated).  No function in
this module has a CPU or PyTorch fallback."""
import ctypes

import torch

from . import _lib as L
from ._lib import check, ptr, stream_ptr

I32, F32 = torch.int32, torch.float32


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise L.NerfLoamHipError(f"{name} must be a CUDA (HIP) tensor - the SDF hot path has no CPU fallback")
    if t.dtype != dtype:
        raise L.NerfLoamHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.NerfLoamHipError(f"{name} must be a contiguous tensor")


def svo_intersect(ray_start, ray_dir, points, children, voxelsize, n_max, idx, min_depth, max_depth):
    for t, d, n in ((ray_start, F32, "ray_start"), (ray_dir, F32, "ray_dir"), (points, F32, "points"), (children, I32, "children"),
                    (idx, I32, "idx"), (min_depth, F32, "min_depth"), (max_depth, F32, "max_depth")):
        _chk(t, d, n)
    b, m = ray_start.shape[0], ray_start.shape[1]
    check(L.lib().nl_svo_intersect(ptr(ray_start), ptr(ray_dir), ptr(points), ptr(children), b, m, points.shape[1],
                                   float(voxelsize), int(n_max), ptr(idx), ptr(min_depth), ptr(max_depth), stream_ptr()), "nl_svo_intersect")


def inverse_cdf_sampling(pts_idx, min_depth, max_depth, noise, probs, steps, fixed_step_size, s_idx, s_depth, s_dists):
    for t, d, n in ((pts_idx, I32, "pts_idx"), (min_depth, F32, "min_depth"), (max_depth, F32, "max_depth"), (noise, F32, "uniform_noise"),
                    (probs, F32, "probs"), (steps, F32, "steps"), (s_idx, I32, "sampled_idx"), (s_depth, F32, "sampled_depth"),
                    (s_dists, F32, "sampled_dists")):
        _chk(t, d, n)
    G, m, P = pts_idx.shape
    check(L.lib().nl_inverse_cdf_sampling(ptr(pts_idx), ptr(min_depth), ptr(max_depth), ptr(noise), ptr(probs), ptr(steps),
                                          G, m, P, noise.shape[-1], float(fixed_step_size), ptr(s_idx), ptr(s_depth), ptr(s_dists),
                                          stream_ptr()), "nl_inverse_cdf_sampling")


def ray_intersect(N, rays_d_sensor, points_gt, cos_gt, frame_id, poses12, blk_hdr, blk_ids, root_side, voxel_size, max_distance,
                  rays_d_world, gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, scratch_rays, lanes=0):
    """lanes: lanes per ray of the traversal's work-list, 0 = by ray count (MapDevice.isect_lanes_for picks 32 on an accumulated map)"""
    check(L.lib().nl_ray_intersect_lanes(int(N), ptr(rays_d_sensor), ptr(points_gt), ptr(cos_gt), ptr(frame_id), ptr(poses12), ptr(blk_hdr),
                                         ptr(blk_ids), int(root_side), float(voxel_size), float(max_distance), ptr(rays_d_world), ptr(gt_dist),
                                         ptr(hit_idx), ptr(hit_t0), ptr(hit_t1), ptr(hit_count), ptr(counters), ptr(scratch_rays), int(lanes),
                                         stream_ptr()), "nl_ray_intersect_lanes")


def exclusive_scan(inp, out, n, flag_mode, total_out, workspace):
    check(L.lib().nl_exclusive_scan_i32(ptr(inp), ptr(out), int(n), int(flag_mode), ptr(total_out), ptr(workspace), stream_ptr()),
          "nl_exclusive_scan_i32")


def compact_hit_rays(N, hit_count, hit_rank, ray_of_rank):
    check(L.lib().nl_compact_hit_rays(int(N), ptr(hit_count), ptr(hit_rank), ptr(ray_of_rank), stream_ptr()), "nl_compact_hit_rays")


def sample_rays(emit, N, hit_idx, hit_t0, hit_t1, hit_count, hit_rank, ray_of_rank, cos_gt, gt_dist, step_size, tau, max_depth,
                seed, use_hash_noise, tail_always, ray_id_base, seed_mix, row_first, counters, samp_count, samp_off, capacity, s_vox, s_depth,
                s_dist, s_ray):
    check(L.lib().nl_sample_rays(int(emit), int(N), ptr(hit_idx), ptr(hit_t0), ptr(hit_t1), ptr(hit_count), ptr(hit_rank), ptr(ray_of_rank),
                                 ptr(cos_gt), ptr(gt_dist), float(step_size), float(tau), float(max_depth),
                                 ctypes.c_uint(int(seed) & 0xFFFFFFFF), int(use_hash_noise), int(tail_always), int(ray_id_base),
                                 ptr(seed_mix), ptr(row_first), ptr(counters), ptr(samp_count), ptr(samp_off), int(capacity), ptr(s_vox), ptr(s_depth), ptr(s_dist),
                                 ptr(s_ray), stream_ptr()), "nl_sample_rays")


def loss_finalize(counters, loss_scalars, fs_weight, sdf_weight, tau, max_depth, capacity):
    check(L.lib().nl_loss_finalize(ptr(counters), ptr(loss_scalars), float(fs_weight), float(sdf_weight), float(tau), float(max_depth),
                                   int(capacity), stream_ptr()), "nl_loss_finalize")


def criterion_forward(sdf, z_vals, valid_u8, points, cos, ray_idx, truncation, max_depth, fs_weight, sdf_weight, workspace, out):
    """Criterion.forward on caller tensors (include/nerfloam_hip.h nl_criterion_forward): out[8] on the device."""
    for t, d, n in ((sdf, F32, "sdf"), (z_vals, F32, "z_vals"), (valid_u8, torch.uint8, "valid_mask"), (points, F32, "points"), (cos, F32, "cos"),
                    (ray_idx, I32, "ray_idx"), (out, F32, "out")):
        _chk(t, d, n)
    R, S = sdf.shape
    check(L.lib().nl_criterion_forward(int(R), int(S), ptr(sdf), ptr(z_vals), ptr(valid_u8), ptr(points), ptr(cos), ptr(ray_idx), float(truncation),
                                       float(max_depth), float(fs_weight), float(sdf_weight), ptr(workspace), ptr(out), stream_ptr()),
          "nl_criterion_forward")


def criterion_backward(sdf, z_vals, valid_u8, points, cos, ray_idx, truncation, max_depth, fs_weight, sdf_weight, out, grad_loss, dsdf):
    for t, d, n in ((sdf, F32, "sdf"), (z_vals, F32, "z_vals"), (valid_u8, torch.uint8, "valid_mask"), (points, F32, "points"), (cos, F32, "cos"),
                    (ray_idx, I32, "ray_idx"), (out, F32, "out"), (grad_loss, F32, "grad_loss"), (dsdf, F32, "dsdf")):
        _chk(t, d, n)
    R, S = sdf.shape
    check(L.lib().nl_criterion_backward(int(R), int(S), ptr(sdf), ptr(z_vals), ptr(valid_u8), ptr(points), ptr(cos), ptr(ray_idx), float(truncation),
                                        float(max_depth), float(fs_weight), float(sdf_weight), ptr(out), ptr(grad_loss), ptr(dsdf), stream_ptr()),
          "nl_criterion_backward")


def gather_trilinear(loss_scalars, s_vox, s_depth, s_ray, rays_d_world, frame_id, poses12, n_frames, centres, vertex_rows, emb,
                     voxel_size, X, nblocks):
    check(L.lib().nl_gather_trilinear(ptr(loss_scalars), ptr(s_vox), ptr(s_depth), ptr(s_ray), ptr(rays_d_world), ptr(frame_id),
                                      ptr(poses12), int(n_frames), ptr(centres), ptr(vertex_rows), ptr(emb), float(voxel_size), ptr(X),
                                      int(nblocks), stream_ptr()), "nl_gather_trilinear")


def gather_points(xyz, vox, centres, vertex_rows, emb, voxel_size, X):
    check(L.lib().nl_gather_points(xyz.shape[0], ptr(xyz), ptr(vox), ptr(centres), ptr(vertex_rows), ptr(emb), float(voxel_size), ptr(X),
                                   stream_ptr()), "nl_gather_points")


def gather_grid(n_vox, vox0, res, lin, centres, vertex_rows, emb, voxel_size, X):
    """get_scores' res^3 points per voxel, generated on the device (include/nerfloam_hip.h nl_gather_grid) -> X[n_vox * res^3, 16]"""
    check(L.lib().nl_gather_grid(int(n_vox), int(vox0), int(res), ptr(lin), ptr(centres), ptr(vertex_rows), ptr(emb), float(voxel_size), ptr(X),
                                 stream_ptr()), "nl_gather_grid")


def marching_cubes(sdf, centres, voxel_size):
    """MeshExtractor.marching_cubes on the device (include/nerfloam_hip.h nl_mc_count / nl_mc_emit): sdf [n, res, res, res] fp32, centres [n, >= 3] fp32
    -> (verts [N, 3] fp32, faces [M, 3] int32) device tensors, voxel after voxel.  One host read-back (the two totals) sizes the outputs."""
    _chk(sdf, F32, "sdf"); _chk(centres, F32, "centres")
    n, res = int(sdf.shape[0]), int(sdf.shape[1])
    if sdf.dim() != 4 or sdf.shape[2] != res or sdf.shape[3] != res or centres.dim() != 2 or centres.shape[0] != n or centres.shape[1] < 3:
        raise L.NerfLoamHipError(f"marching_cubes: sdf {tuple(sdf.shape)} must be [n, res, res, res] and centres {tuple(centres.shape)} [n, >= 3]")
    dev = sdf.device
    counts = torch.zeros(2, n + 1, dtype=I32, device=dev)                 # [0]: vertices, [1]: triangles; the last column receives the totals
    offs = torch.empty(2, max(n, 1), dtype=I32, device=dev)
    ws = torch.empty(max(1, (n + 1023) // 1024) + 8, dtype=I32, device=dev)
    if n > 0:
        check(L.lib().nl_mc_count(ptr(sdf), n, res, ptr(counts[0]), ptr(counts[1]), stream_ptr()), "nl_mc_count")
        for r in range(2):
            exclusive_scan(counts[r], offs[r], n, 0, counts[r, n:], ws)
    nv, nt = (int(x) for x in counts[:, n].tolist())
    verts = torch.empty(nv, 3, dtype=F32, device=dev)
    faces = torch.empty(nt, 3, dtype=I32, device=dev)
    if nv > 0:
        check(L.lib().nl_mc_emit(ptr(sdf), ptr(centres), int(centres.stride(0)), n, res, float(voxel_size), ptr(offs[0]), ptr(offs[1]), ptr(verts), ptr(faces),
                                 stream_ptr()), "nl_mc_emit")
    return verts, faces


def decoder_fwd_bwd(loss_scalars, X, params, W2T, s_ray, s_depth, cos_gt, gt_dist, sdf, dsdf, dX, partials, relu2_mask, nslabs,
                    train_decoder, counters, modes=0):
    """modes: _lib.kernel_modes(gemm_mode, wgrad2_mode) - the kernel selection of this call (0: the process defaults); the
    fwd_bwd / wgrad2 / reduce calls of one iteration must use the same word"""
    check(L.lib().nl_decoder_fwd_bwd_m(ptr(loss_scalars), ptr(X), ptr(params), ptr(W2T), ptr(s_ray), ptr(s_depth), ptr(cos_gt), ptr(gt_dist),
                                       ptr(sdf), ptr(dsdf), ptr(dX), ptr(partials), ptr(relu2_mask), int(nslabs), int(train_decoder),
                                       ptr(counters), int(modes), stream_ptr()), "nl_decoder_fwd_bwd_m")


def decoder_wgrad2(loss_scalars, X, params, dsdf, relu2_mask, partials, nslabs, modes=0):
    check(L.lib().nl_decoder_wgrad2_m(ptr(loss_scalars), ptr(X), ptr(params), ptr(dsdf), ptr(relu2_mask), ptr(partials), int(nslabs),
                                      int(modes), stream_ptr()), "nl_decoder_wgrad2_m")


def decoder_forward(X, params, W2T, P, sdf, nblocks, modes=0):
    check(L.lib().nl_decoder_forward_m(ptr(X), ptr(params), ptr(W2T), int(P), ptr(sdf), int(nblocks), int(modes), stream_ptr()),
          "nl_decoder_forward_m")


def reduce_partials(partials, nslabs, n, out):
    check(L.lib().nl_reduce_partials(ptr(partials), int(nslabs), int(n), ptr(out), stream_ptr()), "nl_reduce_partials")


def decoder_reduce(partials, nslabs, params, grad_out, modes=0):
    check(L.lib().nl_decoder_reduce_m(ptr(partials), int(nslabs), ptr(params), ptr(grad_out), int(modes), stream_ptr()), "nl_decoder_reduce_m")


def decoder_transpose_w2(params, W2T):
    check(L.lib().nl_decoder_transpose_w2(ptr(params), ptr(W2T), stream_ptr()), "nl_decoder_transpose_w2")


def touched_rows(t):
    """(list, count, flags[, copies, copy_stride]) -> NlTouchedRows pointer argument, None -> NULL (dense optimiser sweep).  copies / copy_stride: the
    replicated accumulators of include/nerfloam_hip.h (copy c of the gradient accumulators starts copy_stride floats behind copy c - 1)"""
    if t is None:
        return None
    copies, stride = (int(t[3]), int(t[4])) if len(t) > 3 else (1, 0)
    return ctypes.byref(L.NlTouchedRows(ctypes.sizeof(L.NlTouchedRows), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), copies, stride))


def trilinear_bwd(loss_scalars, s_vox, s_depth, s_ray, rays_d_world, rays_d_sensor, frame_id, poses12, n_frames, centres, vertex_rows,
                  emb, voxel_size, dX, g_emb, g_pose, nblocks, touched=None):
    """touched: (list, count, flags) - the rows whose accumulators receive a contribution are recorded (include/nerfloam_hip.h NlTouchedRows)"""
    check(L.lib().nl_trilinear_bwd_t(ptr(loss_scalars), ptr(s_vox), ptr(s_depth), ptr(s_ray), ptr(rays_d_world), ptr(rays_d_sensor),
                                     ptr(frame_id), ptr(poses12), int(n_frames), ptr(centres), ptr(vertex_rows), ptr(emb), float(voxel_size),
                                     ptr(dX), ptr(g_emb), ptr(g_pose), int(nblocks), touched_rows(touched), stream_ptr()), "nl_trilinear_bwd")


def touched_rows_reset(touched, g_emb, emb_m, emb_v):
    check(L.lib().nl_touched_rows_reset(touched_rows(touched), ptr(g_emb), ptr(emb_m), ptr(emb_v), stream_ptr()), "nl_touched_rows_reset")


def unpack_samples(loss_scalars, s_ray, samp_off, hit_rank, sdf, depth, S_stride, out_sdf, out_z, out_valid):
    check(L.lib().nl_unpack_samples(ptr(loss_scalars), ptr(s_ray), ptr(samp_off), ptr(hit_rank), ptr(sdf), ptr(depth), int(S_stride),
                                    ptr(out_sdf), ptr(out_z), ptr(out_valid), stream_ptr()), "nl_unpack_samples")


def adam_prepare(state, lr_emb, lr_dec, lr_pose):
    check(L.lib().nl_adam_prepare(ptr(state), float(lr_emb), float(lr_dec), float(lr_pose), stream_ptr()), "nl_adam_prepare")


def adam_embeddings(emb, g_acc, m, v, state):
    check(L.lib().nl_adam_embeddings(ptr(emb), ptr(g_acc), ptr(m), ptr(v), emb.numel(), ptr(state), stream_ptr()), "nl_adam_embeddings")


def embedding_grad_bf16(g_acc, g_bf16):
    check(L.lib().nl_embedding_grad_bf16(ptr(g_acc), ptr(g_bf16), g_acc.numel(), stream_ptr()), "nl_embedding_grad_bf16")


def adam_f32(p, g, m, v, state, group):
    check(L.lib().nl_adam_f32(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(state), int(group), stream_ptr()), "nl_adam_f32")


def pose_matrices(pose6, poses12):
    check(L.lib().nl_pose_matrices(ptr(pose6), ptr(poses12), pose6.shape[0], stream_ptr()), "nl_pose_matrices")


def optimiser_step(state, lr_emb, lr_dec, lr_pose, emb, dec, pose, counters=None, skip_mode=0, touched=None):
    """one launch for the whole optimiser step; emb = (emb, g_acc, m, v) or None, dec = (params, grad, m, v, workspace) or None,
    pose = (pose6, g_pose, m, v, enable, grad6_out, poses12, apply) or None; skip_mode: see include/nerfloam_hip.h; touched: the embedding
    group sweeps these rows (NlTouchedRows) instead of the whole table"""
    e = [ptr(t) for t in emb] + [emb[0].numel()] if emb else [None] * 4 + [0]
    d = [ptr(t) for t in dec] if dec else [None] * 5
    q = [ptr(t) for t in pose[:7]] + [pose[0].shape[0], int(pose[7])] if pose else [None] * 7 + [0, 0]
    check(L.lib().nl_optimiser_step_t(ptr(state), float(lr_emb), float(lr_dec), float(lr_pose), *e, *d, *q, ptr(counters), int(skip_mode),
                                      None, None, touched_rows(touched), stream_ptr()), "nl_optimiser_step")


def pose_step(pose6, g_pose, m, v, enable, grad6_out, poses12, state, apply):
    check(L.lib().nl_pose_step(ptr(pose6), ptr(g_pose), ptr(m), ptr(v), ptr(enable), ptr(grad6_out), ptr(poses12), pose6.shape[0],
                               ptr(state), int(apply), stream_ptr()), "nl_pose_step")


def mfma_selftest(A32, B32, D32, A16, B16, D16):
    check(L.lib().nl_mfma_selftest(ptr(A32), ptr(B32), ptr(D32), ptr(A16), ptr(B16), ptr(D16), stream_ptr()), "nl_mfma_selftest")


def unit_dirs(points, out_rays_d, out_rays_norm=None):
    """LidarFrame.get_rays (lidarFrame.py:47-52) for a resident scan: points [M,3] -> rays_d [M,3] (+ rays_norm [M])"""
    check(L.lib().nl_unit_dirs(int(points.shape[0]), ptr(points), ptr(out_rays_d), ptr(out_rays_norm), stream_ptr()), "nl_unit_dirs")


def select_rays(M, n_select, seed, rays_d, points, cos_in, frame, out_rays_d, out_points, out_cos, out_frame_id, mask_out, workspace):
    check(L.lib().nl_select_rays(int(M), int(n_select), int(seed) & 0xFFFFFFFF, ptr(rays_d), ptr(points), ptr(cos_in), int(frame),
                                 ptr(out_rays_d), ptr(out_points), ptr(out_cos), ptr(out_frame_id), ptr(mask_out), ptr(workspace),
                                 stream_ptr()), "nl_select_rays")


def select_rays_batch(Ms, ns, seeds, rays_d, points, cos_in, masks, out_off, out_rays_d, out_points, out_cos, out_frame_id, workspace, parity,
                      fail_word=None):
    """all frames of a call in two launches; returns False (nothing launched) when a frame's shape is outside the window method's
    range - the caller then uses select_rays per frame"""
    F = len(Ms)
    I, U, PP = ctypes.c_int * F, ctypes.c_uint * F, ctypes.c_void_p * F
    rc = L.lib().nl_select_rays_batch(F, I(*[int(m) for m in Ms]), I(*[int(n) for n in ns]), U(*[int(s) & 0xFFFFFFFF for s in seeds]),
                                      PP(*[(t.data_ptr() if t is not None else None) for t in rays_d]), PP(*[t.data_ptr() for t in points]),
                                      PP(*[t.data_ptr() for t in cos_in]), PP(*[(t.data_ptr() if t is not None else None) for t in masks]),
                                      I(*[int(x) for x in out_off]), ptr(out_rays_d), ptr(out_points), ptr(out_cos), ptr(out_frame_id),
                                      ptr(workspace), int(parity), ptr(fail_word), stream_ptr())
    if rc == 4:
        return False
    check(rc, "nl_select_rays_batch")
    return True


def scan_hit_rays(hit_count, hit_rank, ray_of_rank, N, total_out, total_out2, workspace):
    check(L.lib().nl_scan_hit_rays(ptr(hit_count), ptr(hit_rank), ptr(ray_of_rank), int(N), ptr(total_out), ptr(total_out2), ptr(workspace),
                                   stream_ptr()), "nl_scan_hit_rays")
