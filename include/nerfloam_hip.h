/*
 * nerfloam_hip.h -- C ABI of libnerfloam_hip.so: the MI355X (gfx950) implementation of NeRF-LOAM's
 * per-iteration neural-SDF path.  Plain pointers and sizes only (no torch types); every device
 * pointer is a HIP device pointer, `stream` is a hipStream_t passed as void*.  All entry points
 * return 0 (NL_OK) or an NL_ERR_* code; nothing calls exit() (the reference's CUDA_CHECK_ERRORS
 * does, third_party/sparse_voxels/include/cuda_utils.h:37-48).  The library allocates nothing on
 * the hot path: workspaces are caller-owned, sizes are data-independent (worst case), data-dependent
 * counts live in a device counter block, so a whole iteration is launchable without host syncs.
 *
 * Each declaration cites the reference interface it replaces (paths under the reference repo).
 * Reference-side bindings: see INTEGRATION.md.
 */
#ifndef NERFLOAM_HIP_H
#define NERFLOAM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define NL_OK 0
#define NL_ERR_INVALID_ARG 1
#define NL_ERR_LAUNCH 2
#define NL_ERR_NO_DEVICE 3
#define NL_ERR_CAPACITY 4

/* layout constants shared with the host side */
#define NL_MAX_HITS 20          /* src/variations/voxel_helpers.py:533 */
#define NL_CNT_INTS 16          /* int32 counters ... */
#define NL_CNT_DOUBLES 4        /* ... followed by double sums: counter block = 16*4 + 4*8 bytes */
#define NL_LOSS_SCALARS_BYTES 48
#define NL_DEC_PARAMS 70401     /* W1[256x16] b1[256] W2[256x256] b2[256] W3[256] b3[1] */
#define NL_DEC_WS_FLOATS 401424    /* decoder weight workspace: W2^T fp32 + 2 x 3 bf16 operand planes + 2 x 2 fp16 operand planes of W2 + 2 x 2 fp16 operand planes of W1 (nl_dec_ws_floats() returns it) */
#define NL_EMB_CHANNELS 16

/* Multi-GPU ray sharding: fold the all-gathered counter blocks gathered[world][NL_CNT_INTS + 2*NL_CNT_DOUBLES] (ints) into
 * this rank's block.  stage 1 (after nl_ray_intersect): NLC_R_GLOBAL, NLC_R_OFFSET, global NLC_HMAX;
 * stage 2 (after the counting sampler pass): summed loss normalisers, max samples per ray, padded-slot constants. */
int nl_dist_merge_counters(const int* gathered, int world, int rank, int stage, int* counters, void* stream);
int nl_version(void);
int nl_device_count(void);              /* number of HIP devices visible (0 => product cannot run) */
int nl_decoder_grid_hint(void);         /* slabs to provide in `partials` = upper bound of the persistent decoder grids: 2 per compute unit of the current device */

/* ---- (b1) drop-in operators of the reference's `grid` pybind module ----------------------------
 * third_party/sparse_voxels/src/binding.cpp:10-21, include/intersect.h:14-15, include/sample.h:12-14 */

/* grid.svo_intersect(ray_start[B,m,3], ray_dir[B,m,3], points[B,n,3], children[B,n,9], voxelsize, n_max)
 *   -> idx i32[B,m,n_max] (-1 padded), min_depth/max_depth f32[B,m,n_max] (0 where unused)
 * third_party/sparse_voxels/src/intersect.cpp:83-112, intersect_gpu.cu:193-272.  Any n_max >= 1 (the reference's caller passes 20). */
int nl_svo_intersect(const float* ray_start, const float* ray_dir, const float* points, const int* children,
                     int b, int m, int n, float voxelsize, int n_max,
                     int* idx, float* min_depth, float* max_depth, void* stream);

/* grid.inverse_cdf_sampling(pts_idx[G,m,P], min_depth, max_depth, uniform_noise[G,m,T], probs, steps[G,m], fixed_step)
 *   -> sampled_idx i32[G,m,T], sampled_depth, sampled_dists f32[G,m,T]; outputs must be pre-filled (-1, 0, 0)
 * third_party/sparse_voxels/src/sample.cpp:56-95, sample_gpu.cu:133-239 (tail loop kept bug-for-bug). */
int nl_inverse_cdf_sampling(const int* pts_idx, const float* min_depth, const float* max_depth, const float* noise,
                            const float* probs, const float* steps, int b, int num_rays, int max_hits, int max_steps,
                            float fixed_step_size, int* sampled_idx, float* sampled_depth, float* sampled_dists, void* stream);

/* ---- fused iteration stages (replace src/variations/render_helpers.py:190-318 render_rays and the
 *      autograd/optimiser part of :356-423 / :452-512) ------------------------------------------- */

/* ray set-up + ray_intersect: render_helpers.py:366-388 (d_world = d_sensor R^T, o = t),
 * voxel_helpers.py:531-567 (DFS intersect, -1 -> max_distance, sort by t_min, cull).
 * poses[F,12] = rotation row-major | translation.  frame_id may be NULL (single frame).
 * Octree in the children-block layout (the reference's voxel_center_xyz [n,3] + voxel_structure [n,9] re-packed
 * once per map update; one block per interior node, numbered breadth-first): blk_hdr[B] = int2 (first child
 * block, exist mask | own-a-block mask << 8; children blocks are consecutive in octant order), blk_ids[B][8] =
 * child node ids, root_side = side of node 0 in voxels (voxel_structure[0][8]); block 0 is a pseudo block for the
 * root: id slot 0 = node 0, slots 1..7 describe the single-child chain under the root (1 = its length n, 2 / 3 = the n octants, 3 bits
 * each, level 0 first, 30 bits in slot 2, the rest in slot 3; 4 = the block the chain ends in; 5..7 = that block's lattice position;
 * n = 0 and end block = the root's children block when the root branches) - the kernel runs the chain's slab tests in registers and
 * starts its work-list at the chain's end.  Node centres are recomputed from the lattice path ((xyz + side/2) * voxel_size, exact), not loaded.
 * Outputs: rays_d_world[N,3], gt_dist[N] = ||p||*cos, hit_idx/t0/t1[N,20] (only the first hit_count[r]
 * entries of a row are written: sorted by t_min, culled), hit_count[N];
 * counters[NLC_HMAX] is raised (atomic max).  counters must be zeroed per iteration. */
int nl_ray_intersect(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                     const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                     float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                     int* counters, int* scratch_rays /* >= N ints, free to reuse afterwards */, void* stream);

/* exclusive scan of in[0..n) (flag_mode: of (in[i] > 0)); total -> *total_out (device).
 * workspace >= ceil(n/1024) ints.  Replaces the boolean-mask compactions of render_helpers.py:219-257. */
int nl_exclusive_scan_i32(const int* in, int* out, int n, int flag_mode, int* total_out, int* workspace, void* stream);
int nl_compact_hit_rays(int N, const int* hit_count, const int* hit_rank, int* ray_of_rank, void* stream);
/* the two calls above fused (one launch up to 4096 rays, two beyond): hit_rank = exclusive scan of (hit_count > 0),
 * ray_of_rank[rank] = ray, number of hit rays -> *total_out and, if not NULL, *total_out2 */
int nl_scan_hit_rays(const int* hit_count, int* hit_rank, int* ray_of_rank, int N, int* total_out, int* total_out2, int* workspace,
                     void* stream);
/* nl_ray_intersect + nl_scan_hit_rays in one call (ray_of_rank doubles as the intersect kernel's scratch list, as in the stage-wise
 * sequence).  Up to 4096 rays - the reference's live shapes, where an iteration is launch-bound - the fallback pass of the
 * intersect and the scan are ONE launch; beyond, the same launches as the two calls.  Same results. */
int nl_ray_intersect_scan(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                          const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                          float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                          int* counters, int* ray_of_rank, int* hit_rank, int* total_out, int* total_out2, int* scan_ws, void* stream);
/* The same two calls with the lanes a ray's work-list gets chosen by the caller: 0 = by ray count (32 up to 4096 rays, 16 up to 16 384, 8
 * beyond), or 4 / 8 / 16 / 32.  Same results bit for bit; what changes is the number of traversal rounds.  Between 4097 and 16 384 rays 32 lanes
 * pay on an ACCUMULATED map only, where a ray crosses many occupied voxels and has more than 16 nodes pending per round (150-scan map:
 * 16 384 rays 120 -> 95 us; a one-scan map 41 -> 65): the host side passes 32 for maps of >= 60 000 children blocks there
 * (nerf_loam_amd/pipeline.py MapDevice.isect_lanes_for); NlIterDesc.isect_lanes carries the same choice. */
/* the lanes-per-ray choice itself (csrc/nl_common.h: the launch-shape table): 32 up to 4096 rays; up to 16 384 rays 32 on a map of >= 60 000
 * children blocks, else 16; 8 beyond.  n_children_blocks 0 = unknown (the rule by ray count alone). */
int nl_isect_lanes_for(int n_rays, int n_children_blocks);
int nl_ray_intersect_lanes(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                           const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                           float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                           int* counters, int* scratch_rays, int lanes, void* stream);
int nl_ray_intersect_scan_lanes(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                                const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                                float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                                int* counters, int* ray_of_rank, int* hit_rank, int* total_out, int* total_out2, int* scan_ws, int lanes, void* stream);
/* nl_ray_intersect_scan_lanes + nl_dist_x1_pack: the ray-sharded iteration's first stage with the send block of its first exchange
 * (x1_send = [counter block | x1_rays bytes: the hit count of ray i, 0 beyond N]; x1_rays >= N, a multiple of 16; total_out must be
 * counters + 0, the block's hit-ray slot).  Between 4097 and 32 768 rays - a rank's share of a scan - the scan's own launch packs the block
 * (its workgroups hold the hit counts anyway, its last one the total): one ~5 us launch fewer on the rank's critical path. */
int nl_ray_intersect_scan_x1(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                             const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                             float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                             int* counters, int* ray_of_rank, int* hit_rank, int* total_out, int* total_out2, int* scan_ws, int lanes,
                             int* x1_send, int x1_rays, void* stream);

/* LidarFrame.get_rays (src/lidarFrame.py:47-52): rays_d[M,3] = points / (||points||_2 + 1e-8), rays_norm[M] (optional) = that
 * denominator - the arithmetic of the reference's host torch ops, bit for bit (nl_device_math.h nl_unit_dir).  The selection entry
 * points below compute the same direction in flight when their rays_d argument is NULL. */
int nl_unit_dirs(int M, const float* points, float* out_rays_d, float* out_rays_norm /* optional */, void* stream);

/* On-device ray selection (LidarFrame.sample_rays -> sampling_without_replacement, src/lidarFrame.py:55-57,
 * src/utils/sample_util.py:4-19): a uniformly random subset of n_select of the frame's M returns, kept in dataset order,
 * gathered into out_*[0..n_select) (out_frame_id, mask_out optional; mask_out[M] = the reference's boolean sample_mask).
 * Deterministic in (seed, M, n_select).  workspace >= NL_SELECT_WS_INTS(M) ints.  rays_d may be NULL (also per frame in the batch
 * forms): the directions are then computed from `points` (nl_unit_dirs' arithmetic) - a frame is resident as points + cos only. */
#define NL_SELECT_WS_INTS(M) (264 + 2 * (M) + ((M) + 1023) / 1024 + 8)
int nl_select_rays(int M, int n_select, unsigned seed, const float* rays_d, const float* points, const float* cos_in, int frame,
                   float* out_rays_d, float* out_points, float* out_cos, int* out_frame_id, unsigned char* mask_out,
                   int* workspace, void* stream);

/* The same selection for ALL frames of a call in two launches (n << M: a window around the expected threshold key, exact k-th largest
 * among the few candidates inside it; same subset as nl_select_rays).  Host arrays of F entries each; out_off[f] = first output
 * slot of frame f.  Returns 4 (capacity) for shapes outside the method's range - fall back to nl_select_rays per frame.
 * workspace: NL_SELECT_BATCH_WS_INTS(F) ints, zero-filled ONCE at allocation; parity alternates 0 / 1 from call to call. */
#define NL_SELECT_BATCH_WS_INTS(F) ((F) * (8 + 4 * 128 + 2 * 4096))
int nl_select_rays_batch(int F, const int* M, const int* n_select, const unsigned* seed, const float* const* rays_d,
                         const float* const* points, const float* const* cos_in, unsigned char* const* mask_out, const int* out_off,
                         float* out_rays_d, float* out_points, float* out_cos, int* out_frame_id, int* workspace, int parity,
                         int* fail_word /* optional device word set to 1 when the window was missed */, void* stream);

/* the same with the frame id of every entry given explicitly (frame_ids[f] is written to out_frame_id; NULL: f): entries may then be
 * (iteration, frame) pairs - all ray subsets of a whole optimisation call drawn up front, eight entries per call of this function */
int nl_select_rays_batch_ex(int F, const int* M, const int* n_select, const unsigned* seed, const float* const* rays_d,
                            const float* const* points, const float* const* cos_in, unsigned char* const* mask_out, const int* out_off,
                            const int* frame_ids, float* out_rays_d, float* out_points, float* out_cos, int* out_frame_id, int* workspace,
                            int parity, int* fail_word, void* stream);

/* ray_sample: voxel_helpers.py:571-598 + :262-347 + sample_gpu.cu:133-239.
 * emit = 0: per-ray sample count, S_max and the geometry-only loss normalisers (criterion.py:67-88);
 * emit = 1: compacted (voxel, depth, dist, ray) records at samp_off[ray] (capacity-checked).
 * noise: counter-based hash(seed, ray_id_base + ray, step) clamped to [.001,.999], or 0.5 if !use_hash_noise
 * (the reference draws torch uniform_ noise EVERY iteration, voxel_helpers.py:297-301).  seed_mix (optional, NULL = off):
 * device word - e.g. the optimiser's step counter, word 0 of adam_state - folded into the seed as
 * seed + 0x9E3779B9 * *seed_mix, so a hipGraph-replayed or host-loop iteration draws fresh jitter without host work. */
int nl_sample_rays(int emit, int N, const int* hit_idx, const float* hit_t0, const float* hit_t1, const int* hit_count,
                   const int* hit_rank, const int* ray_of_rank, const float* cos_gt, const float* gt_dist,
                   float step_size, float tau, float max_depth, unsigned seed, int use_hash_noise, int tail_always, int ray_id_base,
                   const unsigned* seed_mix, const int* row_first /* multi-GPU: nl_dist_x1_merge table, else NULL */,
                   int* counters, int* samp_count, const int* samp_off, int capacity,
                   int* s_vox, float* s_depth, float* s_dist, int* s_ray, void* stream);
/* global loss normalisers from the counter block (criterion.py:84-88 weights, :65 mean divisor R*S) */
int nl_loss_finalize(int* counters, void* loss_scalars, float fs_weight, float sdf_weight, float tau, float max_depth,
                     int capacity, void* stream);
/* Criterion.forward on caller tensors (replaces /root/reference/src/criterion.py:16-115 for a caller holding its own render_rays
 * `outputs`: l2, no eikonal term).  sdf, z_vals [R,S] fp32 and valid_mask [R,S] bytes are the padded sample block; points [N,3] and
 * cos [N] the frame's observations, ray_idx [R] the rows of the rays that hit (outputs["ray_mask"].nonzero(), or NULL when R == N).
 * workspace: >= 32 bytes, cleared here.  out[8] = loss, fs_loss, sdf_loss, the two data-dependent weights (:84-88), 2 / (R S), the
 * front and sdf-mask counts - all on the device, nothing is read back.  backward: dsdf [R,S] = grad_loss (device scalar, NULL = 1) *
 * dloss / dsdf with the weights treated as constants (count_nonzero has no gradient), `out` from the forward call. */
int nl_criterion_forward(int R, int S, const float* sdf, const float* z_vals, const unsigned char* valid_mask, const float* points,
                         const float* cos, const int* ray_idx, float truncation, float max_depth, float fs_weight, float sdf_weight,
                         void* workspace, float* out, void* stream);
int nl_criterion_backward(int R, int S, const float* sdf, const float* z_vals, const unsigned char* valid_mask, const float* points,
                          const float* cos, const int* ray_idx, float truncation, float max_depth, float fs_weight, float sdf_weight,
                          const float* out, const float* grad_loss, float* dsdf, void* stream);
/* nl_sample_rays(count) + nl_exclusive_scan_i32 + nl_loss_finalize + nl_sample_rays(emit) as ONE launch up to 8192 rays (single
 * GPU: no row_first table): every workgroup walks its rays once, parks the samples in LDS, obtains its offset by decoupled
 * look-back over the workgroups before it (state: >= 8 * (1 + ceil(N / 32)) bytes only this function touches, zero-initialised
 * once: word 0 counts the launches on the device, so a captured launch can be replayed) and the last workgroup computes the loss
 * normalisers.  Same samples at the same places.
 * Beyond 8192 rays or with state == NULL: the four launches. */
int nl_sample_rays_fused(int N, const int* hit_idx, const float* hit_t0, const float* hit_t1, const int* hit_count,
                         const int* hit_rank, const int* ray_of_rank, const float* cos_gt, const float* gt_dist,
                         float step_size, float truncation, float max_depth, unsigned noise_seed, int use_hash_noise, int tail_always,
                         int ray_id_base, const unsigned* seed_mix, int* counters, int* samp_count, int* samp_off, int capacity,
                         int* s_vox, float* s_depth, float* s_dist, int* s_ray, void* loss_scalars, float fs_weight, float sdf_weight,
                         void* state, int* scan_ws, void* stream);
/* nl_exclusive_scan_i32(samp_count -> samp_off, total -> counters[NLC_P]) + nl_loss_finalize: one launch up to 4096 rays, the same
 * launches as the two calls beyond.  workspace as for nl_exclusive_scan_i32. */
int nl_scan_samples_finalize(const int* samp_count, int* samp_off, int N, int* counters, void* loss_scalars, float fs_weight, float sdf_weight,
                             float tau, float max_depth, int capacity, int* workspace, void* stream);

/* get_features: render_helpers.py:74-93 + :39-70 (gather + trilinear interpolation) -> X[P,16] */
int nl_gather_trilinear(const void* loss_scalars, const int* s_vox, const float* s_depth, const int* s_ray,
                        const float* rays_d_world, const int* frame_id, const float* poses, int n_frames,
                        const float* centres, const int* vertex_rows, const void* emb_bf16, float voxel_size,
                        float* X, int nblocks, void* stream);

/* get_features on explicit world points with their voxel ids (get_scores / eval_points, render_helpers.py:96-188) */
int nl_gather_points(int P, const float* xyz, const int* vox, const float* centres, const int* vertex_rows, const void* emb_bf16,
                     float voxel_size, float* X, void* stream);

/* get_scores' points generated on the device (render_helpers.py:103-121): the res^3 grid points lin[ix], lin[iy], lin[iz] (x voxel_size, + centre) of the
 * voxels [vox0, vox0 + n_vox), row-major in (voxel, ix, iy, iz) like the reference's reshape, -> X[n_vox * res^3, 16].  lin[res] = torch.linspace(-0.5, 0.5, res)
 * (device pointer; the caller's own values, so the points are the reference's bit for bit). */
int nl_gather_grid(int n_vox, int vox0, int res, const float* lin, const float* centres, const int* vertex_rows, const void* emb_bf16, float voxel_size,
                   float* X, void* stream);

/* Marching cubes over get_scores' per-voxel grids, replacing MeshExtractor.marching_cubes (src/utils/mesh_util.py:145-169: skimage.measure.marching_cubes
 * on the CPU, one voxel at a time).  sdf[n_vox][res][res][res] (device, the layout get_scores returns), 2 <= res <= 16.
 *   nl_mc_count: n_verts[v] / n_tris[v] = vertices (lattice edges whose end values change sign) and triangles of voxel v; both 0 for a voxel the reference
 *                skips (min > 0 or max < 0, :158-159).
 *   nl_mc_emit:  vert_off / tri_off = exclusive scans of those counts (nl_exclusive_scan_i32); verts[total][3] = ((lattice position / (res - 1)) - 0.5) *
 *                voxel_size + centres[v] (:162-164; centres row stride `centre_stride` floats), faces[total][3] = indices into verts, voxel after voxel like
 *                the reference's concatenation.  Vertices are shared inside a voxel (one per crossed lattice edge, in (axis, ix, iy, iz) order), faces come
 *                cell by cell from the 256-case table csrc/nl_mc_table.h with normals towards the positive values.
 * The vertex SET is that of any linear-interpolation marching cubes; the triangulation of a cell is derived from the cube's geometry
 * (scripts/gen_mc_table.py), NOT taken from scikit-image (absent here): vertex / face order and the choice on ambiguous faces are this library's own. */
int nl_mc_count(const float* sdf, int n_vox, int res, int* n_verts, int* n_tris, void* stream);
int nl_mc_emit(const float* sdf, const float* centres, int centre_stride, int n_vox, int res, float voxel_size, const int* vert_off, const int* tri_off,
               float* verts, int* faces, void* stream);

/* Decoder forward (src/variations/lidar.py:109-131) + Criterion gradient (src/criterion.py:59-100) +
 * decoder backward (autograd of render_helpers.py:422).  params = decoder block, W2T = decoder weight workspace (nl_decoder_transpose_w2).
 * Outputs sdf[P], dsdf[P], dX[P,16]; with train_decoder: per-workgroup weight-gradient slabs
 * partials[nslabs][NL_DEC_PARAMS] (all but the W2 block; nl_decoder_wgrad2 adds that; sum the slabs with
 * nl_decoder_reduce) and relu2_mask[ceil(P/64)][512] scratch (one 32-bit ReLU word per tile and thread). */
int nl_decoder_fwd_bwd(const void* loss_scalars, const float* X, const float* params, const float* W2T,
                       const int* s_ray, const float* s_depth, const float* cos_gt, const float* gt_dist,
                       float* sdf, float* dsdf, float* dX, float* partials, unsigned* relu2_mask, int nslabs, int train_decoder,
                       int* counters, void* stream);
/* The same four decoder entry points with the kernel selection passed per call instead of taken from the process-wide defaults
 * (4, 2; include/nerfloam_hip_debug.h holds the A/B setters): kernel_modes = NL_KERNEL_MODES(gemm_mode, wgrad2_mode), either
 * mode -1 (or kernel_modes == 0) = the process default.  NlIterDesc.kernel_modes carries the same word for nl_iteration. */
#define NL_KERNEL_MODES(gemm_mode, wgrad2_mode) ((((gemm_mode) + 1) & 0xFF) | ((((wgrad2_mode) + 1) & 0xFF) << 8))
/* OR-ed into kernel_modes: the workgroup layout of the fused decoder kernel under the fp16-pair arithmetic (gemm modes 4 / 5; round 6):
 *   0 = by slab count: two independent 4-wave workgroups per compute unit (k_decoder2: one workgroup's VALU phases run under the other's
 *       matrix phases) when `nslabs` >= 2 per compute unit - what nl_decoder_grid_hint() returns -, else layout 1;
 *   1 = one 8-wave workgroup per compute unit (k_decoder, rounds 1-5);  2 = two 4-wave workgroups per compute unit whatever nslabs is.
 * sdf, dsdf and the saved ReLU words are bit-identical between the layouts; dX and the weight gradients differ by summation order.
 * `nslabs` is the number of slabs `partials` holds AND the upper bound of every persistent grid: the 512-thread kernels (layout 1, dW2,
 * forward-only) launch min(nslabs, compute units) workgroups, layout 2 launches nslabs; nl_decoder_reduce_m sums each column over the slabs
 * its producer wrote (the same rule, from the same two arguments - pass all three calls of an iteration the same nslabs and kernel_modes). */
#define NL_KERNEL_LAYOUT(layout) (((layout) & 3) << 16)
/* the layout nl_iteration (and the Python stage-wise path) put into a kernel_modes word whose layout field is 0: 1 for iterations of up to 16 384 rays (there the
 * iteration is a latency chain and the 8-wave workgroup finishes its one or two tiles sooner; also: those shapes keep round 5's bits), 0 = by slab count beyond */
int nl_decoder_layout_for(int n_rays);
int nl_decoder_fwd_bwd_m(const void* loss_scalars, const float* X, const float* params, const float* W2T,
                         const int* s_ray, const float* s_depth, const float* cos_gt, const float* gt_dist,
                         float* sdf, float* dsdf, float* dX, float* partials, unsigned* relu2_mask, int nslabs, int train_decoder,
                         int* counters, int kernel_modes, void* stream);
int nl_decoder_wgrad2_m(const void* loss_scalars, const float* X, const float* params, const float* dsdf, const unsigned* relu2_mask,
                        float* partials, int nslabs, int kernel_modes, void* stream);
int nl_decoder_reduce_m(const float* partials, int nslabs, const float* params, float* grad_out, int kernel_modes, void* stream);
int nl_decoder_forward_m(const float* X, const float* params, const float* W2T, int P, float* sdf, int nblocks, int kernel_modes,
                         void* stream);
/* second half of the decoder weight gradient: dW2 = dH2^T H1 into partials[slab][W2 block] (train only) */
int nl_decoder_wgrad2(const void* loss_scalars, const float* X, const float* params, const float* dsdf, const unsigned* relu2_mask,
                      float* partials, int nslabs, void* stream);
/* sum of the per-workgroup slabs partials[nslabs][NL_DEC_PARAMS] written by nl_decoder_fwd_bwd + nl_decoder_wgrad2 into the decoder
 * gradient grad_out[NL_DEC_PARAMS] */
int nl_decoder_reduce(const float* partials, int nslabs, const float* params, float* grad_out, void* stream);
/* forward only: Decoder.get_values on a dense batch (mesh-time get_scores, render_helpers.py:96-153) */
int nl_decoder_forward(const float* X, const float* params, const float* W2T, int P, float* sdf, int nblocks, void* stream);
int nl_reduce_partials(const float* partials, int nslabs, int n, float* out, void* stream);
/* Decoder weight workspace W2T[NL_DEC_WS_FLOATS], rebuilt from params after every optimiser step:
 *   floats [0, 65536):        W2 transposed (fp32; forward GEMM B operand of gemm mode 0),
 *   floats [65536, 163840):   "W2X"  = w3_j * W2[j][k] as three bf16 planes (hi + mid + lo == the fp32 value exactly) in
 *                             MFMA-fragment order: dgrad GEMM B operand on the bf16 matrix cores,
 *   floats [163840, 262144):  "W2TX" = W2[n][k], same split and order: forward GEMM B operand on the bf16 matrix cores,
 *   floats [262144, 327680):  "W2H"  = (w3_j * W2[j][k]) * 2^10 as two fp16 planes (hi = f16(x), lo = f16(x - hi), round to nearest), same order,
 *   floats [327680, 393216):  "W2TH" = W2[n][k] * 2^8, likewise: the operands of gemm modes 4 / 5,
 *   floats [393216, 397312):  "W1F"  = W1[k][c] * 2^8 as two fp16 planes in the order of layer 1's B fragments,
 *   floats [397312, 401408):  "W1X"  = the same values in the order of dX's B fragments (csrc/nl_common.h NL_W1F_INDEX / NL_W1X_INDEX; round 6:
 *                             the two-workgroups-per-CU decoder kernel has neither registers nor LDS to keep W1's operand forms resident),
 *   words  [401408, 401424):  the RANGE BLOCK (round 6).  The fp16-pair arithmetic clips an operand that leaves its scaled fp16 range (|X| >= 1023.5 - 255.9 with a
 *                             trainable decoder -, |W1|, |W2| >= 255.9, H1 >= 4094, |w3_j W2[j][k]| >= 63.97, a dgrad sum >= 63.97) instead of overflowing, which a caller
 *                             must be able to notice: word 4 = STATUS, sticky NL_SAT_* bits raised by the decoder kernels at their end (X and the weight planes exactly;
 *                             H1 and the dgrad sums through the bounds max|X| * max_k ||W1[k]||_1 + max|b1| and 256 * max|w3_j W2[j][k]| - a set bit there means "may
 *                             have clipped"), words 0-3 the weight statistics those bounds use (maintained with the planes).  nl_optimiser_step* latches STATUS into
 *                             bit 1 of the call status (adam_state[3]; bit 0 = sample overflow); nl_decoder_range_status reads / clears it.  Zero cost inside the
 *                             kernels' tile loops beyond a running max |X|.  A flagged call is re-run under gemm mode 3 (exact products, unbounded range).
 * The workspace grew in rounds 5 and 6: a caller compiled against an older header would hand over a shorter buffer, so it checks
 * nl_dec_ws_floats() == NL_DEC_WS_FLOATS (and nl_abi_version() == NL_ABI_VERSION) once at start-up - the Python loader does. */
int nl_decoder_transpose_w2(const float* params, float* W2T, void* stream);
int nl_dec_ws_floats(void);             /* NL_DEC_WS_FLOATS of the library that is loaded */
#ifndef NL_SAT_X
#define NL_SAT_X 1u         /* an input X left the fp16-pair range, or is NaN / Inf */
#define NL_SAT_H1 2u        /* bound: the first hidden layer may have clipped */
#define NL_SAT_Q 4u         /* bound: the dgrad sums may have clipped */
#define NL_SAT_PLANES 8u    /* a weight operand plane clipped */
#endif
/* the range block's sticky status word -> *status_out (host pointer; synchronises `stream`); clear != 0 zeroes it afterwards */
int nl_decoder_range_status(float* W2T, unsigned* status_out, int clear, void* stream);
#define NL_ABI_VERSION 6                /* bumped whenever a struct, a workspace size or an argument's meaning changes */
int nl_abi_version(void);               /* NL_ABI_VERSION of the library that is loaded */
/* The two 256-deep GEMMs of nl_decoder_fwd_bwd / nl_decoder_forward (gemm_mode).  Values and accumulation are fp32 in every mode; the
 * modes differ in how the matrix cores form the fp32 x fp32 products:
 *   0 = fp32 matrix cores (v_mfma_f32_32x32x2_f32), 1/16 of the 16-bit rate.
 *   1, 3, 2 = bf16 matrix cores (v_mfma_f32_32x32x16_bf16) on EXACT three-term splits (v = hi + mid + lo exactly, 8 + 8 + 8 bits):
 *       forward  H2 = H1 W2^T:  both operands split, partial products exact in fp32;
 *       dgrad    dH1[i][k] = dsdf_i * sum_j m(i,j) * (w3_j W2[j][k]),  m = the {0,1} ReLU mask as A operand, B split in three.
 *     1 = all nine forward products: the exact fp32 x fp32 products (results differ from mode 0 by summation order only);
 *     3 = eight of the nine (without lo x lo, below 2^-30 of a product; bound proven in rational arithmetic, tests/test_device_math_host.py):
 *         the default of rounds 3-4, what a caller that wants exact products at the best speed pins;
 *     2 = six (also without lo x mid, mid x lo: below 2^-24 of a product).
 *   4, 5 = fp16 matrix cores (v_mfma_f32_32x32x16_f16) on TWO-term splits ("fp16 pairs", round 5): every operand, scaled by a fixed power of two,
 *     is hi + lo with hi = f16(x), lo = f16(x - hi), round to nearest - the pair reproduces x to one fp32 rounding (2^-23; nl_split2_f16 in
 *     csrc/nl_device_math.h), each hi/lo product is exact in fp32.  Forward: 4 = hi hi' + hi lo' + lo hi' (THE DEFAULT: what is dropped is below
 *     2^-22 of a product, under the rounding of the 256-deep fp32 accumulation), 5 = all four; dgrad: the {0,1} mask x two terms; layer 1: all four.
 *     6 + 4 matrix instructions per k-step against 16 + 6 of mode 3.  Measured against the oracle and the reference-generated goldens the modes 1, 3,
 *     4, 5 are indistinguishable (sdf 3e-8, the same gradient bars).  Operands saturate instead of overflowing fp16:
 *     |X| < 1023 (256 with a trainable decoder), |W1| < 256, H1 < 4094, |W2| < 256, |w3_j W2[j][k]| < 64, dgrad sums < 64 - far outside what the decoder of an
 *     SDF map holds, and REPORTED when it happens (the range block of the weight workspace, nl_decoder_transpose_w2 below).
 * dW2 kernel (wgrad2_mode): 0 = fp32 matrix cores; 1, 2 = 16-bit matrix cores on dW2[j][k] = w3_j * sum_i m(i,j) * (dsdf_i * H1[i][k]) with the
 * {0,1} mask m as A operand and the fp32 B operand v split: 1 = into three bf16 terms (exact products, fp32 accumulation), 2 = into an fp16
 * pair of v * sigma, sigma the power of two that puts the launch's largest |dsdf| in [8, 16) (nl_decoder_fwd_bwd leaves that maximum in the
 * loss-scalar block; THE DEFAULT: 64 matrix instructions per 64-sample tile and wave against 96).
 * The selection is a PER-CALL argument (kernel_modes of the *_m entry points, NlIterDesc.kernel_modes); the entry points without it use the
 * library defaults (4, 2).  Changing those defaults process-wide is a test / A-B aid: include/nerfloam_hip_debug.h. */

/* backward of get_features: embedding gradient (fp32 accumulation of bf16-rounded contributions, the
 * CUDA embedding_dense_backward semantics) and pose-gradient partials g_pose[F,12] = (dL/dt, dL/dR), accumulated in fp64
 * (the per-sample terms cancel heavily: fp64 makes the sums independent of the summation order).
 * g_emb / g_pose may be NULL to skip either. */
int nl_trilinear_bwd(const void* loss_scalars, const int* s_vox, const float* s_depth, const int* s_ray,
                     const float* rays_d_world, const float* rays_d_sensor, const int* frame_id, const float* poses, int n_frames,
                     const float* centres, const int* vertex_rows, const void* emb_bf16, float voxel_size,
                     const float* dX, float* g_emb, double* g_pose, int nblocks, void* stream);

/* Embedding rows touched since the optimiser of a call was created.  The reference's torch.optim.Adam sweeps the whole [E,16] table every
 * step (render_helpers.py:341-353,421-423); a row that never received a gradient has zero gradient and zero moments and does not move, so
 * sweeping only the touched rows is bit-identical - and on a KITTI-scale map (1e6 - 1e7 rows) the difference between an
 * optimiser step that costs 160 B x E per iteration and one that costs 160 B x (rays x hits x 8).  flags: one bit per row (ceil(E / 32)
 * words, zero-filled at allocation), list: the rows whose bit is set (capacity E), count: [1].  The scatter kernel records a row the first
 * time it adds to its accumulators (nl_trilinear_bwd_t; the multi-GPU unpack: nl_dist_rows_move_t), the optimiser sweeps the list
 * (nl_optimiser_step_t), nl_touched_rows_reset clears the listed rows' accumulators / moments / flags and empties the list at the start of
 * the next call (a fresh Adam per call, render_helpers.py:353) - no E-sized memset.  NULL = the dense behaviour everywhere.
 * copies / copy_stride (round 5): REPLICATED accumulators.  On an accumulated map the voxels around the sensor are crossed by every ray of a scan, a
 * few dozen rows receive thousands of same-address atomics per iteration, and those queue at the memory side (scripts/micro/atomic_rows.hip: 30 % of
 * 384 k row atomics on 48 rows: 76 us against 23 us spread out; with 16 copies 29 us).  With copies > 1 the accumulator array holds `copies` arrays
 * copy_stride floats apart (g_emb + c * copy_stride), a wave of the scatter adds into copy (its index mod copies), and the optimiser's sweep over the
 * touched rows - the only reader - sums a row's copies in copy order before the one bf16 rounding, and clears them.  copies <= 1: one array. */
/* struct_size = sizeof(NlTouchedRows) of the header the caller was built with: the struct grew in round 5 (copies, copy_stride), and a caller with the
 * shorter one would have its stack read as a copy count - every entry point that takes the struct refuses any other size (NL_ERR_INVALID_ARG).
 * C: NlTouchedRows t = {sizeof t, list, count, flags, copies, copy_stride}. */
typedef struct NlTouchedRows { int struct_size; int* list; int* count; unsigned* flags; int copies; long long copy_stride; } NlTouchedRows;
int nl_trilinear_bwd_t(const void* loss_scalars, const int* s_vox, const float* s_depth, const int* s_ray,
                       const float* rays_d_world, const float* rays_d_sensor, const int* frame_id, const float* poses, int n_frames,
                       const float* centres, const int* vertex_rows, const void* emb_bf16, float voxel_size,
                       const float* dX, float* g_emb, double* g_pose, int nblocks, const NlTouchedRows* touched, void* stream);
int nl_touched_rows_reset(const NlTouchedRows* touched, float* g_emb, void* emb_m_bf16, void* emb_v_bf16, void* stream);

/* masked_scatter_ones (render_helpers.py:30-36,301): packed samples -> padded [R,S] tensors */
int nl_unpack_samples(const void* loss_scalars, const int* s_ray, const int* samp_off, const int* hit_rank,
                      const float* sdf, const float* depth, int S_stride, float* out_sdf, float* out_z, unsigned char* out_valid,
                      void* stream);

/* ---- optimiser: torch.optim.Adam as used in render_helpers.py:341-353,421-423,448-450,508-510 ----
 * `state` = device block of NL_ADAM_STATE_BYTES: int32 step counter (zero it when a new Adam is "constructed",
 * render_helpers.py:353,448) + the per-group hyper-parameters.  nl_adam_prepare advances the step and refreshes the
 * bias corrections on the device (no host scalar => hipGraph-replayable); then the per-group kernels. */
#define NL_ADAM_STATE_BYTES 112
int nl_adam_prepare(int* state, double lr_emb, double lr_dec, double lr_pose, void* stream);
int nl_adam_embeddings(void* emb_bf16, float* g_acc, void* m_bf16, void* v_bf16, long long n_elems, const int* state, void* stream);
int nl_embedding_grad_bf16(const float* g_acc, void* g_bf16, long long n_elems, void* stream);
int nl_adam_f32(float* p, const float* g, float* m, float* v, int n, const int* state, int group /*1 decoder, 2 pose*/, void* stream);
/* se3pose.py:18-35: pose6[F,6] = (t, w) -> poses12[F,12] */
int nl_pose_matrices(const float* pose6, float* poses12, int F, void* stream);
/* Rodrigues tail of the pose gradient + Adam on the 6-vectors (enable[f] != 0) + refreshed matrices */
int nl_pose_step(float* pose6, double* g_pose, float* m, float* v, const int* enable, float* grad6_out, float* poses12,
                 int F, const int* state, int apply, void* stream);

/* The whole optim.step() of one iteration as one launch (plus a one-thread counter advance unless only poses step):
 * nl_adam_prepare + nl_adam_embeddings + nl_adam_f32(decoder) + nl_decoder_transpose_w2 + nl_pose_step, bit for bit.  A group is skipped when its first pointer is NULL
 * (emb_bf16 / dec_params / pose6); dec_ws is the decoder workspace of NL_DEC_WS_FLOATS floats (W2^T + operand planes).
 * skip_mode != 0 (needs the iteration's counter block): the step decides ON THE DEVICE whether the iteration was usable, like the
 * reference's `if final_outputs == None` (render_helpers.py:407-410 mapping: carry on; :486-489 tracking: stop) - unusable = no
 * hit ray, the sampler guard, or a sample-buffer overflow.  An unusable iteration's step clears the gradient accumulators and
 * changes nothing else; state[2] counts such steps, state[3] latches the overflow flag (the host checks both once per call).
 * 1 = skip that step only, 2 = sticky (every step after a skipped one is skipped too). */
int nl_optimiser_step(int* state, double lr_emb, double lr_dec, double lr_pose,
                      void* emb_bf16, float* g_emb, void* emb_m_bf16, void* emb_v_bf16, long long n_emb,
                      float* dec_params, const float* dec_grad, float* dec_m, float* dec_v, float* dec_ws,
                      float* pose6, double* g_pose, float* pose_m, float* pose_v, const int* pose_enable, float* grad6_out,
                      float* poses12, int F, int apply_pose, const int* counters, int skip_mode, void* stream);
/* the same step, ending with the hand-over of the counter block (NL_CNT_BYTES at counters_rw): copied to counters_copy and cleared
 * by the launch's last step, so that the next iteration starts without a memset launch (both NULL: nl_optimiser_step) */
int nl_optimiser_step_ex(int* state, double lr_emb, double lr_dec, double lr_pose,
                         void* emb_bf16, float* g_emb, void* emb_m_bf16, void* emb_v_bf16, long long n_emb,
                         float* dec_params, const float* dec_grad, float* dec_m, float* dec_v, float* dec_ws,
                         float* pose6, double* g_pose, float* pose_m, float* pose_v, const int* pose_enable, float* grad6_out,
                         float* poses12, int F, int apply_pose, const int* counters, int skip_mode, int* counters_rw, int* counters_copy,
                         void* stream);

/* nl_optimiser_step_ex with the embedding group sweeping the touched rows (touched == NULL: the whole table) */
int nl_optimiser_step_t(int* state, double lr_emb, double lr_dec, double lr_pose,
                        void* emb_bf16, float* g_emb, void* emb_m_bf16, void* emb_v_bf16, long long n_emb,
                        float* dec_params, const float* dec_grad, float* dec_m, float* dec_v, float* dec_ws,
                        float* pose6, double* g_pose, float* pose_m, float* pose_v, const int* pose_enable, float* grad6_out,
                        float* poses12, int F, int apply_pose, const int* counters, int skip_mode, int* counters_rw, int* counters_copy,
                        const NlTouchedRows* touched, void* stream);

/* ---- multi-GPU: embedding-gradient exchange over the rows an iteration touches (nerf_loam_amd/dist.py).  Every rank marks the rows
 * of the voxels its rays hit in a zero-filled bitmap of ceil(E / 32) words; after an OR-all-reduce of the bitmap the union's rows
 * are packed in row order into buf[capacity][16] (prefix = exclusive scan of the word popcounts), SUM-all-reduced and unpacked. */
int nl_dist_mark_rows(int N, const int* hit_idx, const int* hit_count, const int* vertex_rows, unsigned* bitmap, void* stream);
int nl_dist_rows_prefix(const unsigned* bitmap, int n_words, int* prefix, int* total, int* workspace /* n_words + ceil(n_words / 1024) + 8 ints */,
                        void* stream);
int nl_dist_rows_move(int direction /* 0 pack, 1 unpack */, const unsigned* bitmap, const int* prefix, int n_words, float* g_emb, float* buf,
                      int capacity, int* fail_word, void* stream);

/* nl_dist_rows_move recording the unpacked rows as touched (a row only other ranks' rays touched carries a gradient here as well) */
int nl_dist_rows_move_t(int direction, const unsigned* bitmap, const int* prefix, int n_words, float* g_emb, float* buf, int capacity,
                        int* fail_word, const NlTouchedRows* touched, void* stream);

/* ---- one call per iteration.  The whole launch sequence of an SDF iteration (render_helpers.py:356-423 mapping / :452-512
 * tracking) issued from C: stages bit 0 = intersect .. backward (everything nl_ray_intersect .. nl_trilinear_bwd above, counter
 * block cleared first), bit 1 = nl_optimiser_step.  It issues the fused forms
 * (nl_ray_intersect_scan, nl_scan_samples_finalize, nl_optimiser_step_ex) of the stage calls.  The descriptor is plain C: device pointers + hyper-parameters; fill it once
 * per optimisation call, change N / seeds / flags between iterations.  Same kernels, same results as the stage-wise calls; what
 * it removes is the host's per-launch cost (~15 launches x ~250 marshalled arguments per iteration from Python). */
struct NlComm;
typedef struct NlIterDesc {
    int struct_size;            /* = sizeof(NlIterDesc) of the header the caller was built with: nl_iteration refuses any other value */
    /* rays of this iteration (sensor frame) and the frames' poses */
    int N, F;
    const float* rays_d_sensor; const float* points_gt; const float* cos_gt; const int* frame_id;
    float* pose6; float* poses12; float* pose_m; float* pose_v; const int* pose_enable; double* g_pose; float* pose_grad6;
    /* map: traversal blocks, reference layouts, embedding table (bf16) */
    const void* blk_hdr; const void* blk_ids; int root_side; float voxel_size;
    const float* centres; const int* vertex_rows; void* emb; long long n_emb_elems;
    /* per-ray and per-sample workspaces (P_cap samples) */
    float* rays_d_world; float* gt_dist; int* hit_idx; float* hit_t0; float* hit_t1; int* hit_count; int* hit_rank; int* ray_of_rank;
    int* samp_count; int* samp_off; int* scan_ws;
    int P_cap; int* s_vox; float* s_depth; float* s_dist; int* s_ray; float* X; float* dX; float* sdf; float* dsdf; unsigned* relu2_mask;
    int* counters; void* loss_scalars; int* adam_state;
    /* decoder block, weight workspace, gradient, Adam moments, per-workgroup slabs */
    float* dec_params; float* dec_ws; float* dec_grad; float* dec_m; float* dec_v; float* partials; int n_slabs, field_blocks;
    /* embedding gradient accumulators and moments (bf16) */
    float* g_emb; void* emb_m; void* emb_v;
    /* hyper-parameters */
    float step_size, max_distance, truncation, sdf_weight, fs_weight;
    double lr_emb, lr_dec, lr_pose;
    unsigned noise_seed; int use_hash_noise, tail_always, ray_id_base, fresh_noise;
    int train_decoder, want_emb_grad, want_pose_grad, update_emb, update_decoder, update_pose, skip_mode;
    /* optional counter hand-over: with counters_copy set, a stages == 3 call ends by copying the counter block there and clearing
     * it; the host sets counters_clean = 1 afterwards and the next call skips its memset launch (0: the block is cleared first) */
    int* counters_copy; int counters_clean;
    /* optional state of the one-launch sampler (nl_sample_rays_fused): >= 8 * (1 + ceil(N / 32)) bytes, zero-initialised once
     * (NULL: the four-launch sampler sequence) */
    void* sample_state;
    /* decoder kernel selection of this descriptor: NL_KERNEL_MODES(gemm_mode, wgrad2_mode); 0 = the process defaults */
    int kernel_modes;
    /* ---- ray-sharded multi-GPU iteration (NULL comm: one GPU).  With a communicator the call issues the exchanges itself, on
     * `stream`, between the kernels (nl_exchange_* below) - a sharded iteration is still ONE C call and stays hipGraph-capturable.
     *   xg_recv     [world][xg_stride] ints: receive side of the counter all-gather after the sampler, xg_stride >= 24 (+ rows_words when
     *               the touched-rows bitmaps travel with it); xg_send [xg_stride] ints: its send side
     *   row_first   [row_first_entries][1 + NL_MAX_HITS] ints (nl_dist_x1_merge)
     *   rows_mode   embedding-gradient exchange: 0 = dense all-reduce of g_emb, 1 = over the rows the iteration touches
     *               (rows_bitmap / rows_prefix [rows_words], rows_total [1], rows_ws [rows_words + ceil(rows_words / 1024) + 8],
     *               rows_buf [rows_cap][16] floats; the call is flagged invalid on the device when the union exceeds rows_cap) */
    const struct NlComm* comm;
    int* xg_send; int* xg_recv; int xg_stride;
    int* row_first; int row_first_entries;
    int rows_mode; unsigned* rows_bitmap; int* rows_prefix; int* rows_total; int* rows_ws; float* rows_buf; int rows_cap, rows_words;
    /* touched-rows optimiser (NlTouchedRows above): with the three pointers set the scatter (and the multi-GPU unpack) record the rows they
     * write; sparse_sweep != 0: the optimiser sweeps the list, 0: the whole table (e.g. after a dense multi-GPU gradient exchange) */
    int* touched_list; int* touched_count; unsigned* touched_flags; int sparse_sweep;
    int touched_copies; long long touched_copy_stride;      /* replicated accumulators (NlTouchedRows.copies / copy_stride); 0 / 1 = one array */
    /* exchange 1 as one all-gather: x1_send [24 ints | x1_rays bytes] (a ray's hit count per byte), x1_recv [world] such blocks of
     * x1_stride_bytes (a multiple of 16, >= 96 + x1_rays); x1_rays = the ranks' common ray capacity (a multiple of 16, >= every rank's N) */
    void* x1_send; void* x1_recv; int x1_stride_bytes; int x1_rays;
    /* overlapped gradient exchange (optional; nl_overlap_create): with the three handles set, a stages & 5 == 5 call runs the embedding
     * scatter first and all-reduces [pose partials | embedding accumulators] on comm_stream under the dW2 kernel and the slab reduction
     * (event fork after the scatter, join before the decoder gradient's all-reduce); NULL: the exchanges follow the backward pass on `stream` */
    void* comm_stream; void* ev_fork; void* ev_join;
    int isect_lanes;            /* lanes per ray of the intersect's work-list: 0 = by ray count, or 4 / 8 / 16 / 32 (nl_ray_intersect_lanes) */
    /* optional hipEvent_t handles (NULL = none) recorded on `stream` in front of the decoder kernel, between it and the dW2 kernel and behind the
     * dW2 kernel: per-kernel durations of an iteration issued as ONE call (bench.py's timed region: the host enqueues a whole step in ~50 us
     * and cannot starve the device); the events are the caller's, created with timing enabled */
    void* ev_decoder_begin; void* ev_decoder_end; void* ev_wgrad2_end;
} NlIterDesc;
/* stages: bit 0 = intersect .. backward (with a communicator: + the exchanges of the forward pass), bit 1 = optimiser step,
 * bit 2 = the gradient exchange (only with a communicator; a whole sharded iteration = 7).  The bits exist separately so that the
 * host can size the touched-rows exchange once per call between the backward pass and the gradient exchange of its first iteration. */
int nl_iteration(const NlIterDesc* desc, int stages, void* stream);

/* ---- communicator of the ray-sharded iteration: four entry points a backend provides.  All of them enqueue on `stream` and return
 * 0 or an error code; buffers are device pointers.  Backends: nl_comm_init_rccl (librccl's ncclAllGather / ncclAllReduce /
 * ncclGroupStart / ncclGroupEnd on an existing ncclComm_t - e.g. the one torch.distributed's ProcessGroupNCCL holds - resolved
 * from the RCCL already loaded in the process), or any set of callbacks (nerf_loam_amd/dist.py: torch.distributed ops, used with
 * non-RCCL process groups and by the virtual-rank tests). */
#define NL_COMM_F32 0
#define NL_COMM_F64 1
#define NL_COMM_I32 2
typedef struct NlComm {
    int world, rank;
    void* ctx;
    int (*all_gather)(void* ctx, const void* send, void* recv, long long bytes_per_rank, void* stream);
    int (*all_reduce_sum)(void* ctx, void* buf, long long count, int dtype, void* stream);      /* in place */
    int (*group_begin)(void* ctx);       /* the all_reduce_sum calls up to group_end may be fused into one launch */
    int (*group_end)(void* ctx);
} NlComm;
/* fills *out for an existing RCCL communicator (ncclComm_t passed as void*); NL_ERR_NO_DEVICE when no RCCL is loaded / loadable */
int nl_comm_init_rccl(NlComm* out, void* nccl_comm, int world, int rank);
/* The exchange points of a sharded iteration, as nl_iteration issues them; also callable between the stage calls:
 *  after_intersect: ONE all-gather of [counter block | a byte per ray: its hit count] -> global hit-ray count, this rank's hit-rank offset,
 *                   global max hits, and the row-first table of the sampler's tail quirk (sample_gpu.cu:224-237 only tests whether
 *                   curr_bin is below the hit count of the first ray of a ray's batch row): nl_dist_x1_pack / nl_dist_x1_merge
 *  after_sampling:  all-gather of [counter block | touched-rows bitmap] -> summed loss normalisers, max samples per ray
 *                   (stage 2), nl_loss_finalize on the merged block, union bitmap + its prefix sums (rows_mode 1)
 *  emb_pose:        grouped SUM all-reduce of the fp64 pose partials (want_pose_grad) and the embedding accumulators (want_emb_grad): dense,
 *                   or the union's rows packed into rows_buf, reduced, unpacked (rows_mode 1)
 *  decoder:         SUM all-reduce of the decoder gradient (train_decoder)
 *  gradients:       emb_pose + decoder on one stream
 *  after_intersect_packed: after_intersect without its pack launch, for a send block nl_ray_intersect_scan_x1 already filled */
int nl_exchange_after_intersect(const NlIterDesc* desc, void* stream);
int nl_exchange_after_intersect_packed(const NlIterDesc* desc, void* stream);
int nl_exchange_after_sampling(const NlIterDesc* desc, void* stream);
int nl_exchange_emb_pose(const NlIterDesc* desc, void* stream);
int nl_exchange_decoder(const NlIterDesc* desc, void* stream);
int nl_exchange_gradients(const NlIterDesc* desc, void* stream);
/* side stream + two timing-less events for NlIterDesc.comm_stream / ev_fork / ev_join: once per engine, off the hot path */
int nl_overlap_create(void** comm_stream, void** ev_fork, void** ev_join);
int nl_overlap_destroy(void* comm_stream, void* ev_fork, void* ev_join);
/* exchange 1: send block [24-int counter block | n_rays_cap bytes: hit count of ray i, 0 beyond N] (n_rays_cap a multiple of 16), and the
 * fold of the gathered blocks [world][stride_bytes] into counters (NLC_R_GLOBAL, NLC_R_OFFSET, global NLC_HMAX) + the row-first table
 * [n_entries][1 + NL_MAX_HITS] = (count, 2 for bins below the count, 0 beyond) that nl_sample_rays reads; n_entries >= 200 *
 * ceil(ceil(R_global / 200) / 800).  Shards are contiguous blocks of the global ray order (rank-major). */
int nl_dist_x1_pack(const int* counters, const int* hit_count, int N, int n_rays_cap, int* send, void* stream);
int nl_dist_x1_merge(const void* gathered, int stride_bytes, int world, int rank, int n_rays_cap, int* counters, int* table, int n_entries, void* stream);
/* exchange 2, receive side: stage 2 of nl_dist_merge_counters_strided + nl_loss_finalize on the merged block in one launch */
int nl_dist_merge_finalize(const int* gathered, int stride_ints, int world, int* counters, void* loss_scalars, float fs_weight, float sdf_weight,
                           float tau, float max_depth, int capacity, void* stream);
/* gathered blocks `stride` ints apart (nl_dist_merge_counters: stride = the block itself) */
int nl_dist_merge_counters_strided(const int* gathered, int stride_ints, int world, int rank, int stage, int* counters, void* stream);
/* union[w] = OR over ranks of gathered[r * stride_ints + offset_ints + w]; then nl_dist_rows_prefix on the union */
int nl_dist_rows_union_prefix(const int* gathered, int stride_ints, int offset_ints, int world, unsigned* union_bitmap, int n_words, int* prefix,
                              int* total, int* workspace, void* stream);

/* ---- (b2) host octree behind torch.classes.svo.Octree (third_party/sparse_octree/src/bindings.cpp:4-31) */
void* nl_octree_create(long long grid_dim);                              /* Octree::init   octree.cpp:36-50   */
void nl_octree_destroy(void* h);
int nl_octree_insert(void* h, const int* voxels_xyz, long long n);       /* Octree::insert octree.cpp:51-111  */
long long nl_octree_count_nodes(void* h);                                /* octree.cpp:344-365 */
long long nl_octree_count_leaf_nodes(void* h);                           /* octree.cpp:367-387 */
int nl_octree_has_voxel(void* h, int x, int y, int z);                   /* octree.cpp:173-206 */
int nl_octree_voxels_dfs(void* h, float* out_xyzs);                      /* get_voxels octree.cpp:242-265: [count_nodes,4], pre-order */
long long nl_octree_leaf_voxels(void* h, float* out_xyz);                /* get_leaf_voxels octree.cpp:212-240; NULL: count */
double nl_octree_try_insert(void* h, const int* voxels_xyz, long long n);  /* octree.cpp:113-149: overlap ratio of the vertex keys */
int nl_octree_export(void* h, float* voxels, float* children, int* features);   /* get_centres_and_children :293-342 */
int nl_octree_export_device_layout(void* h, float voxel_size, float* centres, int* structure, int* vertex_idx); /* + mapping.py:319-327 */
/* incremental export: rows, in the layout above, of the nodes that changed since the previous call */
long long nl_octree_delta_count(void* h);
int nl_octree_export_delta(void* h, float voxel_size, int* ids, float* centres, int* structure, int* vertex_idx);
/* the children-block traversal layout nl_ray_intersect* walks (blk_ids [B][8] int, blk_hdr [B][2] int: csrc/nl_geometry.hip), packed straight from the tree:
 * breadth-first block numbering, FEATURE leaves hidden like in the export's children column, the single-child chain under the root in the pseudo block's
 * spare slots - equal, word for word, to what the Python host layer derives from the exported rows (pipeline.pack_children_blocks), without its ~200 launches
 * per map update.  Returns the number of blocks B; -(B) if capacity < B; with NULL buffers: B (count only). */
long long nl_octree_pack_blocks(void* h, int* blk_ids, int* blk_hdr, long long capacity);

#ifdef __cplusplus
}
#endif
#endif
