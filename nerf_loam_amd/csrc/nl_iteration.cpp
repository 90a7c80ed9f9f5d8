// nl_iteration.cpp -- one C call per iteration: the whole launch sequence of SdfEngine.forward_backward + optimiser_step
// (/root/reference/src/variations/render_helpers.py:356-423 mapping, :452-512 tracking) issued from C through the same nl_* entry
// points the stage-wise binding uses.  Why: at the reference's live shapes (2048 rays) an iteration is ~15 kernels of 5-50 us; a
// Python / ctypes launch costs ~10 us (250 pointer arguments per iteration), so the host, not the GPU, set the pace of
// bundle_adjust_frames / track_frame.  The descriptor holds every pointer and hyper-parameter; the host fills it once per call and
// flips a few scalars per iteration.  Host code only (no kernels here).
#include <hip/hip_runtime.h>

#include "../../include/nerfloam_hip.h"

// counter-block slots used here (nl_common.h: NLC_R, NLC_P, NLC_R_GLOBAL; the block is NL_CNT_INTS ints + NL_CNT_DOUBLES doubles)
enum { IT_R = 0, IT_R_GLOBAL = 13, IT_OK = 0, IT_ERR_INVALID_ARG = 1, IT_ERR_LAUNCH = 2 };

extern "C" int nl_iteration(const NlIterDesc* d, int stages, void* stream)
{
    // struct_size: a caller built against another revision of the header (the descriptor has grown every round) is refused instead of
    // having its shorter struct read past the end
    if (!d || d->struct_size != (int)sizeof(NlIterDesc) || d->N <= 0 || d->F <= 0 || !(stages & 7)) return IT_ERR_INVALID_ARG;
    const bool sharded = d->comm != nullptr;                    // ray-sharded multi-GPU iteration: the exchanges of nl_exchange.cpp ride along
    hipStream_t st = (hipStream_t)stream;
    const NlTouchedRows touched_rows = {(int)sizeof(NlTouchedRows), d->touched_list, d->touched_count, d->touched_flags, d->touched_copies, d->touched_copy_stride};
    const NlTouchedRows* touched = d->touched_flags ? &touched_rows : nullptr;      // rows written by the scatter are recorded
    // decoder workgroup layout by ray count where the descriptor leaves it open (csrc/nl_common.h NL_RAYS_DECODER_SPLIT)
    const int kernel_modes = (d->kernel_modes & NL_KERNEL_LAYOUT(3)) ? d->kernel_modes : (d->kernel_modes | NL_KERNEL_LAYOUT(nl_decoder_layout_for(d->N)));
    int rc = IT_OK;
    bool overlapped = false, forked = false;
    // an error after the side stream was forked still JOINS it before returning: a capture in progress is not left with a dangling
    // stream, and later work on `stream` cannot race the all-reduce that may be in flight on the accumulators
    auto leave = [&](int code) {
        if (forked) {
            (void)hipEventRecord((hipEvent_t)d->ev_join, (hipStream_t)d->comm_stream);
            (void)hipStreamWaitEvent(st, (hipEvent_t)d->ev_join, 0);
            forked = false;
        }
        return code;
    };
#define NL_TRY(call) do { rc = (call); if (rc != IT_OK) return leave(rc); } while (0)
    if (stages & 1) {
        int* c = d->counters;
        // the counter block starts an iteration zeroed: by the previous iteration's last kernel when that handed it over
        // (counters_copy + counters_clean, set by the host after a stages == 3 call), by a memset launch otherwise
        if (!(d->counters_copy && d->counters_clean) &&
            hipMemsetAsync(c, 0, NL_CNT_INTS * 4 + NL_CNT_DOUBLES * 8, st) != hipSuccess) return IT_ERR_LAUNCH;
        const bool x1_fused = sharded && d->x1_send && d->x1_rays >= d->N && !(d->x1_rays & 15);     // (a bad block is reported by the exchange below)
        if (x1_fused)
            NL_TRY(nl_ray_intersect_scan_x1(d->N, d->rays_d_sensor, d->points_gt, d->cos_gt, d->frame_id, d->poses12, d->blk_hdr, d->blk_ids, d->root_side,
                                            d->voxel_size, d->max_distance, d->rays_d_world, d->gt_dist, d->hit_idx, d->hit_t0, d->hit_t1, d->hit_count, c,
                                            d->ray_of_rank, d->hit_rank, c + IT_R, c + IT_R_GLOBAL, d->scan_ws, d->isect_lanes, (int*)d->x1_send, d->x1_rays,
                                            stream));
        else
        NL_TRY(nl_ray_intersect_scan_lanes(d->N, d->rays_d_sensor, d->points_gt, d->cos_gt, d->frame_id, d->poses12, d->blk_hdr, d->blk_ids, d->root_side,
                                           d->voxel_size, d->max_distance, d->rays_d_world, d->gt_dist, d->hit_idx, d->hit_t0, d->hit_t1, d->hit_count, c,
                                           d->ray_of_rank, d->hit_rank, c + IT_R, c + IT_R_GLOBAL, d->scan_ws, d->isect_lanes, stream));
        const unsigned* mix = d->fresh_noise ? (const unsigned*)d->adam_state : nullptr;
        if (sharded) {
            // exchange 1 -> global hit ranks + the batch rows' first-ray hit lists; count, scan, emit on the local rays; exchange 2 ->
            // global loss normalisers (+ the union of the touched embedding rows)
            NL_TRY(x1_fused ? nl_exchange_after_intersect_packed(d, stream) : nl_exchange_after_intersect(d, stream));
            for (int emit = 0; emit < 2; ++emit) {
                NL_TRY(nl_sample_rays(emit, d->N, d->hit_idx, d->hit_t0, d->hit_t1, d->hit_count, d->hit_rank, d->ray_of_rank, d->cos_gt, d->gt_dist,
                                      d->step_size, d->truncation, d->max_distance, d->noise_seed, d->use_hash_noise, d->tail_always, d->ray_id_base,
                                      mix, d->row_first, c, d->samp_count, emit ? d->samp_off : nullptr, d->P_cap, emit ? d->s_vox : nullptr,
                                      emit ? d->s_depth : nullptr, emit ? d->s_dist : nullptr, emit ? d->s_ray : nullptr, stream));
                if (!emit) NL_TRY(nl_exclusive_scan_i32(d->samp_count, d->samp_off, d->N, 0, c + 3 /* NLC_P */, d->scan_ws, stream));
            }
            NL_TRY(nl_exchange_after_sampling(d, stream));
        } else
        // count pass + offset scan + loss normalisers + emit pass: one launch up to 8192 rays (needs sample_state), four beyond
        NL_TRY(nl_sample_rays_fused(d->N, d->hit_idx, d->hit_t0, d->hit_t1, d->hit_count, d->hit_rank, d->ray_of_rank, d->cos_gt, d->gt_dist,
                                    d->step_size, d->truncation, d->max_distance, d->noise_seed, d->use_hash_noise, d->tail_always, d->ray_id_base, mix,
                                    c, d->samp_count, d->samp_off, d->P_cap, d->s_vox, d->s_depth, d->s_dist, d->s_ray, d->loss_scalars, d->fs_weight,
                                    d->sdf_weight, d->sample_state, d->scan_ws, stream));
        NL_TRY(nl_gather_trilinear(d->loss_scalars, d->s_vox, d->s_depth, d->s_ray, d->rays_d_world, d->frame_id, d->poses12, d->F, d->centres,
                                   d->vertex_rows, d->emb, d->voxel_size, d->X, d->field_blocks, stream));
        if (d->ev_decoder_begin && hipEventRecord((hipEvent_t)d->ev_decoder_begin, st) != hipSuccess) return leave(IT_ERR_LAUNCH);
        NL_TRY(nl_decoder_fwd_bwd_m(d->loss_scalars, d->X, d->dec_params, d->dec_ws, d->s_ray, d->s_depth, d->cos_gt, d->gt_dist, d->sdf, d->dsdf,
                                    d->dX, d->partials, d->relu2_mask, d->n_slabs, d->train_decoder, c, kernel_modes, stream));
        if (d->ev_decoder_end && hipEventRecord((hipEvent_t)d->ev_decoder_end, st) != hipSuccess) return leave(IT_ERR_LAUNCH);
        // ray-sharded with the gradient exchange in the same call: the embedding scatter goes FIRST and its all-reduce (the large message:
        // 64 B per embedding row or per touched row, + the pose partials) leaves on the side stream while dW2 and the slab reduction run
        // ("overlap the embedding-grad reduce with the decoder wgrad"); the decoder gradient's all-reduce follows the join.
        // Same kernels on the same data as the serial order - the two halves touch disjoint buffers - so the results are those of the
        // serial order bit for bit.
        overlapped = sharded && (stages & 4) && d->comm_stream && d->ev_fork && d->ev_join;
        if (overlapped) {
            hipStream_t cs = (hipStream_t)d->comm_stream;
            NL_TRY(nl_trilinear_bwd_t(d->loss_scalars, d->s_vox, d->s_depth, d->s_ray, d->rays_d_world, d->rays_d_sensor, d->frame_id, d->poses12, d->F,
                                      d->centres, d->vertex_rows, d->emb, d->voxel_size, d->dX, d->want_emb_grad ? d->g_emb : nullptr,
                                      d->want_pose_grad ? d->g_pose : nullptr, 2 * d->field_blocks, touched, stream));
            if (hipEventRecord((hipEvent_t)d->ev_fork, st) != hipSuccess || hipStreamWaitEvent(cs, (hipEvent_t)d->ev_fork, 0) != hipSuccess) return IT_ERR_LAUNCH;
            forked = true;
            NL_TRY(nl_exchange_emb_pose(d, d->comm_stream));
            if (hipEventRecord((hipEvent_t)d->ev_join, cs) != hipSuccess) return leave(IT_ERR_LAUNCH);
        }
        if (d->train_decoder) {
            // (timing events: the caller records ev_decoder_end .. ev_wgrad2_end around the dW2 kernel only where nothing else runs between them -
            //  the overlapped sharded order puts the scatter there, and bench.py times that order without them)
            NL_TRY(nl_decoder_wgrad2_m(d->loss_scalars, d->X, d->dec_params, d->dsdf, d->relu2_mask, d->partials, d->n_slabs, kernel_modes, stream));
            if (d->ev_wgrad2_end && hipEventRecord((hipEvent_t)d->ev_wgrad2_end, st) != hipSuccess) return leave(IT_ERR_LAUNCH);
            NL_TRY(nl_decoder_reduce_m(d->partials, d->n_slabs, d->dec_params, d->dec_grad, kernel_modes, stream));
        }
        if (overlapped) {
            if (hipStreamWaitEvent(st, (hipEvent_t)d->ev_join, 0) != hipSuccess) return leave(IT_ERR_LAUNCH);
            forked = false;
            NL_TRY(nl_exchange_decoder(d, stream));
        } else
        NL_TRY(nl_trilinear_bwd_t(d->loss_scalars, d->s_vox, d->s_depth, d->s_ray, d->rays_d_world, d->rays_d_sensor, d->frame_id, d->poses12, d->F,
                                  d->centres, d->vertex_rows, d->emb, d->voxel_size, d->dX, d->want_emb_grad ? d->g_emb : nullptr,
                                  d->want_pose_grad ? d->g_pose : nullptr, 2 * d->field_blocks, touched, stream));
    }
    if ((stages & 4) && sharded && !overlapped) NL_TRY(nl_exchange_gradients(d, stream));
    if (stages & 2) {
        const bool hand_over = d->counters_copy && (stages & 1);       // only a whole iteration leaves the block to the next one
        NL_TRY(nl_optimiser_step_t(d->adam_state, d->lr_emb, d->lr_dec, d->lr_pose,
                                    d->update_emb ? d->emb : nullptr, d->g_emb, d->emb_m, d->emb_v, d->n_emb_elems,
                                    d->update_decoder ? d->dec_params : nullptr, d->dec_grad, d->dec_m, d->dec_v, d->dec_ws,
                                    d->pose6, d->g_pose, d->pose_m, d->pose_v, d->pose_enable, d->pose_grad6, d->poses12, d->F, d->update_pose,
                                    d->skip_mode ? d->counters : nullptr, d->skip_mode, hand_over ? d->counters : nullptr,
                                    hand_over ? d->counters_copy : nullptr, d->sparse_sweep ? touched : nullptr, stream));
    }
#undef NL_TRY
    return IT_OK;
}
