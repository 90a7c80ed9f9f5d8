// nl_exchange.cpp -- the exchanges of the ray-sharded multi-GPU iteration, issued from C on the launch stream (host code only).
//
// The reference has no distributed code; north_star: "a scan's rays shard embarrassingly across the 8 GPUs of one node
// with RCCL all-reduce of embedding and pose gradients over xGMI".  Rays are independent up to three reductions:
//   1. after intersect : the sampler's tail quirk depends on a ray's GLOBAL hit rank and on the hit COUNT of the first ray of its
//                        batch row (sample_gpu.cu:224-237 tests pts_idx[curr_bin] == -1 on that ray's packed list) -> ONE
//                        all-gather of [96-byte counter block | a byte per ray: its hit count] (16 KB per rank at the headline size);
//                        every rank derives the row-first table from the gathered counts (nl_dist.hip k_x1_merge)
//   2. after sampling  : criterion.py:84-88 weights and the R*S mean divisor are global -> all-gather of the counter blocks, the
//                        touched-rows bitmaps riding along
//   3. gradients       : a grouped SUM all-reduce of [fp64 pose partials | embedding accumulators, dense or packed touched rows] issued
//                        right after the scatter on a SIDE stream, under the dW2 kernel and the slab reduction (event fork / join inside
//                        nl_iteration), then the decoder gradient's all-reduce (282 KB); every rank applies the identical optimiser step.
// Three collectives on the critical path of an iteration (exchange 1, exchange 2, the decoder gradient) + one hidden under compute; a
// sharded iteration is one C call (nl_iteration) with no host work between its launches, and hipGraph-capturable (RCCL collectives
// and cross-stream event dependencies are).
//
// RCCL binding: the functions are looked up ONLY in an RCCL the process has already loaded (torch ships its own librccl.so and
// ProcessGroupNCCL hands out its ncclComm_t) - dlopen(NULL) / RTLD_NOLOAD on the loaded object: no second RCCL, no second communicator.
#include <dlfcn.h>
#include <link.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <hip/hip_runtime.h>

#include "../../include/nerfloam_hip.h"

namespace {

enum { X_OK = 0, X_ERR_INVALID_ARG = 1, X_ERR_LAUNCH = 2, X_ERR_NO_DEVICE = 3 };
enum { CNT_STRIDE = NL_CNT_INTS + 2 * NL_CNT_DOUBLES };        // ints per counter block (24)

// the subset of rccl.h this file needs (values are NCCL's public ABI: ncclDataType_t, ncclRedOp_t)
typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*group_fn)(void);
enum { RCCL_INT32 = 2, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8, RCCL_INT8 = 0, RCCL_SUM = 0 };

struct Rccl {
    allgather_fn all_gather = nullptr;
    allreduce_fn all_reduce = nullptr;
    group_fn group_start = nullptr, group_end = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::mutex mu;                                                        // (two threads in nl_comm_init_rccl: one look-up at a time)
    std::lock_guard<std::mutex> lock(mu);
    if (r.ok) return r;                                                          // (a failed look-up is retried: RCCL may be loaded later)
    r = [] {
        Rccl x;
        // ONLY the RCCL the process already holds (torch's): the ncclComm_t handed to nl_comm_init_rccl belongs to that build, and a second
        // copy of the library (another soname / version found on the loader path) must never receive it.  Global scope first, then the
        // objects loaded with RTLD_LOCAL (Python extensions), found by walking the loaded-object list and re-opened with RTLD_NOLOAD -
        // which never loads anything.  No loaded RCCL: not ok -> nl_comm_init_rccl returns "no device" and dist.py uses backend "torch".
        void* h = dlopen(nullptr, RTLD_NOW);
        if (!h || !dlsym(h, "ncclAllReduce")) {
            h = nullptr;
            std::string found;
            dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* out) -> int {
                const char* nm = info->dlpi_name;
                if (!nm) return 0;
                const char* base = strrchr(nm, '/');
                base = base ? base + 1 : nm;
                if (strncmp(base, "librccl", 7) == 0 || strncmp(base, "libnccl", 7) == 0) { *(std::string*)out = nm; return 1; }
                return 0;
            }, &found);
            if (!found.empty()) h = dlopen(found.c_str(), RTLD_NOW | RTLD_NOLOAD);
            if (h && !dlsym(h, "ncclAllReduce")) h = nullptr;
        }
        if (!h) return x;
        x.all_gather = (allgather_fn)dlsym(h, "ncclAllGather");
        x.all_reduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
        x.group_start = (group_fn)dlsym(h, "ncclGroupStart");
        x.group_end = (group_fn)dlsym(h, "ncclGroupEnd");
        x.ok = x.all_gather && x.all_reduce && x.group_start && x.group_end;
        return x;
    }();
    return r;
}

int rccl_all_gather(void* ctx, const void* send, void* recv, long long bytes, void* stream)
{
    return rccl().all_gather(send, recv, (size_t)bytes, RCCL_INT8, ctx, (hipStream_t)stream) == 0 ? X_OK : X_ERR_LAUNCH;
}
int rccl_all_reduce_sum(void* ctx, void* buf, long long count, int dtype, void* stream)
{
    const int dt = dtype == NL_COMM_F32 ? RCCL_FLOAT32 : dtype == NL_COMM_F64 ? RCCL_FLOAT64 : dtype == NL_COMM_I32 ? RCCL_INT32 : -1;
    if (dt < 0) return X_ERR_INVALID_ARG;
    return rccl().all_reduce(buf, buf, (size_t)count, dt, RCCL_SUM, ctx, (hipStream_t)stream) == 0 ? X_OK : X_ERR_LAUNCH;
}
int rccl_group_begin(void*) { return rccl().group_start() == 0 ? X_OK : X_ERR_LAUNCH; }
int rccl_group_end(void*) { return rccl().group_end() == 0 ? X_OK : X_ERR_LAUNCH; }

bool comm_ok(const NlComm* c)
{
    return c && c->world >= 1 && c->rank >= 0 && c->rank < c->world && c->all_gather && c->all_reduce_sum && c->group_begin && c->group_end;
}

}  // namespace

#define X_TRY(call) do { const int rc__ = (call); if (rc__ != X_OK) return rc__; } while (0)

extern "C" {

int nl_comm_init_rccl(NlComm* out, void* nccl_comm, int world, int rank)
{
    if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world) return X_ERR_INVALID_ARG;
    if (!rccl().ok) return X_ERR_NO_DEVICE;
    out->world = world; out->rank = rank; out->ctx = nccl_comm;
    out->all_gather = rccl_all_gather; out->all_reduce_sum = rccl_all_reduce_sum;
    out->group_begin = rccl_group_begin; out->group_end = rccl_group_end;
    return X_OK;
}

static int after_intersect(const NlIterDesc* d, void* stream, bool pack)
{
    if (!d || !comm_ok(d->comm) || !d->counters || !d->hit_count || !d->x1_send || !d->x1_recv || d->x1_rays < d->N || (d->x1_rays & 15) ||
        d->x1_stride_bytes < CNT_STRIDE * 4 + d->x1_rays || (d->x1_stride_bytes & 15) || !d->row_first || d->row_first_entries <= 0)
        return X_ERR_INVALID_ARG;
    const NlComm* c = d->comm;
    // ONE all-gather of [counter block | a byte per ray: its hit count]; the row-first table of the sampler's tail quirk follows from the
    // gathered counts on every rank (nl_dist.hip k_x1_merge) - no second collective
    if (pack) X_TRY(nl_dist_x1_pack(d->counters, d->hit_count, d->N, d->x1_rays, (int*)d->x1_send, stream));
    X_TRY(c->all_gather(c->ctx, d->x1_send, d->x1_recv, d->x1_stride_bytes, stream));
    return nl_dist_x1_merge(d->x1_recv, d->x1_stride_bytes, c->world, c->rank, d->x1_rays, d->counters, d->row_first, d->row_first_entries, stream);
}
int nl_exchange_after_intersect(const NlIterDesc* d, void* stream) { return after_intersect(d, stream, true); }
// (the send block was filled by nl_ray_intersect_scan_x1: nl_iteration's sequence)
int nl_exchange_after_intersect_packed(const NlIterDesc* d, void* stream) { return after_intersect(d, stream, false); }

int nl_exchange_after_sampling(const NlIterDesc* d, void* stream)
{
    if (!d || !comm_ok(d->comm) || !d->counters || !d->xg_recv || d->xg_stride < CNT_STRIDE || !d->loss_scalars) return X_ERR_INVALID_ARG;
    const NlComm* c = d->comm;
    const bool rows = d->want_emb_grad && d->rows_mode == 1;
    int stride = CNT_STRIDE;
    if (!rows) {
        X_TRY(c->all_gather(c->ctx, d->counters, d->xg_recv, CNT_STRIDE * 4, stream));
    } else {
        // send block = [counter block | bitmap of the embedding rows this rank's hit voxels reference]
        stride = CNT_STRIDE + d->rows_words;
        if (!d->xg_send || d->xg_stride < stride || (stride & 1) || !d->rows_bitmap || !d->rows_prefix || !d->rows_total || !d->rows_ws || d->rows_words <= 0)
            return X_ERR_INVALID_ARG;
        hipStream_t st = (hipStream_t)stream;
        if (hipMemcpyAsync(d->xg_send, d->counters, CNT_STRIDE * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return X_ERR_LAUNCH;
        if (hipMemsetAsync(d->xg_send + CNT_STRIDE, 0, (size_t)d->rows_words * 4, st) != hipSuccess) return X_ERR_LAUNCH;
        X_TRY(nl_dist_mark_rows(d->N, d->hit_idx, d->hit_count, d->vertex_rows, (unsigned*)(d->xg_send + CNT_STRIDE), stream));
        X_TRY(c->all_gather(c->ctx, d->xg_send, d->xg_recv, (long long)stride * 4, stream));
        X_TRY(nl_dist_rows_union_prefix(d->xg_recv, stride, CNT_STRIDE, c->world, d->rows_bitmap, d->rows_words, d->rows_prefix, d->rows_total,
                                        d->rows_ws, stream));
    }
    // gathered counter blocks -> summed loss normalisers, max samples per ray; the loss scalars of the merged block: one launch
    return nl_dist_merge_finalize(d->xg_recv, stride, c->world, d->counters, d->loss_scalars, d->fs_weight, d->sdf_weight, d->truncation, d->max_distance,
                                  d->P_cap, stream);
}

// embedding accumulators (dense, or the union's rows packed) + fp64 pose partials: one grouped SUM all-reduce
int nl_exchange_emb_pose(const NlIterDesc* d, void* stream)
{
    if (!d || !comm_ok(d->comm)) return X_ERR_INVALID_ARG;
    const NlComm* c = d->comm;
    const bool rows = d->want_emb_grad && d->rows_mode == 1;
    int* fail = d->adam_state ? d->adam_state + 3 : nullptr;             // the latched "call invalid" word (nl_optimiser_step)
    if (d->want_emb_grad && !d->g_emb) return X_ERR_INVALID_ARG;
    if (!d->want_emb_grad && !d->want_pose_grad) return X_OK;
    if (rows) {
        if (!d->rows_buf || d->rows_cap <= 0 || !d->rows_bitmap || !d->rows_prefix) return X_ERR_INVALID_ARG;
        // slots beyond the union are never unpacked, so the buffer needs no clearing
        X_TRY(nl_dist_rows_move(0, d->rows_bitmap, d->rows_prefix, d->rows_words, d->g_emb, d->rows_buf, d->rows_cap, fail, stream));
    }
    X_TRY(c->group_begin(c->ctx));
    int rc = X_OK;
    if (rc == X_OK && d->want_pose_grad) rc = c->all_reduce_sum(c->ctx, d->g_pose, (long long)d->F * 12, NL_COMM_F64, stream);
    if (rc == X_OK && d->want_emb_grad)
        rc = rows ? c->all_reduce_sum(c->ctx, d->rows_buf, (long long)d->rows_cap * NL_EMB_CHANNELS, NL_COMM_F32, stream)
                  : c->all_reduce_sum(c->ctx, d->g_emb, d->n_emb_elems, NL_COMM_F32, stream);
    const int rc_end = c->group_end(c->ctx);
    if (rc != X_OK) return rc;
    if (rc_end != X_OK) return rc_end;
    if (rows) {
        const NlTouchedRows touched = {(int)sizeof(NlTouchedRows), d->touched_list, d->touched_count, d->touched_flags, 1, 0};
        X_TRY(nl_dist_rows_move_t(1, d->rows_bitmap, d->rows_prefix, d->rows_words, d->g_emb, d->rows_buf, d->rows_cap, fail,
                                  d->touched_flags ? &touched : nullptr, stream));
    }
    return X_OK;
}

int nl_exchange_decoder(const NlIterDesc* d, void* stream)
{
    if (!d || !comm_ok(d->comm)) return X_ERR_INVALID_ARG;
    if (!d->train_decoder) return X_OK;
    const NlComm* c = d->comm;
    return c->all_reduce_sum(c->ctx, d->dec_grad, NL_DEC_PARAMS, NL_COMM_F32, stream);
}

// the whole gradient exchange on one stream (the stage-wise hook; the first iteration of a touched-rows call, whose row exchange is sized
// between the backward pass and this call).  nl_iteration otherwise issues the two halves itself, the first one under the dW2 kernel.
int nl_exchange_gradients(const NlIterDesc* d, void* stream)
{
    X_TRY(nl_exchange_emb_pose(d, stream));
    return nl_exchange_decoder(d, stream);
}

/* the side stream + the two events of the overlapped gradient exchange (NlIterDesc.comm_stream / ev_fork / ev_join): created once per
 * engine by the host, never on the hot path */
int nl_overlap_create(void** comm_stream, void** ev_fork, void** ev_join)
{
    if (!comm_stream || !ev_fork || !ev_join) return X_ERR_INVALID_ARG;
    hipStream_t s = nullptr; hipEvent_t a = nullptr, b = nullptr;
    // default priority unless NL_COMM_STREAM_PRIORITY=high (measurement switch: scripts/timeline_probe.py section 7)
    int lo = 0, hi = 0;
    const char* pr = getenv("NL_COMM_STREAM_PRIORITY");
    if (!(pr && pr[0] == 'h') || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) return X_ERR_LAUNCH;
    if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
        if (a) (void)hipEventDestroy(a);
        (void)hipStreamDestroy(s);
        return X_ERR_LAUNCH;
    }
    *comm_stream = s; *ev_fork = a; *ev_join = b;
    return X_OK;
}

int nl_overlap_destroy(void* comm_stream, void* ev_fork, void* ev_join)
{
    if (ev_fork) (void)hipEventDestroy((hipEvent_t)ev_fork);
    if (ev_join) (void)hipEventDestroy((hipEvent_t)ev_join);
    if (comm_stream) (void)hipStreamDestroy((hipStream_t)comm_stream);
    return X_OK;
}

}  // extern "C"
