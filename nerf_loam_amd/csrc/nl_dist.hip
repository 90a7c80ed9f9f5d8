// nl_dist.hip -- device helpers of the ray-sharded multi-GPU iteration (nerf_loam_amd/dist.py): the embedding-gradient exchange
// restricted to the rows an iteration touches.
//
// The reference has no distributed code; north_star asks for "RCCL all-reduce of embedding and pose gradients".  The
// dense accumulator table is 64 B per embedding row - a few MB for one scan, hundreds of MB on a KITTI-scale map -
// while an iteration touches at most rays x hits x 8 rows.  So: every rank marks the rows of the voxels its rays hit in a bitmap
// (E / 8 bytes), the bitmaps are OR-all-reduced, and the union's rows are packed - in row order, via a prefix sum of the word
// popcounts, identically on every rank - into a [capacity, 16] buffer that is SUM-all-reduced and unpacked in place.
#include "nl_common.h"
#include "../../include/nerfloam_hip.h"

// bitmap bit of every embedding row referenced by a hit voxel of this rank's rays (superset of the rows with a gradient)
__global__ void k_mark_touched_rows(int N, const int* __restrict__ hit_idx, const int* __restrict__ hit_count,
                                    const int* __restrict__ vertex_rows, unsigned* __restrict__ bitmap)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;         // 8 lanes per ray: one per voxel corner
    const int ray = t >> 3, k = t & 7;
    if (ray >= N) return;
    const int nh = hit_count[ray];
    for (int l = 0; l < nh; ++l) {
        const int vox = hit_idx[(size_t)ray * NL_MAX_HITS + l];
        if (vox < 0) continue;
        const int row = vertex_rows[8 * (size_t)vox + k];
        const unsigned bit = 1u << (row & 31);
        unsigned* wp = bitmap + (row >> 5);
        if (!(*reinterpret_cast<volatile unsigned*>(wp) & bit)) atomicOr(wp, bit);       // most bits are already set by a neighbour ray
    }
}

__global__ void k_popcount_words(const unsigned* __restrict__ bitmap, int n_words, int* __restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) counts[i] = __popc(bitmap[i]);
}

// union of the ranks' bitmaps as they arrive from the all-gather (RCCL has no bitwise reduction) + the word popcounts
__global__ void k_union_popcount(const int* __restrict__ gathered, int stride, int offset, int world, unsigned* __restrict__ bitmap, int n_words,
                                 int* __restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    unsigned u = 0u;
    for (int r = 0; r < world; ++r) u |= (unsigned)gathered[(size_t)r * stride + offset + i];
    bitmap[i] = u;
    counts[i] = __popc(u);
}

// PACK: rows of the union bitmap, in row order, g_emb[row] -> buf[slot]; UNPACK: buf[slot] -> g_emb[row].  16 lanes per row.
template <bool PACK>
__global__ void k_rows_move(const unsigned* __restrict__ bitmap, const int* __restrict__ prefix, int n_words, float* __restrict__ g_emb,
                            float* __restrict__ buf, int capacity, int* __restrict__ fail_word, NlTouchedDev touched)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int bitpos = t >> 4, c = t & 15;                       // one (row, channel) per thread
    const int wi = bitpos >> 5;
    if (wi >= n_words) return;
    const unsigned word = bitmap[wi], bit = 1u << (bitpos & 31);
    if (!(word & bit)) return;
    const int slot = prefix[wi] + __popc(word & (bit - 1u));
    if (slot >= capacity) { if (c == 0 && fail_word) *fail_word = 1; return; }
    const size_t row = (size_t)bitpos;
    if (PACK) buf[(size_t)slot * NL_C + c] = g_emb[row * NL_C + c];
    else {
        g_emb[row * NL_C + c] = buf[(size_t)slot * NL_C + c];
        if (c == 0) nl_touch_row(touched, (int)row);             // a row only OTHER ranks' rays touched now carries a gradient here as well
    }
}

extern "C" {


/* bitmap[ceil(E / 32)] (zero-filled by the caller) |= rows of the voxels hit by the N rays */
int nl_dist_mark_rows(int N, const int* hit_idx, const int* hit_count, const int* vertex_rows, unsigned* bitmap, void* stream)
{
    if (N <= 0 || !hit_idx || !hit_count || !vertex_rows || !bitmap) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_mark_touched_rows, dim3(nl_div_up((long long)N * 8, 256)), dim3(256), 0, (hipStream_t)stream, N, hit_idx, hit_count,
                       vertex_rows, bitmap);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* prefix[w] = number of set bits in words [0, w), *total = all set bits (device); workspace: n_words + ceil(n_words / 1024) + 8 ints */
int nl_dist_rows_prefix(const unsigned* bitmap, int n_words, int* prefix, int* total, int* workspace, void* stream)
{
    if (!bitmap || n_words <= 0 || !prefix || !total || !workspace) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_popcount_words, dim3(nl_div_up(n_words, 256)), dim3(256), 0, (hipStream_t)stream, bitmap, n_words, workspace);
    NL_LAUNCH_CHECK();
    return nl_exclusive_scan_i32(workspace, prefix, n_words, 0, total, workspace + n_words, stream);
}

int nl_dist_rows_union_prefix(const int* gathered, int stride_ints, int offset_ints, int world, unsigned* union_bitmap, int n_words, int* prefix,
                              int* total, int* workspace, void* stream)
{
    if (!gathered || !union_bitmap || n_words <= 0 || !prefix || !total || !workspace || world <= 0 || offset_ints < 0 ||
        stride_ints < offset_ints + n_words)
        return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_union_popcount, dim3(nl_div_up(n_words, 256)), dim3(256), 0, (hipStream_t)stream, gathered, stride_ints, offset_ints, world,
                       union_bitmap, n_words, workspace);
    NL_LAUNCH_CHECK();
    return nl_exclusive_scan_i32(workspace, prefix, n_words, 0, total, workspace + n_words, stream);
}

/* direction 0: g_emb rows of the bitmap -> buf[capacity][16] (row order); 1: back.  *fail_word = 1 if the rows exceed capacity. */
int nl_dist_rows_move_t(int direction, const unsigned* bitmap, const int* prefix, int n_words, float* g_emb, float* buf, int capacity,
                        int* fail_word, const NlTouchedRows* touched, void* stream)
{
    if (!bitmap || !prefix || n_words <= 0 || !g_emb || !buf || capacity <= 0) return NL_ERR_INVALID_ARG;
    NlTouchedDev t = {nullptr, nullptr, nullptr};
    if (touched && touched->flags && touched->list && touched->count) { t.list = touched->list; t.count = touched->count; t.flags = touched->flags; }
    const int nb = nl_div_up((long long)n_words * 32 * 16, 256);
    if (direction == 0) hipLaunchKernelGGL(k_rows_move<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, bitmap, prefix, n_words, g_emb, buf, capacity, fail_word, t);
    else                hipLaunchKernelGGL(k_rows_move<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, bitmap, prefix, n_words, g_emb, buf, capacity, fail_word, t);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_dist_rows_move(int direction, const unsigned* bitmap, const int* prefix, int n_words, float* g_emb, float* buf, int capacity,
                      int* fail_word, void* stream)
{
    return nl_dist_rows_move_t(direction, bitmap, prefix, n_words, g_emb, buf, capacity, fail_word, nullptr, stream);
}

}  // extern "C"
