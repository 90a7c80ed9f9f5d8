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

// ---------------------------------------------------------------------------------------------
// Exchange 1 as ONE all-gather (after the intersect).  What the sampler needs from other ranks: the global number of hit rays, this
// rank's hit-rank offset, the global max hits - and, for the reference's tail quirk (sample_gpu.cu:224-237), the hit list of
// the FIRST ray of every batch row of the [200, L] layout.  The closing loop only tests `pts_idx[curr_bin] == -1` on that list, i.e.
// whether curr_bin is below the row-first ray's HIT COUNT (sorted / culled lists are packed, -1 beyond the count).  So every rank sends
// [counter block | one byte per ray: its hit count]; with the gathered counts every rank finds the row-first rays itself (the global hit
// rank of a ray is a prefix sum over the gathered bytes: shards are contiguous blocks of the global ray order) and fills the table the
// sampler kernels read: row e = (count, then 2 for bins below the count, 0 beyond - the kernels compare entry - 1 with -1).
// ---------------------------------------------------------------------------------------------
__global__ void k_x1_pack(const int* __restrict__ counters, const int* __restrict__ hit_count, int N, int n_rays_cap, int* __restrict__ send)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < NL_CNT_INTS + 2 * NL_CNT_DOUBLES) send[t] = counters[t];
    if (4 * t < n_rays_cap) {
        unsigned w = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * t + j;
            const int c = i < N ? hit_count[i] : 0;
            w |= (unsigned)(c < 0 ? 0 : (c > 255 ? 255 : c)) << (8 * j);
        }
        reinterpret_cast<unsigned*>(send + NL_CNT_INTS + 2 * NL_CNT_DOUBLES)[t] = w;
    }
}

#define X1_THREADS 1024
__device__ __forceinline__ void x1_write_row(int* __restrict__ table, int e, int cnt)
{
    int* row = table + (size_t)e * (1 + NL_MAX_HITS);
    row[0] = cnt;
#pragma unroll
    for (int b = 0; b < NL_MAX_HITS; ++b) row[1 + b] = b < cnt ? 2 : 0;
}

// one workgroup per rank's slice of the gathered buffer; every rank runs all of them and ends with the whole table
__global__ __launch_bounds__(X1_THREADS) void k_x1_merge(const unsigned char* __restrict__ recv, int stride_bytes, int world, int rank, int n_rays_cap,
                                                         int* __restrict__ counters, int* __restrict__ table, int n_entries)
{
    __shared__ int s_wave[X1_THREADS / 64];
    __shared__ int s_cnt0;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int Rg = 0, off_r = 0, off_me = 0, hmax = 0;
    for (int q = 0; q < world; ++q) {
        const int* blk = reinterpret_cast<const int*>(recv + (size_t)q * stride_bytes);
        const int v = blk[NLC_R];
        Rg += v; if (q < r) off_r += v; if (q < rank) off_me += v;
        hmax = max(hmax, blk[NLC_HMAX]);
    }
    if (r == 0 && tid == 0) { counters[NLC_R_GLOBAL] = Rg; counters[NLC_R_OFFSET] = off_me; counters[NLC_HMAX] = hmax; }
    const int L = (Rg + NL_SAMPLER_G - 1) / NL_SAMPLER_G;
    const int nch = L > 0 ? (L + NL_SAMPLER_CHUNK - 1) / NL_SAMPLER_CHUNK : 1;
    const int used = NL_SAMPLER_G * nch < n_entries ? NL_SAMPLER_G * nch : n_entries;
    if (r == 0) for (int e = (L > 0 ? used : 0) + tid; e < n_entries; e += X1_THREADS) x1_write_row(table, e, 0);    // entries no row uses
    if (tid == 0) s_cnt0 = -1;
    if (L == 0) return;
    // a thread's 16 rays = one 16-byte load (n_rays_cap and the block stride are multiples of 16: the send block is laid out that way)
    const uint4* words = reinterpret_cast<const uint4*>(recv + (size_t)r * stride_bytes + (NL_CNT_INTS + 2 * NL_CNT_DOUBLES) * 4);
    const int chunks = n_rays_cap >> 4;
    int q_block = off_r;                                          // hit rays of this slice in front of the current pass
    for (int c0 = 0; c0 < chunks; c0 += X1_THREADS) {
        const int ci = c0 + tid;
        const uint4 v = ci < chunks ? words[ci] : make_uint4(0u, 0u, 0u, 0u);
        const unsigned wd[4] = {v.x, v.y, v.z, v.w};
        int mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) mine += ((wd[j] >> (8 * k)) & 255u) ? 1 : 0;
        int incl = mine;                                          // inclusive scan over the workgroup: wave shuffles, then the wave totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        __syncthreads();                                          // (the previous pass has read s_wave)
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < X1_THREADS / 64; ++k) { const int u = s_wave[k]; if (k < w) before += u; total += u; }
        const int q = q_block + before + incl - mine;             // global hit rank of this thread's first hit ray
        // (batch row, position in the row) of that ray: ONE division per thread and pass, then counted up (sixteen `q / L`, `q % L` pairs by a
        // run-time divisor were most of this kernel's 10 us on a rank's share)
        const int Ls = L > 0 ? L : 1;                             // (no hit ray on any rank: L == 0, nothing below is used - but no division by zero either)
        int qrow = q / Ls, within = q - qrow * Ls;
        bool first = q == 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = (int)((wd[j] >> (8 * k)) & 255u);
                if (!c) continue;
                if (first) { s_cnt0 = c; first = false; }
                if (within % NL_SAMPLER_CHUNK == 0) {
                    const int e = qrow * nch + within / NL_SAMPLER_CHUNK;
                    if (e < n_entries) x1_write_row(table, e, c);
                }
                if (++within == L) { within = 0; ++qrow; }
            }
        q_block += total;
    }
    __syncthreads();
    const int cnt0 = s_cnt0;                                      // this slice holds hit-ray 0: the padding rows replicate it (voxel_helpers.py:278-284)
    if (cnt0 >= 0)
        for (int e = tid; e < used; e += X1_THREADS) {
            const int first = (e / nch) * L + (e % nch) * NL_SAMPLER_CHUNK;
            if (first >= Rg) x1_write_row(table, e, cnt0);
        }
}

extern "C" {

/* exchange 1, send side: send = [counter block (24 ints) | n_rays_cap bytes: hit count of ray i, 0 beyond N]; n_rays_cap a multiple of 16 */
int nl_dist_x1_pack(const int* counters, const int* hit_count, int N, int n_rays_cap, int* send, void* stream)
{
    if (!counters || !hit_count || !send || N < 0 || n_rays_cap < N || (n_rays_cap & 15)) return NL_ERR_INVALID_ARG;
    const int threads = n_rays_cap / 4 > NL_CNT_INTS + 2 * NL_CNT_DOUBLES ? n_rays_cap / 4 : NL_CNT_INTS + 2 * NL_CNT_DOUBLES;
    hipLaunchKernelGGL(k_x1_pack, dim3(nl_div_up(threads, 256)), dim3(256), 0, (hipStream_t)stream, counters, hit_count, N, n_rays_cap, send);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* exchange 1, receive side: gathered [world][stride_bytes] -> counters (NLC_R_GLOBAL, NLC_R_OFFSET, global NLC_HMAX) + the row-first
 * table [n_entries][1 + NL_MAX_HITS] of nl_sample_rays (n_entries >= 200 * ceil(ceil(R_global / 200) / 800)) */
int nl_dist_x1_merge(const void* gathered, int stride_bytes, int world, int rank, int n_rays_cap, int* counters, int* table, int n_entries, void* stream)
{
    if (!gathered || !counters || !table || world <= 0 || rank < 0 || rank >= world || n_rays_cap <= 0 || n_entries <= 0 ||
        stride_bytes < (NL_CNT_INTS + 2 * NL_CNT_DOUBLES) * 4 + n_rays_cap || (stride_bytes & 15) || (n_rays_cap & 15))
        return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_x1_merge, dim3(world), dim3(X1_THREADS), 0, (hipStream_t)stream, (const unsigned char*)gathered, stride_bytes, world, rank,
                       n_rays_cap, counters, table, n_entries);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

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
    if (touched && touched->struct_size != (int)sizeof(NlTouchedRows)) return NL_ERR_INVALID_ARG;
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
