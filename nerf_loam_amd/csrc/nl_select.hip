// nl_select.hip -- on-device ray selection.
//
// Reference behaviour: LidarFrame.sample_rays -> sample_util.sampling_without_replacement
// (src/lidarFrame.py:55-57, src/utils/sample_util.py:4-19): Gumbel-top-k over uniform scores = a uniformly random subset of
// N_rays of the frame's M returns, returned as a boolean mask so the chosen rays stay in dataset order.  The reference draws
// the noise with torch's CPU generator every iteration and uploads the mask; its exact random stream is not reproducible
// on another backend, so parity here is distributional: exactly N distinct rays, dataset order, every ray
// equally likely.
//
// Device formulation: key_i = mix32(i ^ seed') with mix32 a BIJECTION on 32-bit integers, so the keys of distinct rays are
// distinct and "the N largest keys" is a well-defined subset with no ties.  The N-th largest key T is found by a 4-pass
// radix select (256-bin histograms, one tiny pick kernel per pass), then flags = key >= T, an exclusive scan and a gather
// of (direction, point, cos) into the engine's ray buffers.  No host round trip: the whole selection is ~10 short
// launches on the stream and is hipGraph-capturable.
#include "nl_common.h"

extern "C" int nl_exclusive_scan_i32(const int* in, int* out, int n, int flag_mode, int* total_out, int* workspace, void* stream);

// state ints: [0] prefix of T found so far, [1] how many keys still to take below the prefix, [2] number selected (out)
__global__ void k_sel_init(int* state, int* hist, int n_select)
{
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { state[0] = 0; state[1] = n_select; state[2] = 0; }
}

__global__ void k_sel_hist(int M, unsigned seed, int pass, const int* __restrict__ state, int* __restrict__ hist)
{
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 8 * pass;
    const unsigned prefix = (unsigned)state[0];
    const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const unsigned key = nl_select_key(seed, (unsigned)i);
        if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&h[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// digit of T at this pass: the largest d with count(keys sharing the prefix and digit >= d) >= remaining
__global__ void k_sel_pick(int pass, int* state, int* hist)
{
    if (threadIdx.x == 0) {
        int remaining = state[1];
        int d = 255, above = 0;
        for (; d > 0; --d) {
            if (above + hist[d] >= remaining) break;
            above += hist[d];
        }
        state[0] = (int)((unsigned)state[0] | ((unsigned)d << (8 * pass)));
        state[1] = remaining - above;
    }
    __syncthreads();
    hist[threadIdx.x] = 0;
}

__global__ void k_sel_flags(int M, unsigned seed, int n_select, const int* __restrict__ state, int* __restrict__ flags, unsigned char* __restrict__ mask)
{
    const unsigned T = n_select >= M ? 0u : (unsigned)state[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const int f = nl_select_key(seed, (unsigned)i) >= T ? 1 : 0;
        flags[i] = f;
        if (mask) mask[i] = (unsigned char)f;
    }
}

// one selected return -> the engine's ray buffers.  rays_d == NULL: the unit direction is computed here from the point
// (LidarFrame.get_rays, src/lidarFrame.py:47-52 - nl_unit_dir), so a frame is resident as points + cos only and no direction array
// is ever built, uploaded or read.
__device__ __forceinline__ void sel_emit(const float* __restrict__ rays_d, const float* __restrict__ points, const float* __restrict__ cos_in, size_t i,
                                         int frame, float* __restrict__ out_d, float* __restrict__ out_p, float* __restrict__ out_cos,
                                         int* __restrict__ out_frame, size_t o)
{
    const float px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
    float dx, dy, dz;
    if (rays_d) { dx = rays_d[3 * i]; dy = rays_d[3 * i + 1]; dz = rays_d[3 * i + 2]; }
    else nl_unit_dir(px, py, pz, &dx, &dy, &dz);
    out_d[3 * o] = dx; out_d[3 * o + 1] = dy; out_d[3 * o + 2] = dz;
    out_p[3 * o] = px; out_p[3 * o + 1] = py; out_p[3 * o + 2] = pz;
    out_cos[o] = cos_in[i];
    if (out_frame) out_frame[o] = frame;
}

// LidarFrame.get_rays for a whole scan (src/lidarFrame.py:47-52): rays_d [M,3] (and rays_norm [M], optional) from points [M,3]
__global__ void k_unit_dirs(int M, const float* __restrict__ points, float* __restrict__ out_d, float* __restrict__ out_norm)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        float dx, dy, dz;
        const float n = nl_unit_dir(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2], &dx, &dy, &dz);
        out_d[3 * (size_t)i] = dx; out_d[3 * (size_t)i + 1] = dy; out_d[3 * (size_t)i + 2] = dz;
        if (out_norm) out_norm[i] = n;
    }
}

__global__ void k_sel_gather(int M, const int* __restrict__ flags, const int* __restrict__ rank, const float* __restrict__ rays_d,
                             const float* __restrict__ points, const float* __restrict__ cos_in, int frame, int cap,
                             float* __restrict__ out_d, float* __restrict__ out_p, float* __restrict__ out_cos, int* __restrict__ out_frame)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        if (!flags[i]) continue;
        const int o = rank[i];
        if (o >= cap) continue;
        sel_emit(rays_d, points, cos_in, (size_t)i, frame, out_d, out_p, out_cos, out_frame, (size_t)o);
    }
}

// ---------------------------------------------------------------------------------------------
// Two-launch selection for n << M (the live shapes: 2048 / 4096 of ~131 072 returns per frame, several frames per call).
// The keys are uniform, so the n-th largest one is known in advance up to the fluctuation of a binomial count: T0 = 2^32 (1 - n/M),
// +- a window of W = 16 sigma + 64 keys' worth, sigma = sqrt(n (1 - n/M)).  Pass A classifies every key - above the window: surely
// selected (counted per wave, the waves own contiguous index ranges), inside it: a candidate (appended to a small list, a few
// hundred to a few thousand keys) - and pass B finds the exact threshold among the candidates (radix select in LDS, every
// workgroup for itself), derives each wave's output rank base from the per-wave counts and the candidates below its range, and
// emits the selected rays in dataset order.  Same subset as the radix path: the n largest keys, exactly.  All frames of a call
// go through the same two launches (grid.y = frame).  If a count ever fell outside the window (probability ~1e-50) the fail
// word is set and the caller falls back to nl_select_rays.
// ---------------------------------------------------------------------------------------------
#define SEL_MAX_FRAMES 8
#define SEL_CAP 4096                                    // candidates per frame (LDS: 16 KB keys + 16 KB indices)
#define SEL_MAXB 128                                    // workgroups per frame
#define SEL_WS_INTS_PER_FRAME (8 + 4 * SEL_MAXB + 2 * SEL_CAP)      // [cand_n[2], fail, pad..][sure per wave][cand keys][cand idx]

struct SelFrame {
    const float* d; const float* p; const float* c; unsigned char* mask;
    int M, n, out_off, frame; unsigned seed, lo, hi; int nblk, wchunk;
};
struct SelArgs { SelFrame f[SEL_MAX_FRAMES]; int* ws; float* out_d; float* out_p; float* out_c; int* out_frame; int parity; int* fail_word; };

__global__ __launch_bounds__(256) void k_sel_window_a(SelArgs a)
{
    const SelFrame fr = a.f[blockIdx.y];
    if ((int)blockIdx.x >= fr.nblk) return;
    int* ws = a.ws + (size_t)blockIdx.y * SEL_WS_INTS_PER_FRAME;
    // candidates are collected per workgroup in LDS and appended with ONE global atomic per workgroup (a same-address device atomic
    // costs ~12 ns: two thousand of them, one per candidate, would be most of this kernel's time)
    __shared__ int s_ck[256], s_ci[256], s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int wstart = gw * fr.wchunk, wend = min(fr.M, wstart + fr.wchunk);
    int sure = 0;
    for (int i0 = wstart; i0 < wend; i0 += 64) {
        const int i = i0 + lane;
        const unsigned key = i < wend ? nl_select_key(fr.seed, (unsigned)i) : 0u;
        const bool live = i < wend;
        sure += __popcll(__ballot(live && key > fr.hi));
        if (live && key >= fr.lo && key <= fr.hi) {
            const int q = atomicAdd(&s_cnt, 1);
            if (q < 256) { s_ck[q] = (int)key; s_ci[q] = i; }
            else {                                               // a workgroup with > 256 candidates (expected ~16): straight to memory
                const int pos = atomicAdd(&ws[a.parity], 1);
                if (pos < SEL_CAP) { ws[8 + 4 * SEL_MAXB + pos] = (int)key; ws[8 + 4 * SEL_MAXB + SEL_CAP + pos] = i; }
            }
        }
    }
    if (lane == 0) ws[8 + gw] = sure;
    __syncthreads();
    const int cnt = min(s_cnt, 256);
    if (threadIdx.x == 0 && cnt > 0) s_base = atomicAdd(&ws[a.parity], cnt);
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
        const int pos = s_base + threadIdx.x;
        if (pos < SEL_CAP) { ws[8 + 4 * SEL_MAXB + pos] = s_ck[threadIdx.x]; ws[8 + 4 * SEL_MAXB + SEL_CAP + pos] = s_ci[threadIdx.x]; }
    }
}

__global__ __launch_bounds__(256) void k_sel_window_b(SelArgs a)
{
    __shared__ unsigned s_key[SEL_CAP];
    __shared__ int s_idx[SEL_CAP];
    __shared__ int s_hist[256];
    __shared__ int s_red[8];
    __shared__ unsigned s_T;
    __shared__ int s_take;
    const SelFrame fr = a.f[blockIdx.y];
    if ((int)blockIdx.x >= fr.nblk) return;
    int* ws = a.ws + (size_t)blockIdx.y * SEL_WS_INTS_PER_FRAME;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, gw = blockIdx.x * 4 + w;
    const int cand_n = ws[a.parity];
    const int C = min(cand_n, SEL_CAP);
    if (blockIdx.x == 0 && tid == 0) ws[a.parity ^ 1] = 0;       // the other slot: last read by the previous call's pass B, used by the next call
    // candidates relative to the window's lower edge, scaled so that the window fills the 32-bit range: the raw keys of a window
    // share their leading byte(s), and the first radix pass would serialise every candidate on two or three LDS bins
    const int sh = __clz((int)((fr.hi - fr.lo) | 1u));
    for (int i = tid; i < C; i += 256) { s_key[i] = ((unsigned)ws[8 + 4 * SEL_MAXB + i] - fr.lo) << sh; s_idx[i] = ws[8 + 4 * SEL_MAXB + SEL_CAP + i]; }
    // surely selected keys: all of them, and those in the waves before this one
    int tot = 0, before = 0;
    for (int j = tid; j < 4 * fr.nblk; j += 256) { const int v = ws[8 + j]; tot += v; }
    for (int j = lane; j < gw; j += 64) before += ws[8 + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { tot += __shfl_xor(tot, off); before += __shfl_xor(before, off); }
    if (lane == 0) s_red[w] = tot;
    __syncthreads();
    const int total_sure = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const int k = fr.n - total_sure;                             // candidates to take: the k largest of them
    const bool fail = k < 0 || k > C || cand_n > SEL_CAP;
    if (fail && blockIdx.x == 0 && tid == 0) { ws[2] = 1; if (a.fail_word) *a.fail_word = 1; }
    // exact k-th largest candidate key: 4-pass radix select over the LDS copy (k == 0: nothing is taken)
    unsigned prefix = 0u; int rem = k;
    if (k > 0 && !fail) {
        for (int pass = 3; pass >= 0; --pass) {
            s_hist[tid] = 0;
            __syncthreads();
            const int shift = 8 * pass;
            const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int i = tid; i < C; i += 256) {
                const unsigned key = s_key[i];
                if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&s_hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            if (w == 0) {                                        // digit = the largest d with count(digit >= d) >= rem: one wave, four bins
                const int h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2], h3 = s_hist[4 * lane + 3];    // per lane
                const int mine = h0 + h1 + h2 + h3;
                int suf = mine;                                  // keys whose digit lies in this lane's bins or above
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_down(suf, off); if (lane + off < 64) suf += v; }
                const unsigned long long bal = __ballot(suf >= rem);
                const int top = bal ? 63 - __clzll(bal) : 0;     // (rem <= C: some lane qualifies; lane 0 otherwise, like the serial scan)
                if (lane == top) {
                    int above = suf - mine, d;                   // keys whose digit lies in a higher lane's bins
                    if (above + h3 >= rem) d = 4 * lane + 3;
                    else {
                        above += h3;
                        if (above + h2 >= rem) d = 4 * lane + 2;
                        else {
                            above += h2;
                            if (above + h1 >= rem) d = 4 * lane + 1;
                            else { above += h1; d = 4 * lane; }
                        }
                    }
                    s_T = prefix | ((unsigned)d << shift); s_take = rem - above;
                }
            }
            __syncthreads();
            prefix = s_T; rem = s_take;
            __syncthreads();
        }
    }
    const bool take = k > 0 && !fail;
    const unsigned T = prefix;                                   // in the scaled space of s_key
    const unsigned Tkey = (prefix >> sh) + fr.lo;                // the same threshold as a raw key
    // selected candidates in front of this wave's range
    const int wstart = gw * fr.wchunk, wend = min(fr.M, wstart + fr.wchunk);
    int cb = 0;
    if (take) for (int i = lane; i < C; i += 64) cb += (s_key[i] >= T && s_idx[i] < wstart) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cb += __shfl_xor(cb, off);
    int rank = before + cb;
    const int cap = fr.n;
    for (int i0 = wstart; i0 < wend; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < wend;
        const unsigned key = live ? nl_select_key(fr.seed, (unsigned)i) : 0u;
        const bool sel = live && (key > fr.hi || (take && key >= Tkey && key >= fr.lo));
        const unsigned long long bal = __ballot(sel);
        if (live && fr.mask) fr.mask[i] = sel ? 1 : 0;
        if (sel) {
            const int o = rank + __popcll(bal & ((1ull << lane) - 1ull));
            if (o < cap) {
                const size_t oo = (size_t)(fr.out_off + o);
                sel_emit(fr.d, fr.p, fr.c, (size_t)i, fr.frame, a.out_d, a.out_p, a.out_c, a.out_frame, oo);
            }
        }
        rank += __popcll(bal);
    }
}

extern "C" {

int nl_unit_dirs(int M, const float* points, float* out_rays_d, float* out_rays_norm, void* stream)
{
    if (M < 0 || (M > 0 && (!points || !out_rays_d))) return NL_ERR_INVALID_ARG;
    if (M == 0) return NL_OK;
    const int blocks = nl_div_up(M, 256) < 2048 ? nl_div_up(M, 256) : 2048;
    hipLaunchKernelGGL(k_unit_dirs, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, points, out_rays_d, out_rays_norm);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_select_rays(int M, int n_select, unsigned seed, const float* rays_d, const float* points, const float* cos_in, int frame,
                   float* out_rays_d, float* out_points, float* out_cos, int* out_frame_id, unsigned char* mask_out,
                   int* workspace, void* stream)
{
    if (M <= 0 || n_select <= 0 || !points || !cos_in || !out_rays_d || !out_points || !out_cos || !workspace)
        return NL_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    int* state = workspace;                 // 8 ints
    int* hist = workspace + 8;              // 256
    int* flags = workspace + 264;           // M
    int* rank = flags + M;                  // M
    int* scan_ws = rank + M;                // ceil(M / 1024) + 1
    const int blocks = nl_div_up(M, 256) < 1024 ? nl_div_up(M, 256) : 1024;
    hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(256), 0, st, state, hist, n_select);
    if (n_select < M) {
        for (int pass = 3; pass >= 0; --pass) {
            hipLaunchKernelGGL(k_sel_hist, dim3(blocks), dim3(256), 0, st, M, seed, pass, state, hist);
            hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(256), 0, st, pass, state, hist);
        }
    }
    hipLaunchKernelGGL(k_sel_flags, dim3(blocks), dim3(256), 0, st, M, seed, n_select, state, flags, mask_out);
    NL_LAUNCH_CHECK();
    const int rc = nl_exclusive_scan_i32(flags, rank, M, 1, state + 2, scan_ws, stream);
    if (rc != NL_OK) return rc;
    hipLaunchKernelGGL(k_sel_gather, dim3(blocks), dim3(256), 0, st, M, flags, rank, rays_d, points, cos_in, frame,
                       n_select < M ? n_select : M, out_rays_d, out_points, out_cos, out_frame_id);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* All frames of a call in two launches (k_sel_window_a / _b above).  Returns NL_ERR_CAPACITY when a frame's shape is outside the
 * window method's range (n close to M, or a candidate window above SEL_CAP keys): use nl_select_rays per frame then.
 * workspace: NL_SELECT_BATCH_WS_INTS(F) ints, ZERO-FILLED once by the caller at allocation; parity alternates 0 / 1 between calls
 * (the candidate counter of one call is cleared by the next one's second pass).  ws[f][2] != 0 afterwards (and *fail_word = 1, if
 * given) = the window was missed (never observed; ~1e-50) and the selection of that call is incomplete. */
int nl_select_rays_batch_ex(int F, const int* M, const int* n_select, const unsigned* seed, const float* const* rays_d,
                            const float* const* points, const float* const* cos_in, unsigned char* const* mask_out, const int* out_off,
                            const int* frame_ids, float* out_rays_d, float* out_points, float* out_cos, int* out_frame_id, int* workspace,
                            int parity, int* fail_word, void* stream)
{
    if (F <= 0 || F > SEL_MAX_FRAMES || !M || !n_select || !seed || !points || !cos_in || !out_off || !out_rays_d || !out_points ||
        !out_cos || !workspace || (parity != 0 && parity != 1))
        return NL_ERR_INVALID_ARG;
    SelArgs a;
    int max_blk = 1;
    for (int f = 0; f < F; ++f) {
        const int m = M[f], n = n_select[f];
        if (m <= 0 || n <= 0 || !points[f] || !cos_in[f]) return NL_ERR_INVALID_ARG;
        if (n >= m) return NL_ERR_CAPACITY;
        const double p = (double)n / (double)m, sigma = sqrt((double)n * (1.0 - p)), W = 16.0 * sigma + 64.0;
        if (2.0 * W + 64.0 > (double)SEL_CAP) return NL_ERR_CAPACITY;
        const double T0 = 4294967296.0 * (1.0 - p), delta = W / (double)m * 4294967296.0;
        const double lo = T0 - delta, hi = T0 + delta;
        if (lo < 1.0 || hi > 4294967294.0) return NL_ERR_CAPACITY;
        SelFrame& fr = a.f[f];
        fr.d = rays_d ? rays_d[f] : nullptr; fr.p = points[f]; fr.c = cos_in[f]; fr.mask = mask_out ? mask_out[f] : nullptr;
        fr.M = m; fr.n = n; fr.out_off = out_off[f]; fr.frame = frame_ids ? frame_ids[f] : f; fr.seed = seed[f]; fr.lo = (unsigned)lo; fr.hi = (unsigned)hi;
        int nblk = nl_div_up(m, 1024); if (nblk > SEL_MAXB) nblk = SEL_MAXB;
        fr.nblk = nblk;
        fr.wchunk = nl_div_up(nl_div_up(m, 4 * nblk), 64) * 64;
        if (nblk > max_blk) max_blk = nblk;
    }
    a.ws = workspace; a.out_d = out_rays_d; a.out_p = out_points; a.out_c = out_cos; a.out_frame = out_frame_id; a.parity = parity; a.fail_word = fail_word;
    hipLaunchKernelGGL(k_sel_window_a, dim3(max_blk, F), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_sel_window_b, dim3(max_blk, F), dim3(256), 0, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_select_rays_batch(int F, const int* M, const int* n_select, const unsigned* seed, const float* const* rays_d,
                         const float* const* points, const float* const* cos_in, unsigned char* const* mask_out, const int* out_off,
                         float* out_rays_d, float* out_points, float* out_cos, int* out_frame_id, int* workspace, int parity,
                         int* fail_word, void* stream)
{
    return nl_select_rays_batch_ex(F, M, n_select, seed, rays_d, points, cos_in, mask_out, out_off, nullptr, out_rays_d, out_points, out_cos,
                                   out_frame_id, workspace, parity, fail_word, stream);
}

}  // extern "C"
