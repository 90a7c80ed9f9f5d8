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

__global__ void k_sel_gather(int M, const int* __restrict__ flags, const int* __restrict__ rank, const float* __restrict__ rays_d,
                             const float* __restrict__ points, const float* __restrict__ cos_in, int frame, int cap,
                             float* __restrict__ out_d, float* __restrict__ out_p, float* __restrict__ out_cos, int* __restrict__ out_frame)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        if (!flags[i]) continue;
        const int o = rank[i];
        if (o >= cap) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) { out_d[3 * (size_t)o + c] = rays_d[3 * (size_t)i + c]; out_p[3 * (size_t)o + c] = points[3 * (size_t)i + c]; }
        out_cos[o] = cos_in[i];
        if (out_frame) out_frame[o] = frame;
    }
}

extern "C" {

int nl_select_rays(int M, int n_select, unsigned seed, const float* rays_d, const float* points, const float* cos_in, int frame,
                   float* out_rays_d, float* out_points, float* out_cos, int* out_frame_id, unsigned char* mask_out,
                   int* workspace, void* stream)
{
    if (M <= 0 || n_select <= 0 || !rays_d || !points || !cos_in || !out_rays_d || !out_points || !out_cos || !workspace)
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

}  // extern "C"
