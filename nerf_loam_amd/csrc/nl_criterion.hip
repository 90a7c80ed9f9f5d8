// nl_criterion.hip -- Criterion.forward / backward on caller tensors.
//
// Reference behaviour: /root/reference/src/criterion.py:16-115 (l2, no eikonal term): free-space and truncated-surface losses over the
// padded [R, S] sample block of render_rays, weighted by the data-dependent factors of :84-88, mean over ALL R * S elements.
// Inside the optimisation loops this library never materialises that block - the masks, the weights and dL/dsdf come out of the
// sampler and the decoder kernel (nl_geometry.hip, nl_decoder.hip phase D).  This file serves the other use of the class: a caller
// holding its own `outputs` dict (sdf, z_vals, ray_mask, valid_mask) and wanting the reference's loss and its gradient with
// respect to sdf.  Two launches forward (sums, scalars), one backward; fp64 sums, everything else fp32 as in the reference.
#include "nl_common.h"

#define NL_CRIT_THREADS 256

struct CritArgs {
    int R, S;
    const float* sdf; const float* z_vals; const unsigned char* valid;
    const float* points; const float* cos; const int* ray_idx;
    float tau, max_depth;
};

struct CritElem { float f, m, v, r_fs, r_sdf; };

__device__ __forceinline__ CritElem crit_elem(const CritArgs& a, long long e)
{
    const int r = (int)(e / a.S);
    const int ray = a.ray_idx ? a.ray_idx[r] : r;
    const float c = a.cos[ray];
    const float px = a.points[3 * (size_t)ray], py = a.points[3 * (size_t)ray + 1], pz = a.points[3 * (size_t)ray + 2];
    const float d = sqrtf(px * px + py * py + pz * pz) * c;            // criterion.py:33-35
    const float z = a.z_vals[e] * c;                                   // :36
    bool front, sdfm;
    nl_loss_masks(z, d, a.tau, a.max_depth, &front, &sdfm);            // :66-82
    CritElem o;
    o.f = front ? 1.0f : 0.0f; o.m = sdfm ? 1.0f : 0.0f; o.v = a.valid[e] ? 1.0f : 0.0f;
    const float s = a.sdf[e];
    o.r_fs = s * o.f * o.v - o.f;                                      // :96-97
    o.r_sdf = (z + s * a.tau) * o.m * o.v - d * o.m;                   // :98-99
    return o;
}

// ws: int[2] (front count, sdf-mask count) + double[2] (sums of squared residuals), cleared by the caller
__global__ __launch_bounds__(NL_CRIT_THREADS) void k_criterion_sums(CritArgs a, int* icnt, double* dsum)
{
    __shared__ int s_i[2];
    __shared__ double s_d[2];
    if (threadIdx.x < 2) { s_i[threadIdx.x] = 0; s_d[threadIdx.x] = 0.0; }
    __syncthreads();
    const long long n = (long long)a.R * a.S;
    int nf = 0, nm = 0;
    double q1 = 0.0, q2 = 0.0;
    for (long long e = (long long)blockIdx.x * NL_CRIT_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * NL_CRIT_THREADS) {
        const CritElem o = crit_elem(a, e);
        nf += o.f != 0.0f; nm += o.m != 0.0f;
        q1 += (double)(o.r_fs * o.r_fs); q2 += (double)(o.r_sdf * o.r_sdf);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nf += __shfl_xor(nf, off); nm += __shfl_xor(nm, off);
        q1 += __shfl_xor(q1, off); q2 += __shfl_xor(q2, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_i[0], nf); atomicAdd(&s_i[1], nm); atomicAdd(&s_d[0], q1); atomicAdd(&s_d[1], q2); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_i[0]) atomicAdd(&icnt[0], s_i[0]);
        if (s_i[1]) atomicAdd(&icnt[1], s_i[1]);
        if (s_d[0] != 0.0) atomicAdd(&dsum[0], s_d[0]);
        if (s_d[1] != 0.0) atomicAdd(&dsum[1], s_d[1]);
    }
}

// out[8]: loss, fs_loss, sdf_loss, w_fs, w_sdf, 2 / (R S), n_fs, n_sdf
__global__ void k_criterion_scalars(const int* icnt, const double* dsum, long long n, float fs_weight, float sdf_weight, float* out)
{
    const float n_fs = (float)icnt[0], n_sdf = (float)icnt[1];
    const float n_tot = n_sdf + n_fs;
    const float w_fs = 1.0f - n_fs / n_tot, w_sdf = 1.0f - n_sdf / n_tot;       // :84-88 (0 / 0 = NaN as in the reference)
    const float fs_loss = (float)(dsum[0] / (double)n) * w_fs;
    const float sdf_loss = (float)(dsum[1] / (double)n) * w_sdf;
    out[0] = fs_weight * fs_loss + sdf_weight * sdf_loss;                       // :46-47
    out[1] = fs_loss; out[2] = sdf_loss; out[3] = w_fs; out[4] = w_sdf; out[5] = 2.0f / (float)n; out[6] = n_fs; out[7] = n_sdf;
}

__global__ __launch_bounds__(NL_CRIT_THREADS) void k_criterion_backward(CritArgs a, const float* __restrict__ out, const float* __restrict__ grad_loss,
                                                                        float fs_weight, float sdf_weight, float* __restrict__ dsdf)
{
    const long long n = (long long)a.R * a.S;
    const float g = grad_loss ? *grad_loss : 1.0f;
    const float k_fs = fs_weight * out[3] * out[5], k_sdf = sdf_weight * out[4] * out[5] * a.tau;
    for (long long e = (long long)blockIdx.x * NL_CRIT_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * NL_CRIT_THREADS) {
        const CritElem o = crit_elem(a, e);
        dsdf[e] = g * (k_fs * o.r_fs * o.f * o.v + k_sdf * o.r_sdf * o.m * o.v);
    }
}

static int crit_args(CritArgs& a, int R, int S, const float* sdf, const float* z_vals, const unsigned char* valid, const float* points,
                     const float* cos, const int* ray_idx, float truncation, float max_depth)
{
    if (R <= 0 || S <= 0 || !sdf || !z_vals || !valid || !points || !cos) return NL_ERR_INVALID_ARG;
    a.R = R; a.S = S; a.sdf = sdf; a.z_vals = z_vals; a.valid = valid; a.points = points; a.cos = cos; a.ray_idx = ray_idx;
    a.tau = truncation; a.max_depth = max_depth;
    return NL_OK;
}

static int crit_blocks(long long n)
{
    const long long nb = (n + NL_CRIT_THREADS - 1) / NL_CRIT_THREADS;
    return (int)(nb < 2048 ? nb : 2048);
}

extern "C" {

int nl_criterion_forward(int R, int S, const float* sdf, const float* z_vals, const unsigned char* valid_mask, const float* points,
                         const float* cos, const int* ray_idx, float truncation, float max_depth, float fs_weight, float sdf_weight,
                         void* workspace, float* out, void* stream)
{
    CritArgs a;
    if (crit_args(a, R, S, sdf, z_vals, valid_mask, points, cos, ray_idx, truncation, max_depth) != NL_OK || !workspace || !out)
        return NL_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, 32, st) != hipSuccess) return NL_ERR_LAUNCH;
    int* icnt = (int*)workspace; double* dsum = (double*)((char*)workspace + 16);
    const long long n = (long long)R * S;
    hipLaunchKernelGGL(k_criterion_sums, dim3(crit_blocks(n)), dim3(NL_CRIT_THREADS), 0, st, a, icnt, dsum);
    hipLaunchKernelGGL(k_criterion_scalars, dim3(1), dim3(1), 0, st, icnt, dsum, n, fs_weight, sdf_weight, out);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_criterion_backward(int R, int S, const float* sdf, const float* z_vals, const unsigned char* valid_mask, const float* points,
                          const float* cos, const int* ray_idx, float truncation, float max_depth, float fs_weight, float sdf_weight,
                          const float* out, const float* grad_loss, float* dsdf, void* stream)
{
    CritArgs a;
    if (crit_args(a, R, S, sdf, z_vals, valid_mask, points, cos, ray_idx, truncation, max_depth) != NL_OK || !out || !dsdf)
        return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_criterion_backward, dim3(crit_blocks((long long)R * S)), dim3(NL_CRIT_THREADS), 0, (hipStream_t)stream, a, out, grad_loss,
                       fs_weight, sdf_weight, dsdf);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

}  // extern "C"
