// nl_field.hip -- sparse-voxel embedding field: trilinear gather (forward) and scatter-add +
// position gradient (backward), plus the SE3 pose-gradient reduction.  HBM/L2-bound.
//
// Reference behaviour: src/variations/render_helpers.py:39-93 (get_features / get_embeddings /
// trilinear_interp), autograd for the backward.  The reference chains four
// F.embedding gathers, one of them through an 8 GB host table (render_helpers.py:85-89); here the
// node->row composition is precomputed once per map update into vertex_rows[n,8] and a sample needs
// one 32-byte index read + eight 32-byte bf16 rows.
//
// Two lanes per sample, 8 channels (one 16-byte bf16 load per corner) each.
#include "nl_common.h"

#define NL_FIELD_THREADS 256
#define NL_MAX_FRAMES 32

struct FieldArgs {
    const NlLossScalars* ls;        // ->P valid samples (device)
    const int* s_vox; const float* s_depth; const int* s_ray;
    const float* rays_d_world;      // [N,3]
    const float* rays_d_sensor;     // [N,3]   (backward: dR = sum depth * dx (x) d_sensor)
    const int* frame_id;            // [N] or null
    const float* poses;             // [F,12]
    const float* centres;           // [n,3]
    const int* vertex_rows;         // [n,8]
    const uint16_t* emb;            // [E,16] bf16
    float voxel_size;
    float* X;                       // [P,16] out (forward)
    const float* dX;                // [P,16] in (backward)
    float* g_emb;                   // [E,16] fp32 accumulators (backward, atomics)
    float* g_pose;                  // [F,12] (dt[3], dR[9]) fp32 accumulators (backward)
    int n_frames;
    int want_emb_grad;
    int want_pose_grad;
};

struct SampleGeom { float p[3]; float depth; int ray; int vox; };

__device__ __forceinline__ SampleGeom sample_geom(const FieldArgs& a, int s)
{
    SampleGeom g;
    g.vox = a.s_vox[s]; g.ray = a.s_ray[s]; g.depth = a.s_depth[s];
    const float* T = a.poses + 12 * (a.frame_id ? a.frame_id[g.ray] : 0) + 9;
    float x[3], c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x[i] = T[i] + a.rays_d_world[3 * g.ray + i] * g.depth;       // ray(): o + d * depth
        c[i] = a.centres[3 * g.vox + i];
    }
    nl_trilinear_p(x, c, a.voxel_size, g.p);
    return g;
}

__device__ __forceinline__ void load_rows(const FieldArgs& a, int vox, int rows[8])
{
    const int4 r0 = *reinterpret_cast<const int4*>(a.vertex_rows + 8 * (size_t)vox);
    const int4 r1 = *reinterpret_cast<const int4*>(a.vertex_rows + 8 * (size_t)vox + 4);
    rows[0] = r0.x; rows[1] = r0.y; rows[2] = r0.z; rows[3] = r0.w; rows[4] = r1.x; rows[5] = r1.y; rows[6] = r1.z; rows[7] = r1.w;
}

__device__ __forceinline__ void load_emb8(const uint16_t* emb, int row, int half, float e[8])
{
    const uint4 v = *reinterpret_cast<const uint4*>(emb + (size_t)row * NL_C + 8 * half);
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[2 * i] = nl_bf16_to_f32((uint16_t)(u[i] & 0xFFFFu)); e[2 * i + 1] = nl_bf16_to_f32((uint16_t)(u[i] >> 16)); }
}

__global__ __launch_bounds__(NL_FIELD_THREADS) void k_gather_trilinear(FieldArgs a)
{
    const int P = a.ls->P;
    const int half = threadIdx.x & 1;
    for (int s = (blockIdx.x * NL_FIELD_THREADS + threadIdx.x) >> 1; s < P; s += (gridDim.x * NL_FIELD_THREADS) >> 1) {
        const SampleGeom g = sample_geom(a, s);
        float w[8]; nl_trilinear_w(g.p, w);
        int rows[8]; load_rows(a, g.vox, rows);
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float e[8]; load_emb8(a.emb, rows[k], half, e);
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = f[c] + w[k] * e[c];
        }
        float4* o = reinterpret_cast<float4*>(a.X + (size_t)s * NL_C + 8 * half);
        o[0] = make_float4(f[0], f[1], f[2], f[3]); o[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
}

// get_features on explicit points (mesh-time get_scores / eval_points, render_helpers.py:96-188): xyz[P,3] world
// positions with their voxel ids -> X[P,16].  Same arithmetic as k_gather_trilinear, no ray bookkeeping.
__global__ __launch_bounds__(NL_FIELD_THREADS) void k_gather_points(int P, const float* __restrict__ xyz, const int* __restrict__ vox,
                                                                    const float* __restrict__ centres, const int* __restrict__ vertex_rows,
                                                                    const uint16_t* __restrict__ emb, float voxel_size, float* __restrict__ X)
{
    const int half = threadIdx.x & 1;
    for (int s = (blockIdx.x * NL_FIELD_THREADS + threadIdx.x) >> 1; s < P; s += (gridDim.x * NL_FIELD_THREADS) >> 1) {
        const int v = vox[s];
        float x[3], c[3], p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { x[i] = xyz[3 * (size_t)s + i]; c[i] = centres[3 * (size_t)v + i]; }
        nl_trilinear_p(x, c, voxel_size, p);
        float w[8]; nl_trilinear_w(p, w);
        const int4 r0 = *reinterpret_cast<const int4*>(vertex_rows + 8 * (size_t)v);
        const int4 r1 = *reinterpret_cast<const int4*>(vertex_rows + 8 * (size_t)v + 4);
        const int rows[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float e[8]; load_emb8(emb, rows[k], half, e);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) f[ch] = f[ch] + w[k] * e[ch];
        }
        float4* o = reinterpret_cast<float4*>(X + (size_t)s * NL_C + 8 * half);
        o[0] = make_float4(f[0], f[1], f[2], f[3]); o[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
}

// backward: dE[row_k] += bf16(w_k * dX)   (fp32 accumulation of bf16-rounded contributions, then one
// bf16 rounding in the optimiser = torch's CUDA embedding_dense_backward semantics), and
// dL/dx = (1/vs) * d/dp sum_k w_k <e_k, dX>  ->  dt += dx, dR += depth * dx (x) d_sensor.
//
// The scatter is the HBM-atomic hot spot (8 rows x 16 channels per sample = 140 M fp32 atomics for one
// 64x2048 scan).  Samples are packed in ray order, so a chunk of consecutive samples touches few
// distinct vertex rows: each workgroup aggregates a 1024-sample chunk in an LDS hash table
// (open addressing on the row id, ds_add_f32 accumulation; a lane pair pre-accumulates each voxel run of its 8
// consecutive samples in registers) and flushes every touched row once with global atomics - ~40x fewer HBM atomics.  Table overflow falls back to direct global atomics.
#define TB_CHUNK 1024
#define TB_SLOTS 1024
#define TB_PROBES 16

__device__ __forceinline__ int tb_insert(int* s_key, int key)
{
    unsigned h = ((unsigned)key * 2654435761u) >> 22;          // top 10 bits -> [0, 1024)
#pragma unroll 1
    for (int i = 0; i < TB_PROBES; ++i) {
        const int prev = atomicCAS(&s_key[h], -1, key);
        if (prev == -1 || prev == key) return (int)h;
        h = (h + 1) & (TB_SLOTS - 1);
    }
    return -1;
}

__global__ __launch_bounds__(NL_FIELD_THREADS) void k_trilinear_bwd(FieldArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_val[TB_SLOTS * NL_C];
    __shared__ int s_key[TB_SLOTS];
    __shared__ float s_pose[NL_MAX_FRAMES * 12];
    for (int i = threadIdx.x; i < a.n_frames * 12; i += NL_FIELD_THREADS) s_pose[i] = 0.f;
    for (int i = threadIdx.x; i < TB_SLOTS; i += NL_FIELD_THREADS) s_key[i] = -1;
    for (int i = threadIdx.x; i < TB_SLOTS * NL_C; i += NL_FIELD_THREADS) s_val[i] = 0.f;
    __syncthreads();
    const int P = a.ls->P;
    const int half = threadIdx.x & 1;
    const int pair = threadIdx.x >> 1;
    const int nchunks = (P + TB_CHUNK - 1) / TB_CHUNK;
    float pa[12]; int pf = -1;                                  // running pose partials of this lane's current frame
#pragma unroll
    for (int i = 0; i < 12; ++i) pa[i] = 0.f;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int s_end = min(P, (chunk + 1) * TB_CHUNK);
        // a lane pair walks 8 CONSECUTIVE samples: consecutive samples of a ray mostly stay in one voxel, so the
        // 8 corner rows' contributions accumulate in registers and reach the LDS table once per voxel run
        int cur_vox = -1; int rows[8];
        float acc[8][8];
        auto flush_run = [&]() {
            if (cur_vox < 0 || !a.want_emb_grad) return;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int slot = tb_insert(s_key, rows[k]);
                if (slot >= 0) {
                    float* dst = s_val + slot * NL_C + 8 * half;
#pragma unroll
                    for (int c = 0; c < 8; ++c) atomicAdd(dst + c, acc[k][c]);                  // ds_add_f32
                } else {
                    float* dst = a.g_emb + (size_t)rows[k] * NL_C + 8 * half;
#pragma unroll
                    for (int c = 0; c < 8; ++c) atomicAdd(dst + c, acc[k][c]);                  // global_atomic_add_f32
                }
            }
        };
        const int s_base = chunk * TB_CHUNK + pair * (TB_CHUNK / (NL_FIELD_THREADS / 2));
#pragma unroll 1
        for (int j = 0; j < TB_CHUNK / (NL_FIELD_THREADS / 2); ++j) {
            const int s = s_base + j;
            if (s >= s_end) break;                                  // both lanes of a pair share s: shuffles stay convergent
            const SampleGeom g = sample_geom(a, s);
            float w[8]; nl_trilinear_w(g.p, w);
            if (g.vox != cur_vox) {
                flush_run();
                cur_vox = g.vox; load_rows(a, g.vox, rows);
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[k][c] = 0.f;
            }
            float d[8];
            const float4* di = reinterpret_cast<const float4*>(a.dX + (size_t)s * NL_C + 8 * half);
            const float4 d0 = di[0], d1 = di[1];
            d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
            float dot[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (a.want_emb_grad) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[k][c] += nl_round_bf16(w[k] * d[c]);
                }
                if (a.want_pose_grad) {
                    float e[8]; load_emb8(a.emb, rows[k], half, e);
                    float t = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) t += e[c] * d[c];
                    dot[k] = t;
                }
            }
            if (!a.want_pose_grad) continue;
#pragma unroll
            for (int k = 0; k < 8; ++k) dot[k] += __shfl_xor(dot[k], 1);
            if (half != 0) continue;
            float dp[3]; nl_trilinear_dp(g.p, dot, dp);
            const int f = a.frame_id ? a.frame_id[g.ray] : 0;
            if (f != pf) {
                if (pf >= 0) { for (int i = 0; i < 12; ++i) atomicAdd(s_pose + 12 * pf + i, pa[i]); }
#pragma unroll
                for (int i = 0; i < 12; ++i) pa[i] = 0.f;
                pf = f;
            }
            const float ds0 = a.rays_d_sensor[3 * g.ray], ds1 = a.rays_d_sensor[3 * g.ray + 1], ds2 = a.rays_d_sensor[3 * g.ray + 2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float dx = dp[i] / a.voxel_size;
                pa[i] += dx;
                const float t = g.depth * dx;
                pa[3 + 3 * i] += t * ds0; pa[4 + 3 * i] += t * ds1; pa[5 + 3 * i] += t * ds2;
            }
        }
        flush_run();
        if (a.want_emb_grad) {
            __syncthreads();
            // flush: 16 lanes per slot (one channel each), touched rows only; slot is reset for the next chunk
            for (int base = 0; base < TB_SLOTS; base += NL_FIELD_THREADS / NL_C) {
                const int slot = base + (threadIdx.x >> 4), c = threadIdx.x & 15;
                const int key = s_key[slot];
                if (key >= 0) {
                    const float v = s_val[slot * NL_C + c];
                    if (v != 0.f) atomicAdd(a.g_emb + (size_t)key * NL_C + c, v);
                    s_val[slot * NL_C + c] = 0.f;
                    if (c == 0) s_key[slot] = -1;
                }
            }
            __syncthreads();
        }
    }
    if (a.want_pose_grad) {
        if (pf >= 0) { for (int i = 0; i < 12; ++i) atomicAdd(s_pose + 12 * pf + i, pa[i]); }
        __syncthreads();
        for (int i = threadIdx.x; i < a.n_frames * 12; i += NL_FIELD_THREADS)
            if (s_pose[i] != 0.f) atomicAdd(a.g_pose + i, s_pose[i]);
    }
}

// scatter packed per-sample values into the reference's padded [R,S] layout (masked_scatter_ones,
// render_helpers.py:30-36,301) for API parity / tests: out[rank(ray)][slot] = v
__global__ void k_unpack_samples(const NlLossScalars* ls, const int* s_ray, const int* samp_off, const int* hit_rank,
                                 const float* sdf, const float* depth, int S_stride, float* out_sdf, float* out_z, unsigned char* out_valid)
{
    const int P = ls->P;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < P; s += gridDim.x * blockDim.x) {
        const int ray = s_ray[s];
        const int slot = s - samp_off[ray];
        const size_t o = (size_t)hit_rank[ray] * S_stride + slot;
        if (slot < S_stride) { out_sdf[o] = sdf[s]; out_z[o] = depth[s]; out_valid[o] = 1; }
    }
}

extern "C" {

static int fill_args(FieldArgs& a, const void* ls, const int* s_vox, const float* s_depth, const int* s_ray,
                     const float* rays_d_world, const float* rays_d_sensor, const int* frame_id, const float* poses, int n_frames,
                     const float* centres, const int* vertex_rows, const void* emb, float voxel_size)
{
    if (!ls || !s_vox || !s_depth || !s_ray || !rays_d_world || !poses || !centres || !vertex_rows || !emb) return NL_ERR_INVALID_ARG;
    if (n_frames <= 0 || n_frames > NL_MAX_FRAMES) return NL_ERR_INVALID_ARG;
    a.ls = (const NlLossScalars*)ls; a.s_vox = s_vox; a.s_depth = s_depth; a.s_ray = s_ray; a.rays_d_world = rays_d_world;
    a.rays_d_sensor = rays_d_sensor; a.frame_id = frame_id; a.poses = poses; a.centres = centres; a.vertex_rows = vertex_rows;
    a.emb = (const uint16_t*)emb; a.voxel_size = voxel_size; a.n_frames = n_frames;
    a.X = nullptr; a.dX = nullptr; a.g_emb = nullptr; a.g_pose = nullptr; a.want_emb_grad = 0; a.want_pose_grad = 0;
    return NL_OK;
}

int nl_gather_trilinear(const void* loss_scalars, const int* s_vox, const float* s_depth, const int* s_ray,
                        const float* rays_d_world, const int* frame_id, const float* poses, int n_frames,
                        const float* centres, const int* vertex_rows, const void* emb, float voxel_size,
                        float* X, int nblocks, void* stream)
{
    FieldArgs a;
    int rc = fill_args(a, loss_scalars, s_vox, s_depth, s_ray, rays_d_world, nullptr, frame_id, poses, n_frames, centres, vertex_rows, emb, voxel_size);
    if (rc != NL_OK || !X || nblocks <= 0) return NL_ERR_INVALID_ARG;
    a.X = X;
    hipLaunchKernelGGL(k_gather_trilinear, dim3(nblocks), dim3(NL_FIELD_THREADS), 0, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_gather_points(int P, const float* xyz, const int* vox, const float* centres, const int* vertex_rows, const void* emb,
                     float voxel_size, float* X, void* stream)
{
    if (P < 0 || !xyz || !vox || !centres || !vertex_rows || !emb || !X) return NL_ERR_INVALID_ARG;
    if (P == 0) return NL_OK;
    const int nb = nl_div_up((long long)P * 2, NL_FIELD_THREADS);
    hipLaunchKernelGGL(k_gather_points, dim3(nb < 4096 ? nb : 4096), dim3(NL_FIELD_THREADS), 0, (hipStream_t)stream, P, xyz, vox, centres,
                       vertex_rows, (const uint16_t*)emb, voxel_size, X);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_trilinear_bwd(const void* loss_scalars, const int* s_vox, const float* s_depth, const int* s_ray,
                     const float* rays_d_world, const float* rays_d_sensor, const int* frame_id, const float* poses, int n_frames,
                     const float* centres, const int* vertex_rows, const void* emb, float voxel_size,
                     const float* dX, float* g_emb, float* g_pose, int nblocks, void* stream)
{
    FieldArgs a;
    int rc = fill_args(a, loss_scalars, s_vox, s_depth, s_ray, rays_d_world, rays_d_sensor, frame_id, poses, n_frames, centres, vertex_rows, emb, voxel_size);
    if (rc != NL_OK || !dX || nblocks <= 0 || (g_pose && !rays_d_sensor)) return NL_ERR_INVALID_ARG;
    a.dX = dX; a.g_emb = g_emb; a.g_pose = g_pose; a.want_emb_grad = g_emb != nullptr; a.want_pose_grad = g_pose != nullptr;
    hipLaunchKernelGGL(k_trilinear_bwd, dim3(nblocks), dim3(NL_FIELD_THREADS), 0, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_unpack_samples(const void* loss_scalars, const int* s_ray, const int* samp_off, const int* hit_rank,
                      const float* sdf, const float* depth, int S_stride, float* out_sdf, float* out_z, unsigned char* out_valid,
                      void* stream)
{
    if (!loss_scalars || !s_ray || !samp_off || !hit_rank || !sdf || !depth || !out_sdf || !out_z || !out_valid || S_stride <= 0)
        return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_unpack_samples, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const NlLossScalars*)loss_scalars, s_ray, samp_off,
                       hit_rank, sdf, depth, S_stride, out_sdf, out_z, out_valid);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

}  // extern "C"
