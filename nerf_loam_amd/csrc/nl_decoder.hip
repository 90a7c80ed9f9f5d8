// nl_decoder.hip -- the SDF decoder (16 -> 256 -> 256 -> 1 ReLU MLP) forward + loss gradient +
// backward as ONE persistent kernel on the fp32 matrix cores of gfx950.
//
// Reference behaviour: src/variations/lidar.py:109-131 (forward), src/criterion.py:59-100 (loss),
// torch autograd for the backward.  fp32 in / fp32 accumulate MFMA
// (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32) is bit-for-bit an fmaf chain, so the 1e-4 SDF
// parity bar holds without any reduced-precision path.
//
// Structure (MI355X-first, not how the reference does it - it runs 3 GEMMs + autograd):
//   * one 512-thread workgroup (8 waves, 2 per SIMD) per CU, persistent over 64-sample tiles;
//   * per tile everything stays on chip: X tile, H1 = relu(X W1^T + b1) and dH2/dH1 live in LDS
//     (row stride 257 floats: conflict-free both as MFMA A operand [row-per-lane] and as
//     B operand / epilogue target [column-per-lane]); H2 never leaves registers;
//   * the loss gradient dL/dsdf needs only per-sample geometry + iteration-global scalars that are
//     known before the decoder runs (nl_geometry.hip), so forward, loss and backward fuse;
//   * W2 (256 KB fp32 > LDS) is streamed from L2 straight into MFMA B operands: wave w owns output
//     columns [32w, 32w+32), reads W2T rows (forward) / W2 rows (dgrad) coalesced, each operand
//     register feeds both 32-row sub-tiles;
//   * weight gradients accumulate in registers across ALL tiles of the workgroup and are flushed
//     once as a per-workgroup partial slab - no atomics; nl_reduce_partials sums the slabs.  The
//     big one, dW2 = dH2^T H1 (8 tiles of 32x32 per wave = 128 accumulator registers), runs in its
//     own persistent kernel (k_decoder_wgrad2) so that neither kernel spills: it rebuilds H1 from
//     X (K = 16, 2 % extra MFMA work) and dH2 from dsdf + the 256-bit ReLU mask the first kernel
//     saves per sample (32 B instead of a 1 KB activation row).
// FLOPs per sample: 3 * 2 * (16*256 + 256*256 + 256) = 419,328 (279,552 with a frozen decoder).
#include "nl_common.h"

#define DEC_M 64
#define DEC_THREADS 512
#define LDH 257
#define LDX 17

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// LDS carve (floats)
#define S_H1 0
#define S_D (S_H1 + DEC_M * LDH)
#define S_X (S_D + DEC_M * LDH)
#define S_W1 (S_X + DEC_M * LDX)
#define S_S (S_W1 + NL_W * NL_C)
#define S_DS (S_S + DEC_M)
#define S_TOTAL (S_DS + DEC_M)

struct DecArgs {
    const NlLossScalars* ls;
    const float* X;             // [P,16] interpolated embeddings
    const float* params;        // decoder parameter block (NL_OFF_*)
    const float* W2T;           // [256][256], W2T[k][j] = W2[j][k]
    const int* s_ray;           // [P]
    const float* s_depth;       // [P]
    const float* cos_gt;        // [N]
    const float* gt_dist;       // [N]
    float* sdf;                 // [P]
    float* dsdf;                // [P]
    float* dX;                  // [P,16]
    float* partials;            // [gridDim.x][NL_DEC_PARAMS] (train only)
    unsigned* relu2_mask;       // [tiles][512] one word per (tile, thread): that lane's 32 ReLU bits of H2 (train only)
    double* dcounters;          // loss sums
    long long* dbg;             // optional [16 tiles][16] s_memtime stamps of workgroup 0 / thread 0 (profiling aid)
    int stagger;                // k_decoder32: start delay (cycles) of the second workgroup on each CU
};

#define DBG_STAMP(slot)                                                                          \
    do { if (a.dbg && blockIdx.x == 0 && tid == 0 && tile_no < 16) a.dbg[tile_no * 16 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

// row of accumulator register r in a 32x32 MFMA result for this lane
__device__ __forceinline__ int d32_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// 256-deep GEMM main loop shared by forward (B = W2T) and dgrad (B = W2): per k-pair one coalesced
// B-operand load straight from L2 feeds both 32-row sub-tiles; B operands are register
// double-buffered one group of 8 ahead so the L2 latency hides under 16 MFMAs of the previous group.
typedef __amdgpu_buffer_rsrc_t i32x4;   // 128-bit buffer resource descriptor

// 128-bit buffer resource over a [256][256] fp32 matrix (wave-uniform): loads then need ONE per-lane
// 32-bit byte offset + a scalar offset, instead of a 64-bit address pair per unrolled load.
__device__ __forceinline__ i32x4 make_w_rsrc(const float* base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, NL_W * NL_W * 4, 0x00020000);
}
__device__ __forceinline__ float bload(i32x4 rsrc, int voff_bytes, int soff_bytes)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes, soff_bytes, 0));
}

__device__ __forceinline__ void gemm256(i32x4 rsrc, int voff_bytes, const float* ap0, const float* ap1, f32x16& c0, f32x16& c1)
{
    constexpr int RB = 2 * NL_W * 4;                   // bytes between consecutive k-pairs of the B matrix
    float bA[8], bB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bA[i] = bload(rsrc, voff_bytes, i * RB);
#pragma unroll 1
    for (int g = 0; g < NL_W / 2; g += 16) {
        const int so = g * RB;
#pragma unroll
        for (int i = 0; i < 8; ++i) bB[i] = bload(rsrc, voff_bytes, so + (8 + i) * RB);
#pragma unroll
        for (int i = 0; i < 8; ++i) { c0 = MFMA32(ap0[2 * (g + i)], bA[i], c0); c1 = MFMA32(ap1[2 * (g + i)], bA[i], c1); }
        if (g + 16 < NL_W / 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bA[i] = bload(rsrc, voff_bytes, so + (16 + i) * RB);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { c0 = MFMA32(ap0[2 * (g + 8 + i)], bB[i], c0); c1 = MFMA32(ap1[2 * (g + 8 + i)], bB[i], c1); }
    }
}

// For each of the 32 rows a half-wave holds (16 of h0 ++ 16 of h1), the sum over its 32 lanes of
// h[row] * w3c, by recursive halving: 31 shuffles instead of 160; lane l31 ends up with the total of
// list entry e = l31.  Register-lean: level 1 consumes h0/h1 directly (16 live values), then 8, 4, 2, 1.
__device__ __forceinline__ float halfwave_rowsum(const f32x16& h0, const f32x16& h1, float w3c, int l31)
{
    float v16[16], v8[8], v4[4], v2[2];
    {
        const bool up = (l31 & 16) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a0 = h0[i] * w3c, a1 = h1[i] * w3c;
            v16[i] = (up ? a1 : a0) + __shfl_xor(up ? a0 : a1, 16);
        }
    }
    {
        const bool up = (l31 & 8) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = (up ? v16[i + 8] : v16[i]) + __shfl_xor(up ? v16[i] : v16[i + 8], 8);
    }
    {
        const bool up = (l31 & 4) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v4[i] = (up ? v8[i + 4] : v8[i]) + __shfl_xor(up ? v8[i] : v8[i + 4], 4);
    }
    {
        const bool up = (l31 & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) v2[i] = (up ? v4[i + 2] : v4[i]) + __shfl_xor(up ? v4[i] : v4[i + 2], 2);
    }
    const bool up = (l31 & 1) != 0;
    return (up ? v2[1] : v2[0]) + __shfl_xor(up ? v2[0] : v2[1], 1);
}

template <bool TRAIN>
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decoder(DecArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[S_TOTAL];
    float* sH1 = lds + S_H1; float* sD = lds + S_D; float* sX = lds + S_X; float* sW1 = lds + S_W1;
    float* sS = lds + S_S; float* sdS = lds + S_DS;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const int col = 32 * w + l31;                 // this lane's output column in 32x32 tiles
    const NlLossScalars ls = *a.ls;
    const int P = ls.P;
    const int ntiles = (P + DEC_M - 1) / DEC_M;

    const i32x4 rsW2 = make_w_rsrc(a.params + NL_OFF_W2), rsW2T = make_w_rsrc(a.W2T);
    const float b1c = a.params[NL_OFF_B1 + col], b2c = a.params[NL_OFF_B2 + col], w3c = a.params[NL_OFF_W3 + col];
    const float b3 = a.params[NL_OFF_B3];

    // persistent weight-gradient accumulators (dW2 lives in k_decoder_wgrad2)
    f32x4 accW1[4];
    float aW3 = 0.f, aB2 = 0.f, aB1 = 0.f, aB3 = 0.f;
    double lossFs = 0.0, lossSdf = 0.0;
    if (TRAIN) {
#pragma unroll
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) accW1[t][r] = 0.f;
    }
    for (int i = tid; i < NL_W * NL_C; i += DEC_THREADS) sW1[i] = a.params[NL_OFF_W1 + i];
    if (tid < DEC_M) sS[tid] = 0.f;

    // software prefetch of the next tile's inputs (X slice; loss inputs for the 64 sample-owner threads)
    const int xe = tid * 2, xi = xe >> 4, xc = xe & 15;
    float2 xv = make_float2(0.f, 0.f);
    float pz = 0.f, pd = 0.f;
    {
        const int row0 = blockIdx.x * DEC_M;
        if (blockIdx.x < ntiles && row0 + xi < P) xv = *reinterpret_cast<const float2*>(a.X + (size_t)(row0 + xi) * NL_C + xc);
        if (tid < DEC_M && blockIdx.x < ntiles && row0 + tid < P) {
            const int ray = a.s_ray[row0 + tid];
            pz = a.s_depth[row0 + tid] * a.cos_gt[ray]; pd = a.gt_dist[ray];
        }
    }
    __syncthreads();

    int tile_no = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++tile_no) {
        const int row0 = tile * DEC_M;
        DBG_STAMP(0);
        // ---------------- A: X tile -> LDS ; issue the next tile's loads ----------------
        sX[xi * LDX + xc] = xv.x; sX[xi * LDX + xc + 1] = xv.y;
        const float cz = pz, cd = pd;
        {
            const int nrow0 = (tile + gridDim.x) * DEC_M;
            xv = make_float2(0.f, 0.f);
            if (nrow0 + xi < P) xv = *reinterpret_cast<const float2*>(a.X + (size_t)(nrow0 + xi) * NL_C + xc);
            if (tid < DEC_M && nrow0 + tid < P) {
                const int ray = a.s_ray[nrow0 + tid];
                pz = a.s_depth[nrow0 + tid] * a.cos_gt[ray]; pd = a.gt_dist[ray];
            }
        }
        __syncthreads();
        DBG_STAMP(1);
        // ---------------- B: H1 = relu(X W1^T + b1) ----------------
        {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < NL_C / 2; ++kk) {
                const float bw = sW1[col * NL_C + 2 * kk + lh];
                c0 = MFMA32(sX[l31 * LDX + 2 * kk + lh], bw, c0); c1 = MFMA32(sX[(32 + l31) * LDX + 2 * kk + lh], bw, c1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = d32_row(r, lh);
                sH1[row * LDH + col] = fmaxf(c0[r] + b1c, 0.f);
                sH1[(32 + row) * LDH + col] = fmaxf(c1[r] + b1c, 0.f);
            }
        }
        __syncthreads();
        DBG_STAMP(2);
        // ---------------- C: H2 = relu(H1 W2^T + b2), s = H2 w3 + b3 ----------------
        f32x16 h0, h1;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
            gemm256(rsW2T, (lh * NL_W + col) * 4, sH1 + l31 * LDH + lh, sH1 + (32 + l31) * LDH + lh, h0, h1);
            DBG_STAMP(3);
#pragma unroll
            for (int r = 0; r < 16; ++r) { h0[r] = fmaxf(h0[r] + b2c, 0.f); h1[r] = fmaxf(h1[r] + b2c, 0.f); }
            const float tot = halfwave_rowsum(h0, h1, w3c, l31);               // entry e = l31 of [h0 rows | h1 rows]
            atomicAdd(&sS[(l31 >> 4) * 32 + d32_row(l31 & 15, lh)], tot);
        }
        __syncthreads();
        DBG_STAMP(4);
        // ---------------- D: sdf, loss gradient (criterion.py) ----------------
        if (tid < DEC_M) {
            const int g = row0 + tid;
            float ds = 0.f;
            if (g < P) {
                const float s = sS[tid] + b3;
                bool f, m;
                nl_loss_masks(cz, cd, ls.tau, ls.max_depth, &f, &m);
                float q1, q2;
                ds = nl_loss_grad(s, cz, cd, f, m, ls, &q1, &q2);
                a.sdf[g] = s; a.dsdf[g] = ds;
                lossFs += (double)q1; lossSdf += (double)q2;
            }
            sdS[tid] = ds; sS[tid] = 0.f;
            if (TRAIN) aB3 += ds;
        }
        __syncthreads();
        DBG_STAMP(5);
        // ---------------- E: dH2 = ds * w3 * [H2 > 0] -> LDS ----------------
        {
            unsigned mw = 0u;                       // this lane's 32 ReLU bits: bit r = h0[r] > 0, bit 16+r = h1[r] > 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = d32_row(r, lh);
                const float ds0 = sdS[row], ds1 = sdS[32 + row];
                const bool on0 = h0[r] > 0.f, on1 = h1[r] > 0.f;
                const float g0 = on0 ? ds0 * w3c : 0.f, g1 = on1 ? ds1 * w3c : 0.f;
                sD[row * LDH + col] = g0; sD[(32 + row) * LDH + col] = g1;
                if (TRAIN) {
                    aW3 = fmaf(ds0, h0[r], aW3); aW3 = fmaf(ds1, h1[r], aW3); aB2 += g0; aB2 += g1;
                    mw |= (on0 ? (1u << r) : 0u) | (on1 ? (1u << (16 + r)) : 0u);
                }
            }
            // one word per thread, thread-major per tile: k_decoder_wgrad2's thread (same wave/lane) reads it back
            if (TRAIN) a.relu2_mask[(size_t)tile * DEC_THREADS + tid] = mw;
        }
        __syncthreads();
        DBG_STAMP(6);
        // ---------------- F: dH1 = (dH2 W2) * [H1 > 0]  (kept in registers) ----------------
        f32x16 g0v, g1v;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { g0v[r] = 0.f; g1v[r] = 0.f; }
            gemm256(rsW2, (lh * NL_W + col) * 4, sD + l31 * LDH + lh, sD + (32 + l31) * LDH + lh, g0v, g1v);
            DBG_STAMP(7);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = d32_row(r, lh);
                g0v[r] = sH1[row * LDH + col] > 0.f ? g0v[r] : 0.f;
                g1v[r] = sH1[(32 + row) * LDH + col] > 0.f ? g1v[r] : 0.f;
                if (TRAIN) aB1 += g0v[r] + g1v[r];
            }
        }
        __syncthreads();
        DBG_STAMP(8);
        // ---------------- H: dH1 -> LDS (over dH2) ----------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = d32_row(r, lh);
            sD[row * LDH + col] = g0v[r]; sD[(32 + row) * LDH + col] = g1v[r];
        }
        __syncthreads();
        DBG_STAMP(9);
        // ---------------- I: waves 0-3: dX[16 rows each] = dH1 W1 -> global ; waves 4-7: dW1 += dH1^T X ----------------
        if (w < 4) {
            f32x4 cxa = {0.f, 0.f, 0.f, 0.f}, cxb = {0.f, 0.f, 0.f, 0.f};
            const float* ap = sD + (16 * w + l15) * LDH + lq;
            const float* bq = sW1 + lq * NL_C + l15;
#pragma unroll 8
            for (int q = 0; q < NL_W / 4; q += 2) {
                cxa = MFMA16(ap[4 * q], bq[4 * q * NL_C], cxa);
                cxb = MFMA16(ap[4 * q + 4], bq[(4 * q + 4) * NL_C], cxb);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = row0 + 16 * w + 4 * lq + r;
                if (g < P) a.dX[(size_t)g * NL_C + l15] = cxa[r] + cxb[r];
            }
        } else if (TRAIN) {
            const int hb = 64 * (w - 4);
#pragma unroll 4
            for (int ii = 0; ii < DEC_M / 4; ++ii) {
                const float xb = sX[(4 * ii + lq) * LDX + l15];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    accW1[t] = MFMA16(sD[(4 * ii + lq) * LDH + hb + 16 * t + l15], xb, accW1[t]);
            }
        }
        __syncthreads();
        DBG_STAMP(10);
    }

    // ---------------- loss sums ----------------
    if (tid < 64) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lossFs += __shfl_xor(lossFs, off); lossSdf += __shfl_xor(lossSdf, off); }
        if (tid == 0 && (lossFs != 0.0 || lossSdf != 0.0)) {
            atomicAdd(&a.dcounters[NLD_FS_SQ], lossFs); atomicAdd(&a.dcounters[NLD_SDF_SQ], lossSdf);
        }
    }
    // ---------------- flush weight-gradient partials ----------------
    if (TRAIN) {
        float* base = a.partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
        if (w >= 4) {
            const int hb = 64 * (w - 4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    base[NL_OFF_W1 + (hb + 16 * t + 4 * lq + r) * NL_C + l15] = accW1[t][r];
        }
        aW3 += __shfl_xor(aW3, 32); aB2 += __shfl_xor(aB2, 32); aB1 += __shfl_xor(aB1, 32);
        if (lh == 0) { base[NL_OFF_W3 + col] = aW3; base[NL_OFF_B2 + col] = aB2; base[NL_OFF_B1 + col] = aB1; }
        if (tid < 64) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) aB3 += __shfl_xor(aB3, off);
            if (tid == 0) base[NL_OFF_B3] = aB3;
        }
    }
}

// =============================================================================================
// k_decoder32: the same fused forward + loss + backward on 32-SAMPLE tiles with 256-thread workgroups, TWO workgroups
// per CU (77 KB LDS each).  One 64-sample workgroup per CU leaves the matrix pipe idle during every epilogue / barrier
// phase (measured: MFMA busy 77.6 %); two independent workgroups drift apart, so one's serial phases run under the
// other's MFMA loops.  Each of the 4 waves owns 64 output columns (two 32x32 tiles on the same 32 rows: one A operand
// from LDS feeds both), W1 lives in registers (32 values per lane), W2 streams from L2 as before.
// =============================================================================================
// LICM hoists every "base + constant" LDS address out of the persistent tile loop into its own VGPR (dozens of them);
// laundering the base through an empty asm inside the loop keeps ONE base register and lets the constants fold into the
// ds_read/ds_write offset fields.
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
#define D32_RR(r) (((r) & 3) + 8 * ((r) >> 2))          // row of accumulator register r within a half-wave (+ 4*lh)

// Two workgroups that share a CU start in lock-step and - with the matrix pipe arbitrated fairly between their waves -
// STAY in lock-step: both hit their serial phases (epilogues, loss, LDS round trips) at the same time and the pipe idles.
// The second arrival on each CU therefore waits half a tile period once, at kernel start; the offset persists, and from
// then on one workgroup's serial phases run under the other's GEMM loops.
__device__ unsigned g_cu_arrivals[1024];
__device__ __forceinline__ unsigned cu_key()
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: cu_id[11:8] sh_id[12] se_id[14:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // HW_REG_XCC_ID[3:0]
    return ((xcc & 7u) << 7) | (((hw >> 13) & 3u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
}
__device__ __forceinline__ void stagger_second_workgroup(int cycles)
{
    if (threadIdx.x == 0) {
        const unsigned slot = atomicAdd(&g_cu_arrivals[cu_key()], 1u);
        if (slot & 1u) {
            const long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(16);
        }
    }
}

#define D32_M 32
#define D32_THREADS 256
#define D32_KP 8                                        // B-operand prefetch depth (k-pairs): covers an L2 hit with ONE wave per SIMD in the loop
#define T_H1 0
#define T_D (T_H1 + D32_M * LDH)
#define T_X (T_D + D32_M * LDH)
#define T_PX (T_X + D32_M * LDX)
#define T_S (T_PX + 4 * D32_M * LDX)
#define T_DS (T_S + D32_M)
#define T_TOTAL (T_DS + D32_M)

template <int KP>                                        // k-pairs per prefetch stage: B operands run 2*KP MFMAs (128*KP cycles) ahead
__device__ __forceinline__ void gemm256_2col(i32x4 rsrc, int voff_bytes, const float* ap, f32x16& c0, f32x16& c1)
{
    // Both operand streams are software-pipelined one stage ahead: B from L2 (buffer loads) AND A from LDS.  With two
    // workgroups per CU a wave is often alone on its SIMD, so an LDS read issued right before its MFMA would leave the
    // matrix pipe idle for the whole LDS latency every few instructions.
    constexpr int RB = 2 * NL_W * 4;
    float bA[2 * KP], bB[2 * KP], aA[KP], aB[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) { bA[2 * i] = bload(rsrc, voff_bytes, i * RB); bA[2 * i + 1] = bload(rsrc, voff_bytes + 128, i * RB); }
#pragma unroll
    for (int i = 0; i < KP; ++i) aA[i] = ap[2 * i];
#pragma unroll 1
    for (int g = 0; g < NL_W / 2; g += 2 * KP) {
        const int so = g * RB;
#pragma unroll
        for (int i = 0; i < KP; ++i) { bB[2 * i] = bload(rsrc, voff_bytes, so + (KP + i) * RB); bB[2 * i + 1] = bload(rsrc, voff_bytes + 128, so + (KP + i) * RB); }
#pragma unroll
        for (int i = 0; i < KP; ++i) aB[i] = ap[2 * (g + KP + i)];
#pragma unroll
        for (int i = 0; i < KP; ++i) { c0 = MFMA32(aA[i], bA[2 * i], c0); c1 = MFMA32(aA[i], bA[2 * i + 1], c1); }
        if (g + 2 * KP < NL_W / 2) {
#pragma unroll
            for (int i = 0; i < KP; ++i) { bA[2 * i] = bload(rsrc, voff_bytes, so + (2 * KP + i) * RB); bA[2 * i + 1] = bload(rsrc, voff_bytes + 128, so + (2 * KP + i) * RB); }
#pragma unroll
            for (int i = 0; i < KP; ++i) aA[i] = ap[2 * (g + 2 * KP + i)];
        }
#pragma unroll
        for (int i = 0; i < KP; ++i) { c0 = MFMA32(aB[i], bB[2 * i], c0); c1 = MFMA32(aB[i], bB[2 * i + 1], c1); }
    }
}

// sum over the 32 lanes of a half-wave of 16 per-lane row values; lane l31 ends with the total of entry (l31 >> 1) & 15
__device__ __forceinline__ float halfwave_rowsum16(const float (&v)[16], int l31)
{
    float v8[8], v4[4], v2[2];
    { const bool up = (l31 & 16) != 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) v8[i] = (up ? v[i + 8] : v[i]) + __shfl_xor(up ? v[i] : v[i + 8], 16); }
    { const bool up = (l31 & 8) != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) v4[i] = (up ? v8[i + 4] : v8[i]) + __shfl_xor(up ? v8[i] : v8[i + 4], 8); }
    { const bool up = (l31 & 4) != 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) v2[i] = (up ? v4[i + 2] : v4[i]) + __shfl_xor(up ? v4[i] : v4[i + 2], 4); }
    const bool up = (l31 & 2) != 0;
    float t = (up ? v2[1] : v2[0]) + __shfl_xor(up ? v2[0] : v2[1], 2);
    return t + __shfl_xor(t, 1);
}

#define STAMP32(slot)                                                                            \
    do { if (a.dbg && tid == 0 && tile_no < 8) a.dbg[((size_t)blockIdx.x * 8 + tile_no) * 16 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

template <bool TRAIN>
__global__ __launch_bounds__(D32_THREADS, 2) void k_decoder32(DecArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[T_TOTAL];
    float* sH1 = lds + T_H1; float* sD = lds + T_D; float* sX = lds + T_X; float* sPX = lds + T_PX;
    float* sS = lds + T_S; float* sdS = lds + T_DS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const int colA = 64 * w + l31, colB = colA + 32;
    const NlLossScalars ls = *a.ls;
    const int P = ls.P;
    const int ntiles = (P + D32_M - 1) / D32_M;
    const i32x4 rsW2 = make_w_rsrc(a.params + NL_OFF_W2), rsW2T = make_w_rsrc(a.W2T);
    const float b1A = a.params[NL_OFF_B1 + colA], b1B = a.params[NL_OFF_B1 + colB];
    const float b2A = a.params[NL_OFF_B2 + colA], b2B = a.params[NL_OFF_B2 + colB];
    const float w3A = a.params[NL_OFF_W3 + colA], w3B = a.params[NL_OFF_W3 + colB];
    const float b3 = a.params[NL_OFF_B3];
    float w1b[16];                                       // register-resident W1 (phase B operands)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        w1b[2 * kk] = a.params[NL_OFF_W1 + colA * NL_C + 2 * kk + lh];
        w1b[2 * kk + 1] = a.params[NL_OFF_W1 + colB * NL_C + 2 * kk + lh];
    }
    const float* w1i_src = a.params + NL_OFF_W1 + (64 * w + lq) * NL_C + l15;      // phase I operands: L2-resident, re-read per tile

    f32x4 accW1[4];
    float aW3A = 0.f, aW3B = 0.f, aB2A = 0.f, aB2B = 0.f, aB1A = 0.f, aB1B = 0.f, aB3 = 0.f;
    double lossFs = 0.0, lossSdf = 0.0;
    if (TRAIN) {
#pragma unroll
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) accW1[t][r] = 0.f;
    }
    if (tid < D32_M) sS[tid] = 0.f;
    if (a.stagger > 0) stagger_second_workgroup(a.stagger);
    const int xe = tid * 2, xi = xe >> 4, xc = xe & 15;
    float2 xv = make_float2(0.f, 0.f);
    float pz = 0.f, pd = 0.f;
    {
        const int row0 = blockIdx.x * D32_M;
        if (blockIdx.x < ntiles && row0 + xi < P) xv = *reinterpret_cast<const float2*>(a.X + (size_t)(row0 + xi) * NL_C + xc);
        if (tid < D32_M && blockIdx.x < ntiles && row0 + tid < P) {
            const int ray = a.s_ray[row0 + tid];
            pz = a.s_depth[row0 + tid] * a.cos_gt[ray]; pd = a.gt_dist[ray];
        }
    }
    __syncthreads();

    int tile_no = 0;
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 128 + 15] = (long long)cu_key();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++tile_no) {
        const int row0 = tile * D32_M;
        STAMP32(0);
        // ---- A
        sX[xi * LDX + xc] = xv.x; sX[xi * LDX + xc + 1] = xv.y;
        const float cz = pz, cd = pd;
        {
            const int nrow0 = (tile + gridDim.x) * D32_M;
            xv = make_float2(0.f, 0.f);
            if (nrow0 + xi < P) xv = *reinterpret_cast<const float2*>(a.X + (size_t)(nrow0 + xi) * NL_C + xc);
            if (tid < D32_M && nrow0 + tid < P) {
                const int ray = a.s_ray[nrow0 + tid];
                pz = a.s_depth[nrow0 + tid] * a.cos_gt[ray]; pd = a.gt_dist[ray];
            }
        }
        __syncthreads();
        STAMP32(1);
        // ---- B: H1 = relu(X W1^T + b1)
        {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < NL_C / 2; ++kk) {
                const float av = sX[l31 * LDX + 2 * kk + lh];
                c0 = MFMA32(av, w1b[2 * kk], c0); c1 = MFMA32(av, w1b[2 * kk + 1], c1);
            }
            float* hw = sH1 + opaque(4 * lh * LDH + colA);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hw[D32_RR(r) * LDH] = fmaxf(c0[r] + b1A, 0.f); hw[D32_RR(r) * LDH + 32] = fmaxf(c1[r] + b1B, 0.f);
            }
        }
        __syncthreads();
        STAMP32(2);
        // ---- C: H2, sdf partial sums
        f32x16 h0, h1;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
            gemm256_2col<D32_KP>(rsW2T, (lh * NL_W + colA) * 4, sH1 + l31 * LDH + lh, h0, h1);
            STAMP32(3);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h0[r] = fmaxf(h0[r] + b2A, 0.f); h1[r] = fmaxf(h1[r] + b2B, 0.f);
                v[r] = h0[r] * w3A + h1[r] * w3B;
            }
            const float tot = halfwave_rowsum16(v, l31);
            if ((l31 & 1) == 0) atomicAdd(&sS[d32_row((l31 >> 1) & 15, lh)], tot);
        }
        __syncthreads();
        STAMP32(4);
        // ---- D: sdf + loss gradient
        if (tid < D32_M) {
            const int g = row0 + tid;
            float ds = 0.f;
            if (g < P) {
                const float s = sS[tid] + b3;
                bool f, m;
                nl_loss_masks(cz, cd, ls.tau, ls.max_depth, &f, &m);
                float q1, q2;
                ds = nl_loss_grad(s, cz, cd, f, m, ls, &q1, &q2);
                a.sdf[g] = s; a.dsdf[g] = ds;
                lossFs += (double)q1; lossSdf += (double)q2;
            }
            sdS[tid] = ds; sS[tid] = 0.f;
            if (TRAIN) aB3 += ds;
        }
        __syncthreads();
        STAMP32(5);
        // ---- E: dH2 -> LDS, ReLU mask word
        {
            unsigned mw = 0u;
            float* dw = sD + opaque(4 * lh * LDH + colA);
            const float* dsr = sdS + opaque(4 * lh);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ds = dsr[D32_RR(r)];
                const bool on0 = h0[r] > 0.f, on1 = h1[r] > 0.f;
                const float g0 = on0 ? ds * w3A : 0.f, g1 = on1 ? ds * w3B : 0.f;
                dw[D32_RR(r) * LDH] = g0; dw[D32_RR(r) * LDH + 32] = g1;
                if (TRAIN) {
                    aW3A = fmaf(ds, h0[r], aW3A); aW3B = fmaf(ds, h1[r], aW3B); aB2A += g0; aB2B += g1;
                    mw |= (on0 ? (1u << r) : 0u) | (on1 ? (1u << (16 + r)) : 0u);
                }
            }
            if (TRAIN) a.relu2_mask[(size_t)tile * D32_THREADS + tid] = mw;
        }
        __syncthreads();
        STAMP32(6);
        // ---- F: dH1 = (dH2 W2) * [H1 > 0]
        f32x16 g0v, g1v;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { g0v[r] = 0.f; g1v[r] = 0.f; }
            gemm256_2col<D32_KP>(rsW2, (lh * NL_W + colA) * 4, sD + l31 * LDH + lh, g0v, g1v);
            STAMP32(7);
            const float* hr = sH1 + opaque(4 * lh * LDH + colA);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g0v[r] = hr[D32_RR(r) * LDH] > 0.f ? g0v[r] : 0.f;
                g1v[r] = hr[D32_RR(r) * LDH + 32] > 0.f ? g1v[r] : 0.f;
                if (TRAIN) { aB1A += g0v[r]; aB1B += g1v[r]; }
            }
        }
        __syncthreads();
        STAMP32(8);
        // ---- H: dH1 -> LDS
        {
            float* dw = sD + opaque(4 * lh * LDH + colA);
#pragma unroll
            for (int r = 0; r < 16; ++r) { dw[D32_RR(r) * LDH] = g0v[r]; dw[D32_RR(r) * LDH + 32] = g1v[r]; }
        }
        float w1i[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) w1i[q] = w1i_src[4 * q * NL_C];
        __syncthreads();
        STAMP32(9);
        // ---- I: dX partial over this wave's 64-deep k slab; dW1 += dH1^T X for its 64 hidden rows
        {
            f32x4 cx0 = {0.f, 0.f, 0.f, 0.f}, cx1 = {0.f, 0.f, 0.f, 0.f};
            const float* ap = sD + opaque(l15 * LDH + 64 * w + lq);
#pragma unroll
            for (int q = 0; q < 16; ++q) { cx0 = MFMA16(ap[4 * q], w1i[q], cx0); cx1 = MFMA16(ap[16 * LDH + 4 * q], w1i[q], cx1); }
            float* px = sPX + opaque(w * (D32_M * LDX) + 4 * lq * LDX + l15);
#pragma unroll
            for (int r = 0; r < 4; ++r) { px[r * LDX] = cx0[r]; px[(16 + r) * LDX] = cx1[r]; }
            if (TRAIN) {
                const float* xr = sX + opaque(lq * LDX + l15);
                const float* dr = sD + opaque(lq * LDH + 64 * w + l15);
#pragma unroll 2
                for (int ii = 0; ii < D32_M / 4; ++ii) {
                    const float xb = xr[4 * ii * LDX];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        accW1[t] = MFMA16(dr[4 * ii * LDH + 16 * t], xb, accW1[t]);
                }
            }
        }
        __syncthreads();
        STAMP32(10);
        // ---- J: dX = sum of the four k-slab partials -> global
        if (row0 + xi < P) {
            const float* pr = sPX + opaque(xi * LDX + xc);
            float2 o;
            o.x = (pr[0] + pr[D32_M * LDX]) + (pr[2 * D32_M * LDX] + pr[3 * D32_M * LDX]);
            o.y = (pr[1] + pr[D32_M * LDX + 1]) + (pr[2 * D32_M * LDX + 1] + pr[3 * D32_M * LDX + 1]);
            *reinterpret_cast<float2*>(a.dX + (size_t)(row0 + xi) * NL_C + xc) = o;
        }
        STAMP32(11);
        // (the next tile's phase A writes sX only; sPX / sD / sH1 are rewritten after its barriers)
    }

    if (tid < 64) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lossFs += __shfl_xor(lossFs, off); lossSdf += __shfl_xor(lossSdf, off); }
        if (tid == 0 && (lossFs != 0.0 || lossSdf != 0.0)) {
            atomicAdd(&a.dcounters[NLD_FS_SQ], lossFs); atomicAdd(&a.dcounters[NLD_SDF_SQ], lossSdf);
        }
    }
    if (TRAIN) {
        float* base = a.partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                base[NL_OFF_W1 + (64 * w + 16 * t + 4 * lq + r) * NL_C + l15] = accW1[t][r];
        aW3A += __shfl_xor(aW3A, 32); aW3B += __shfl_xor(aW3B, 32); aB2A += __shfl_xor(aB2A, 32); aB2B += __shfl_xor(aB2B, 32);
        aB1A += __shfl_xor(aB1A, 32); aB1B += __shfl_xor(aB1B, 32);
        if (lh == 0) {
            base[NL_OFF_W3 + colA] = aW3A; base[NL_OFF_W3 + colB] = aW3B; base[NL_OFF_B2 + colA] = aB2A; base[NL_OFF_B2 + colB] = aB2B;
            base[NL_OFF_B1 + colA] = aB1A; base[NL_OFF_B1 + colB] = aB1B;
        }
        if (tid < 64) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) aB3 += __shfl_xor(aB3, off);
            if (tid == 0) base[NL_OFF_B3] = aB3;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dW2 = dH2^T H1 over all samples (K = samples).  Persistent; per 64-sample tile: H1 = relu(X W1^T+b1)
// -> LDS, dH2[i][j] = mask(i,j) ? dsdf_i * w3_j : 0 -> LDS, then 256 MFMAs per wave into the 8
// persistent 32x32 accumulators of the wave's 32-row slab of dW2.  The next tile's inputs (X slice,
// dsdf, ReLU-mask word: one per thread) are prefetched into registers under the MFMA phase.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decoder_wgrad2(const NlLossScalars* __restrict__ lsp, const float* __restrict__ X,
                                                                    const float* __restrict__ params, const float* __restrict__ dsdf,
                                                                    const unsigned* __restrict__ relu2_mask, float* __restrict__ partials,
                                                                    int mask32)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * DEC_M * LDH + DEC_M * LDX + NL_W * NL_C + DEC_M];
    float* sH1 = lds; float* sD = lds + DEC_M * LDH; float* sX = sD + DEC_M * LDH; float* sW1 = sX + DEC_M * LDX;
    float* sdS = sW1 + NL_W * NL_C;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int col = 32 * w + l31;
    const int P = lsp->P;
    const int ntiles = (P + DEC_M - 1) / DEC_M;
    const float b1c = params[NL_OFF_B1 + col], w3c = params[NL_OFF_W3 + col];
    f32x16 accW2[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) accW2[t][r] = 0.f;
    for (int i = tid; i < NL_W * NL_C; i += DEC_THREADS) sW1[i] = params[NL_OFF_W1 + i];

    const int xe = tid * 2, xi = xe >> 4, xc = xe & 15;
    float2 xv = make_float2(0.f, 0.f); float pds = 0.f; unsigned pmk = 0u;
    auto prefetch = [&](int tile) {
        const int row0 = tile * DEC_M;
        xv = make_float2(0.f, 0.f); pds = 0.f; pmk = 0u;
        if (tile < ntiles) {
            if (row0 + xi < P) xv = *reinterpret_cast<const float2*>(X + (size_t)(row0 + xi) * NL_C + xc);
            if (tid < DEC_M && row0 + tid < P) pds = dsdf[row0 + tid];
            if (!mask32) {
                pmk = relu2_mask[(size_t)tile * DEC_THREADS + tid];
            } else {
                // masks written by k_decoder32: [32-row tile][256 threads], bits 0-15 = column 64w'+l31, bits 16-31 = +32.
                // this thread's column 32w+l31 lives in producer wave w>>1, low or high half by w&1; rows 0-31 / 32-63
                // of this 64-row tile are the 32-row tiles 2*tile and 2*tile+1.
                const int src = (w >> 1) * 64 + lane, sh = (w & 1) * 16;
                const unsigned lo = relu2_mask[(size_t)(2 * tile) * D32_THREADS + src];
                const unsigned hi = (2 * tile + 1) * D32_M < P ? relu2_mask[(size_t)(2 * tile + 1) * D32_THREADS + src] : 0u;
                pmk = ((lo >> sh) & 0xFFFFu) | (((hi >> sh) & 0xFFFFu) << 16);
            }
        }
    };
    prefetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        sX[xi * LDX + xc] = xv.x; sX[xi * LDX + xc + 1] = xv.y;
        if (tid < DEC_M) sdS[tid] = pds;
        const unsigned mw = pmk;
        __syncthreads();
        prefetch(tile + gridDim.x);
        {   // H1 -> LDS
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < NL_C / 2; ++kk) {
                const float bw = sW1[col * NL_C + 2 * kk + lh];
                c0 = MFMA32(sX[l31 * LDX + 2 * kk + lh], bw, c0); c1 = MFMA32(sX[(32 + l31) * LDX + 2 * kk + lh], bw, c1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = d32_row(r, lh);
                sH1[row * LDH + col] = fmaxf(c0[r] + b1c, 0.f); sH1[(32 + row) * LDH + col] = fmaxf(c1[r] + b1c, 0.f);
            }
        }
        {   // dH2 -> LDS: this lane's column, the same 32 rows its mask word describes
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = d32_row(r, lh);
                sD[row * LDH + col] = ((mw >> r) & 1u) ? sdS[row] * w3c : 0.f;
                sD[(32 + row) * LDH + col] = ((mw >> (16 + r)) & 1u) ? sdS[32 + row] * w3c : 0.f;
            }
        }
        __syncthreads();
        {
            const float* ap = sD + lh * LDH + col;
            const float* bp = sH1 + lh * LDH + l31;
#pragma unroll 2
            for (int ii = 0; ii < DEC_M / 2; ++ii) {
                const float av = ap[2 * ii * LDH];
#pragma unroll
                for (int t = 0; t < 8; ++t) accW2[t] = MFMA32(av, bp[2 * ii * LDH + 32 * t], accW2[t]);
            }
        }
        __syncthreads();
    }
    float* base = partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            base[NL_OFF_W2 + (32 * w + d32_row(r, lh)) * NL_W + 32 * t + l31] = accW2[t][r];
}

// ---------------------------------------------------------------------------------------------
// forward-only variant for dense SDF queries (mesh-time get_scores, render_helpers.py:96-153) and
// tests: sdf = decoder(X).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decoder_fwd(const float* __restrict__ X, const float* __restrict__ params,
                                                                 const float* __restrict__ W2T, int P, float* __restrict__ sdf)
{
    __shared__ __attribute__((aligned(16))) float lds[DEC_M * LDH + DEC_M * LDX + DEC_M];
    float* sH1 = lds; float* sX = lds + DEC_M * LDH; float* sS = sX + DEC_M * LDX;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int col = 32 * w + l31;
    const float* W1 = params + NL_OFF_W1;
    const float b1c = params[NL_OFF_B1 + col], b2c = params[NL_OFF_B2 + col], w3c = params[NL_OFF_W3 + col], b3 = params[NL_OFF_B3];
    const int ntiles = (P + DEC_M - 1) / DEC_M;
    if (tid < DEC_M) sS[tid] = 0.f;
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * DEC_M;
        {
            const int e = tid * 2, i = e >> 4, c = e & 15;
            float2 v = make_float2(0.f, 0.f);
            if (row0 + i < P) v = *reinterpret_cast<const float2*>(X + (size_t)(row0 + i) * NL_C + c);
            sX[i * LDX + c] = v.x; sX[i * LDX + c + 1] = v.y;
        }
        __syncthreads();
        {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < NL_C / 2; ++kk) {
                const float bw = W1[col * NL_C + 2 * kk + lh];
                c0 = MFMA32(sX[l31 * LDX + 2 * kk + lh], bw, c0); c1 = MFMA32(sX[(32 + l31) * LDX + 2 * kk + lh], bw, c1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = d32_row(r, lh);
                sH1[row * LDH + col] = fmaxf(c0[r] + b1c, 0.f); sH1[(32 + row) * LDH + col] = fmaxf(c1[r] + b1c, 0.f);
            }
        }
        __syncthreads();
        {
            f32x16 h0, h1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
            const float* bp = W2T + lh * NL_W + col;
            const float* ap0 = sH1 + l31 * LDH + lh; const float* ap1 = sH1 + (32 + l31) * LDH + lh;
#pragma unroll 8
            for (int kk = 0; kk < NL_W / 2; ++kk) {
                const float bw = bp[(size_t)kk * 2 * NL_W];
                h0 = MFMA32(ap0[2 * kk], bw, h0); h1 = MFMA32(ap1[2 * kk], bw, h1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p0 = fmaxf(h0[r] + b2c, 0.f) * w3c, p1 = fmaxf(h1[r] + b2c, 0.f) * w3c;
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) { p0 += __shfl_xor(p0, off); p1 += __shfl_xor(p1, off); }
                if (l31 == 0) { const int row = d32_row(r, lh); atomicAdd(&sS[row], p0); atomicAdd(&sS[32 + row], p1); }
            }
        }
        __syncthreads();
        if (tid < DEC_M) { if (row0 + tid < P) sdf[row0 + tid] = sS[tid] + b3; sS[tid] = 0.f; }
        __syncthreads();
    }
}

// sum per-workgroup partial slabs: out[i] = sum_b partials[b][i]; the W2 block is written by k_decoder_wgrad2's
// workgroups (nslabs_w2), every other parameter by the fused decoder kernel's workgroups (nslabs)
__global__ void k_reduce_partials(const float* __restrict__ partials, int nslabs, int nslabs_w2, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nb = (i >= NL_OFF_W2 && i < NL_OFF_B2 && n == NL_DEC_PARAMS) ? nslabs_w2 : nslabs;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partials[(size_t)b * n + i];
    out[i] = s;
}

// MFMA layout self-test: D = A(32x2) B(2x32) and D = A(16x4) B(4x16) written row-major using the lane
// maps this file assumes (cdna_hip_programming.md section 3).
__global__ void k_mfma_selftest(const float* A32, const float* B32, float* D32, const float* A16, const float* B16, float* D16)
{
    const int lane = threadIdx.x & 63;
    f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = MFMA32(A32[(lane & 31) * 2 + (lane >> 5)], B32[(lane >> 5) * 32 + (lane & 31)], c);
    for (int r = 0; r < 16; ++r) D32[d32_row(r, lane >> 5) * 32 + (lane & 31)] = c[r];
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    e = MFMA16(A16[(lane & 15) * 4 + (lane >> 4)], B16[(lane >> 4) * 16 + (lane & 15)], e);
    for (int r = 0; r < 4; ++r) D16[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = e[r];
}

static long long* g_dec_dbg = nullptr;
static int g_dec_stagger = 29000;        // k_decoder32: ~half of a tile's stand-alone period
static int g_dec_variant = 0;            // 0: 64-sample tiles, 1 workgroup per CU; 1: 32-sample tiles, 2 workgroups per CU

extern "C" {

/* profiling aid: device buffer of 256 int64 receiving per-phase s_memtime stamps (NULL disables) */
int nl_decoder_set_debug_buffer(void* dbg) { g_dec_dbg = (long long*)dbg; return NL_OK; }

/* decoder kernel variant: 0 = k_decoder (64-sample tiles, 512 threads, 1 workgroup/CU; default), 1 = k_decoder32 (32-sample tiles,
 * 256 threads, 2 workgroups/CU).  Both produce the same results; kept selectable for A/B measurements. */
int nl_decoder_set_variant(int v) { if (v < 0 || v > 1) return NL_ERR_INVALID_ARG; g_dec_variant = v; return NL_OK; }
int nl_decoder_get_variant(void) { return g_dec_variant; }
/* start offset (cycles) between the two workgroups that share a CU in variant 1; 0 disables it (measurement aid) */
int nl_decoder_set_stagger(int cycles) { if (cycles < 0 || cycles > 10000000) return NL_ERR_INVALID_ARG; g_dec_stagger = cycles; return NL_OK; }

int nl_decoder_grid_hint(void)
{
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

int nl_decoder_fwd_bwd(const void* loss_scalars, const float* X, const float* params, const float* W2T,
                       const int* s_ray, const float* s_depth, const float* cos_gt, const float* gt_dist,
                       float* sdf, float* dsdf, float* dX, float* partials, unsigned* relu2_mask, int nslabs, int train_decoder,
                       int* counters, void* stream)
{
    if (!loss_scalars || !X || !params || !W2T || !s_ray || !s_depth || !cos_gt || !gt_dist || !sdf || !dsdf || !dX || !counters)
        return NL_ERR_INVALID_ARG;
    if (nslabs <= 0 || (train_decoder && (!partials || !relu2_mask))) return NL_ERR_INVALID_ARG;
    DecArgs a;
    a.ls = (const NlLossScalars*)loss_scalars; a.X = X; a.params = params; a.W2T = W2T; a.s_ray = s_ray; a.s_depth = s_depth;
    a.cos_gt = cos_gt; a.gt_dist = gt_dist; a.sdf = sdf; a.dsdf = dsdf; a.dX = dX; a.partials = partials;
    a.relu2_mask = relu2_mask;
    a.dcounters = (double*)(counters + NL_CNT_INTS);
    a.dbg = g_dec_dbg;
    a.stagger = g_dec_stagger;
    if (g_dec_variant == 1) {       // nslabs = workgroups = 2 x CUs
        if (train_decoder) hipLaunchKernelGGL(k_decoder32<true>, dim3(nslabs), dim3(D32_THREADS), 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL(k_decoder32<false>, dim3(nslabs), dim3(D32_THREADS), 0, (hipStream_t)stream, a);
    } else {
        if (train_decoder) hipLaunchKernelGGL(k_decoder<true>, dim3(nslabs), dim3(DEC_THREADS), 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL(k_decoder<false>, dim3(nslabs), dim3(DEC_THREADS), 0, (hipStream_t)stream, a);
    }
    NL_LAUNCH_CHECK();
    return NL_OK;
}

// dW2 slab of the decoder weight gradient (second persistent kernel; needs nl_decoder_fwd_bwd's dsdf + relu2_mask)
int nl_decoder_wgrad2(const void* loss_scalars, const float* X, const float* params, const float* dsdf, const unsigned* relu2_mask,
                      float* partials, int nslabs, void* stream)
{
    if (!loss_scalars || !X || !params || !dsdf || !relu2_mask || !partials || nslabs <= 0) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_decoder_wgrad2, dim3(nslabs), dim3(DEC_THREADS), 0, (hipStream_t)stream, (const NlLossScalars*)loss_scalars, X,
                       params, dsdf, relu2_mask, partials, g_dec_variant == 1 ? 1 : 0);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_decoder_forward(const float* X, const float* params, const float* W2T, int P, float* sdf, int nblocks, void* stream)
{
    if (!X || !params || !W2T || !sdf || P < 0 || nblocks <= 0) return NL_ERR_INVALID_ARG;
    if (P == 0) return NL_OK;
    hipLaunchKernelGGL(k_decoder_fwd, dim3(nblocks), dim3(DEC_THREADS), 0, (hipStream_t)stream, X, params, W2T, P, sdf);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_reduce_partials(const float* partials, int nslabs, int nslabs_w2, int n, float* out, void* stream)
{
    if (!partials || !out || nslabs <= 0 || nslabs_w2 <= 0 || n <= 0) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nl_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, partials, nslabs, nslabs_w2, n, out);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_mfma_selftest(const float* A32, const float* B32, float* D32, const float* A16, const float* B16, float* D16, void* stream)
{
    hipLaunchKernelGGL(k_mfma_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, A32, B32, D32, A16, B16, D16);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

}  // extern "C"
