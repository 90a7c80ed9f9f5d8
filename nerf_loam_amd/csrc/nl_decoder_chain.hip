// nl_decoder_chain.hip -- the SDF decoder as a REGISTER-CHAINED kernel: the second kernel family of the decoder (gemm mode 3).
//
// Same arithmetic contract as nl_decoder.hip (fp32 values, fp32 accumulation, the 256-deep contractions on the bf16 matrix cores
// as exact-product splits; reference: src/variations/lidar.py:109-131, src/criterion.py:59-100, autograd), different dataflow:
//
//   * nl_decoder.hip keeps the activations of a 64-sample tile in LDS, gives every wave 32 output COLUMNS, and synchronises the
//     eight waves of a workgroup five times per tile; between the barriers the matrix pipes idle while the waves split H1 into
//     operand planes, store 2-byte elements, reduce row sums with 31 shuffles, ...: 47.6 k cycles per tile against 30.7 k of MFMA.
//   * here a wave owns 32 SAMPLES end to end.  All GEMMs are issued TRANSPOSED (D^T = B^T A^T: the weights are the A operand,
//     the activations the B operand), so an accumulator tile holds, per lane, one sample and 16 hidden units - and those 16
//     registers ARE two B-operand fragments of the next layer's MFMA (nl_chain_slot in nl_device_math.h: the weight planes
//     are stored in that slot order by k_prepare_w2a).  H1, H2, the ReLU masks and dH2 never leave the register file; the row
//     sum of the output layer is 16 in-lane FMAs and one shuffle; dL/dsdf is a lane constant.  No workgroup barrier, no
//     activation traffic through LDS; the only LDS round trip is the 32x32 dH1 tile a wave transposes for itself to contract over
//     samples (dX, dW1: K = 16 layers on the fp32 matrix cores, 16x16x4).
//   * one wave per SIMD (256-thread workgroups, 512 registers per lane: three operand planes of H1 are 192 of them).  The weight
//     planes stream from L2 into a register ring; the micro-benchmark of that stream is scripts/micro/chain_stream.hip.
//
// Weight gradients: dW1 / db1 / db3 accumulate in registers over all tiles of a wave and leave once per workgroup.  dW2 stays in
// its own kernel (k_decoder_wgrad2_x, "natural" mask format below), which now also accumulates g[n] = sum_i m2(i,n) dsdf_i,
// and two identities give the rest without keeping H2:
//     db2[n] = w3_n g[n],    dW3[n] = sum_i dsdf_i h2[i][n] = sum_k W2[n][k] G[n][k] + b2[n] g[n],   G = dW2 / w3 (raw accumulators)
// (h2 = m2 (H1 W2^T + b2)); nl_decoder_reduce applies them while summing the per-workgroup slabs.
#include "nl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#define CH_THREADS 256
#define CH_W1_STRIDE 20                                 // floats per W1 row in LDS: conflict-free ds_read_b128 of half a row per lane
#define CH_T_STRIDE 36                                  // floats per sample row of a wave's dH1 transposition tile
#define CH_RING 4                                       // k-steps of weight fragments in flight (divides 16: static ring slots)
#define CH_ACC_FLOATS (16 * 64 * 4 + 16 * 64)            // per wave: dW1 tiles [kt][ks][lane][4] + db1 partials [kt][ks][lane]
#define CH_PLANE_BYTES (NL_W * NL_W * 2)
#define CH_WS_W2A_OFF (NL_W * NL_W + 6 * NL_W * NL_W / 2)     // floats into the decoder weight workspace (nl_optim.hip)
#define CH_WS_W2XA_OFF (NL_W * NL_W + 9 * NL_W * NL_W / 2)

struct ChainArgs {
    const NlLossScalars* ls;    // NULL: forward only over P samples
    int P;
    const float* X; const float* params; const float* ws;
    const int* s_ray; const float* s_depth; const float* cos_gt; const float* gt_dist;
    float* sdf; float* dsdf; float* dX;
    float* partials;            // [gridDim.x][NL_DEC_PARAMS]: W1, b1, b3 regions (train)
    unsigned* relu2_nat;        // [ceil(P/32)][256]: bit b of word (tile, unit) = ReLU of H2[32 tile + b][unit] (train)
    double* dcounters;
    long long* dbg;             // optional [16 tiles][16] shader-clock stamps of workgroup 0 / wave 0 (profiling aid)
};
#define CH_STAMP(slot)                                                                          \
    do { if (a.dbg && blockIdx.x == 0 && tid == 0 && tile_no < 16) a.dbg[tile_no * 16 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ uint4 ch_bload4(rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ unsigned ch_pack_hi16(float a, float b)
{
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);      // (lo = bf16 of a, hi = bf16 of b), truncating
}
__device__ __forceinline__ float ch_trunc(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

// MODE 0: forward only (sdf).  1: forward + loss gradient + dgrad + dX (frozen decoder: tracking, mapping after freeze_frame).
// 2: 1 + decoder weight gradients (dW1, db1, db3 here; the ReLU words of H2 for the dW2 kernel).
// NP: partial products of the forward GEMM (9 = exact, 6 = without lo x lo, lo x mid, mid x lo: gemm mode 2's arithmetic)
template <int MODE, int NP>
__global__ __launch_bounds__(CH_THREADS, 1) void k_decoder_chain(ChainArgs a)
{
    __shared__ __attribute__((aligned(16))) float sW1[NL_W * CH_W1_STRIDE];
    __shared__ __attribute__((aligned(16))) float sTab[3 * 256];          // b1 | b2 | w3 in accumulator order [tile][lh][r]
    __shared__ __attribute__((aligned(16))) uint4 sLut[256];              // byte -> 8 bf16 (1.0 where the bit is set)
    __shared__ __attribute__((aligned(16))) float sT[4 * 32 * CH_T_STRIDE];
    // MODE 2: every wave's running dW1 (16 accumulator tiles of 16x16) and db1 partials live HERE between tiles - 80 registers
    // per lane otherwise, on top of the three H1 operand planes - in the lane-major order a ds_read_b128 / ds_write_b128 wants
    __shared__ __attribute__((aligned(16))) float sAccW[MODE == 2 ? 4 * CH_ACC_FLOATS : 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const float* params = a.params;
    for (int i = tid; i < NL_W * NL_C; i += CH_THREADS) sW1[(i >> 4) * CH_W1_STRIDE + (i & 15)] = params[NL_OFF_W1 + i];
    {
        const int u = 32 * (tid >> 5) + nl_chain_unit(tid & 15, (tid >> 4) & 1);
        sTab[tid] = params[NL_OFF_B1 + u]; sTab[256 + tid] = params[NL_OFF_B2 + u]; sTab[512 + tid] = params[NL_OFF_W3 + u];
        uint4 e;
        e.x = ((tid & 1) ? 0x3F80u : 0u) | ((tid & 2) ? 0x3F800000u : 0u);
        e.y = ((tid & 4) ? 0x3F80u : 0u) | ((tid & 8) ? 0x3F800000u : 0u);
        e.z = ((tid & 16) ? 0x3F80u : 0u) | ((tid & 32) ? 0x3F800000u : 0u);
        e.w = ((tid & 64) ? 0x3F80u : 0u) | ((tid & 128) ? 0x3F800000u : 0u);
        sLut[tid] = e;
    }
    __syncthreads();
    NlLossScalars ls;
    int P = a.P;
    if (MODE >= 1) { ls = *a.ls; P = ls.P; }
    const float b3 = params[NL_OFF_B3];
    const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ws + CH_WS_W2A_OFF), 0, 3 * CH_PLANE_BYTES, 0x00020000);
    const rsrc_t rsXA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ws + CH_WS_W2XA_OFF), 0, 3 * CH_PLANE_BYTES, 0x00020000);
    const int voff = lane * 16;
    float* sTw = sT + w * (32 * CH_T_STRIDE);

    // weight-gradient accumulators (MODE 2)
    float aB3 = 0.f;
    double lossFs = 0.0, lossSdf = 0.0;
    float* sAw = sAccW + (MODE == 2 ? w * CH_ACC_FLOATS : 0);
    if (MODE == 2) {
        for (int i = lane; i < CH_ACC_FLOATS; i += 64) sAw[i] = 0.f;
        __builtin_amdgcn_wave_barrier();
    }
    // ReLU-word assembly (MODE 2): lane j < 32 collects the word of unit j of the current 32-unit tile
    const int mw_r = (l31 & 3) + 4 * (l31 >> 3), mw_hi = (l31 >> 2) & 1;

    const int ntiles = (P + 31) >> 5;
    // inputs of a tile: this lane's sample (channels 8 lh .. 8 lh + 7) and its loss geometry; the NEXT tile's are fetched under the
    // current tile's backward phase
    float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0;
    float ncz = 0.f, ncd = 0.f;
    auto fetch_inputs = [&](int tile) {
        const int gg = (tile << 5) + l31;
        nx0 = make_float4(0.f, 0.f, 0.f, 0.f); nx1 = nx0; ncz = 0.f; ncd = 0.f;
        if (tile < ntiles && gg < P) {
            const float4* xp = reinterpret_cast<const float4*>(a.X + (size_t)gg * NL_C + 8 * lh);
            nx0 = xp[0]; nx1 = xp[1];
            if (MODE >= 1) {
                const int ray = a.s_ray[gg];
                ncz = a.s_depth[gg] * a.cos_gt[ray]; ncd = a.gt_dist[ray];
            }
        }
    };
    fetch_inputs(blockIdx.x * 4 + w);
    int tile_no = 0;
    for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += gridDim.x * 4, ++tile_no) {
        CH_STAMP(0);
        const int row0 = tile << 5, g = row0 + l31;
        const bool live = g < P;
        const float xf[8] = {nx0.x, nx0.y, nx0.z, nx0.w, nx1.x, nx1.y, nx1.z, nx1.w};
        const float cz = ncz, cd = ncd;
        if (MODE == 0) fetch_inputs(tile + gridDim.x * 4);
        // first k-steps of the forward weight stream: in flight under layer 1
        uint4 aq[CH_RING][3];
#pragma unroll
        for (int j = 0; j < CH_RING - 1; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) aq[j][p] = ch_bload4(rsA, voff, p * CH_PLANE_BYTES + j * 1024);

        CH_STAMP(1);
        // ---------------- layer 1: H1^T = relu(W1 X^T + b1), split into three bf16 operand planes (registers) ----------------
        uint4 hb[16][3];
        unsigned m1w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ut = 0; ut < 8; ++ut) {
            f32x16 c;
            {
                const float4* tb = reinterpret_cast<const float4*>(sTab + ut * 32 + lh * 16);
                const float4 t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
                c[0] = t0.x; c[1] = t0.y; c[2] = t0.z; c[3] = t0.w; c[4] = t1.x; c[5] = t1.y; c[6] = t1.z; c[7] = t1.w;
                c[8] = t2.x; c[9] = t2.y; c[10] = t2.z; c[11] = t2.w; c[12] = t3.x; c[13] = t3.y; c[14] = t3.z; c[15] = t3.w;
            }
            const float4* wr = reinterpret_cast<const float4*>(sW1 + (32 * ut + l31) * CH_W1_STRIDE + 8 * lh);
            const float4 w0 = wr[0], w1 = wr[1];
            c = MFMA32(w0.x, xf[0], c); c = MFMA32(w0.y, xf[1], c); c = MFMA32(w0.z, xf[2], c); c = MFMA32(w0.w, xf[3], c);
            c = MFMA32(w1.x, xf[4], c); c = MFMA32(w1.y, xf[5], c); c = MFMA32(w1.z, xf[6], c); c = MFMA32(w1.w, xf[7], c);
            unsigned bits = 0u, hi[8], mid[8], lo[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float h0 = fmaxf(c[2 * q], 0.f), h1 = fmaxf(c[2 * q + 1], 0.f);
                bits |= (h0 > 0.f ? (1u << (2 * q)) : 0u) | (h1 > 0.f ? (2u << (2 * q)) : 0u);
                hi[q] = ch_pack_hi16(h0, h1);
                const float r0 = h0 - ch_trunc(h0), r1 = h1 - ch_trunc(h1);
                mid[q] = ch_pack_hi16(r0, r1);
                lo[q] = ch_pack_hi16(r0 - ch_trunc(r0), r1 - ch_trunc(r1));
            }
            hb[2 * ut][0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); hb[2 * ut + 1][0] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            hb[2 * ut][1] = make_uint4(mid[0], mid[1], mid[2], mid[3]); hb[2 * ut + 1][1] = make_uint4(mid[4], mid[5], mid[6], mid[7]);
            hb[2 * ut][2] = make_uint4(lo[0], lo[1], lo[2], lo[3]); hb[2 * ut + 1][2] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
            m1w[ut >> 1] |= bits << (16 * (ut & 1));
        }

        CH_STAMP(2);
        // ---------------- layer 2 + output layer: s = w3 . relu(W2 H1 + b2) + b3, H2 never stored ----------------
        float spart = 0.f;
        unsigned m2w[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int nt = 0; nt < 8; ++nt) {
            f32x16 h;
            {
                const float4* tb = reinterpret_cast<const float4*>(sTab + 256 + nt * 32 + lh * 16);
                const float4 t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
                h[0] = t0.x; h[1] = t0.y; h[2] = t0.z; h[3] = t0.w; h[4] = t1.x; h[5] = t1.y; h[6] = t1.z; h[7] = t1.w;
                h[8] = t2.x; h[9] = t2.y; h[10] = t2.z; h[11] = t2.w; h[12] = t3.x; h[13] = t3.y; h[14] = t3.z; h[15] = t3.w;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int so = ((nt * 16 + s + CH_RING - 1) & 127) * 1024;      // (wraps at the end: three harmless extra loads)
#pragma unroll
                for (int p = 0; p < 3; ++p) aq[(s + CH_RING - 1) % CH_RING][p] = ch_bload4(rsA, voff, p * CH_PLANE_BYTES + so);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                    for (int pb = 0; pb < 3; ++pb) {
                        if (NP == 6 && pa + pb > 2) continue;
                        h = MFMA_BF16(__builtin_bit_cast(bf16x8, aq[s % CH_RING][pa]), __builtin_bit_cast(bf16x8, hb[s][pb]), h);
                    }
            }
            const float4* tw = reinterpret_cast<const float4*>(sTab + 512 + nt * 32 + lh * 16);
            const float4 w0 = tw[0], w1 = tw[1], w2 = tw[2], w3 = tw[3];
            const float wv[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
            unsigned bits = 0u, mword = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = fmaxf(h[r], 0.f);
                const bool on = hv > 0.f;
                bits |= on ? (1u << r) : 0u;
                spart = fmaf(hv, wv[r], spart);
                if (MODE == 2) {
                    const unsigned long long bal = __ballot(on);
                    const unsigned pick = mw_hi ? (unsigned)(bal >> 32) : (unsigned)bal;
                    mword = (mw_r == r) ? pick : mword;
                }
            }
            // 128-bit shift register: after the 8th tile, tile nt sits in bits [16 nt, 16 nt + 16)
            m2w[0] = (m2w[0] >> 16) | (m2w[1] << 16); m2w[1] = (m2w[1] >> 16) | (m2w[2] << 16);
            m2w[2] = (m2w[2] >> 16) | (m2w[3] << 16); m2w[3] = (m2w[3] >> 16) | (bits << 16);
            if (MODE == 2 && lh == 0) a.relu2_nat[(size_t)tile * NL_W + 32 * nt + l31] = mword;
        }
        CH_STAMP(3);
        const float sv = (spart + __shfl_xor(spart, 32)) + b3;
        if (MODE == 0) {
            if (live && lh == 0) a.sdf[g] = sv;
            continue;
        }
        // ---------------- loss gradient (criterion.py): a lane constant ----------------
        float ds = 0.f;
        if (live) {
            bool f, m;
            nl_loss_masks(cz, cd, ls.tau, ls.max_depth, &f, &m);
            float q1, q2;
            ds = nl_loss_grad(sv, cz, cd, f, m, ls, &q1, &q2);
            if (lh == 0) {
                a.sdf[g] = sv; a.dsdf[g] = ds;
                lossFs += (double)q1; lossSdf += (double)q2;
                if (MODE == 2) aB3 += ds;
            }
        }
        CH_STAMP(4);
        // ---------------- dgrad: dH1^T = ((w3 W2)^T mask^T) * dsdf * [H1 > 0]; layer-1 backward per 32-unit tile ----------------
        fetch_inputs(tile + gridDim.x * 4);                     // the next tile's inputs arrive under this tile's backward
        uint4 mf[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) mf[s] = sLut[(m2w[s >> 2] >> (8 * (s & 3))) & 0xFFu];
#pragma unroll
        for (int j = 0; j < CH_RING - 1; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) aq[j][p] = ch_bload4(rsXA, voff, p * CH_PLANE_BYTES + j * 1024);
        float xw[8];
        if (MODE == 2) {
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) { const int gi = row0 + 4 * ii + lq; xw[ii] = gi < P ? a.X[(size_t)gi * NL_C + l15] : 0.f; }
        }
        f32x4 dxa[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) for (int r = 0; r < 4; ++r) dxa[sub][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            f32x16 gacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int so = ((kt * 16 + s + CH_RING - 1) & 127) * 1024;
#pragma unroll
                for (int p = 0; p < 3; ++p) aq[(s + CH_RING - 1) % CH_RING][p] = ch_bload4(rsXA, voff, p * CH_PLANE_BYTES + so);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pa = 0; pa < 3; ++pa)
                    gacc = MFMA_BF16(__builtin_bit_cast(bf16x8, aq[s % CH_RING][pa]), __builtin_bit_cast(bf16x8, mf[s]), gacc);
            }
            if (kt == 0) CH_STAMP(5);                            // after the first tile's 48 MFMAs: stream start-up + one dgrad tile
            const unsigned b1bits = (m1w[kt >> 1] >> (16 * (kt & 1))) & 0xFFFFu;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = gacc[4 * q] * (((b1bits >> (4 * q)) & 1u) ? ds : 0.f);
                v.y = gacc[4 * q + 1] * (((b1bits >> (4 * q + 1)) & 1u) ? ds : 0.f);
                v.z = gacc[4 * q + 2] * (((b1bits >> (4 * q + 2)) & 1u) ? ds : 0.f);
                v.w = gacc[4 * q + 3] * (((b1bits >> (4 * q + 3)) & 1u) ? ds : 0.f);
                *reinterpret_cast<float4*>(sTw + l31 * CH_T_STRIDE + 8 * q + 4 * lh) = v;       // column = unit nl_chain_unit(r, lh)
            }
            __builtin_amdgcn_wave_barrier();                     // same-wave LDS write -> read (in order in hardware; pins the compiler)
            // dX[i][c] += sum_k dH1[i][32 kt + k] W1[32 kt + k][c]   (16x16x4 fp32: A = dH1 rows of one 16-sample half, B = W1 rows)
            {
                const float* ta = sTw + l15 * CH_T_STRIDE + lq;
                const float* wb = sW1 + (32 * kt + lq) * CH_W1_STRIDE + l15;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float bw = wb[4 * kk * CH_W1_STRIDE];
                    dxa[0] = MFMA16(ta[4 * kk], bw, dxa[0]);
                    dxa[1] = MFMA16(ta[16 * CH_T_STRIDE + 4 * kk], bw, dxa[1]);
                }
            }
            if (MODE == 2) {
                // dW1[32 kt + k][c] += sum_i dH1[i][32 kt + k] X[i][c]   (A = dH1 columns of one 16-unit half, B = X rows)
                const float* ta = sTw + lq * CH_T_STRIDE + l15;
                f32x4* aw = reinterpret_cast<f32x4*>(sAw + (2 * kt * 64 + lane) * 4);
                float* ab = sAw + 16 * 64 * 4 + 2 * kt * 64 + lane;
                f32x4 w0 = aw[0], w1 = aw[64];
                float b0 = ab[0], b1 = ab[64];
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
                    const float a0 = ta[4 * ii * CH_T_STRIDE], a1 = ta[4 * ii * CH_T_STRIDE + 16];
                    w0 = MFMA16(a0, xw[ii], w0);
                    w1 = MFMA16(a1, xw[ii], w1);
                    b0 += a0; b1 += a1;
                }
                aw[0] = w0; aw[64] = w1; ab[0] = b0; ab[64] = b1;
            }
            __builtin_amdgcn_wave_barrier();
            if (kt == 0) CH_STAMP(6);                            // one tile's layer-1 backward
        }
        CH_STAMP(7);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = row0 + 16 * sub + 4 * lq + r;
                if (gi < P) a.dX[(size_t)gi * NL_C + l15] = dxa[sub][r];
            }
    }
    if (MODE == 0) return;

    // ---------------- loss sums ----------------
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { lossFs += __shfl_xor(lossFs, off); lossSdf += __shfl_xor(lossSdf, off); }
    if (lane == 0 && (lossFs != 0.0 || lossSdf != 0.0)) { atomicAdd(&a.dcounters[NLD_FS_SQ], lossFs); atomicAdd(&a.dcounters[NLD_SDF_SQ], lossSdf); }
    if (MODE != 2) return;

    // ---------------- weight-gradient slab of this workgroup: the four waves' accumulators summed in LDS, one wave at a time ----------
    __syncthreads();                                             // every wave is done with sW1 / sTab
    float* sAcc = sW1;                                           // [4096] dW1 | [256] db1 | [1] db3  (20 KB region)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) aB3 += __shfl_xor(aB3, off);
    for (int round = 0; round < 4; ++round) {
        if (w == round) {
#pragma unroll 1
            for (int t = 0; t < 16; ++t) {                          // t = 2 kt + ks: units 16 t .. 16 t + 15
                const f32x4 v = *reinterpret_cast<const f32x4*>(sAw + (t * 64 + lane) * 4);
                float b = sAw[16 * 64 * 4 + t * 64 + lane];
                b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (16 * t + 4 * lq + r) * NL_C + l15;
                    sAcc[idx] = (round ? sAcc[idx] : 0.f) + v[r];
                }
                if (lq == 0) { const int idx = 4096 + 16 * t + l15; sAcc[idx] = (round ? sAcc[idx] : 0.f) + b; }
            }
            if (lane == 0) sAcc[4352] = (round ? sAcc[4352] : 0.f) + aB3;
        }
        __syncthreads();
    }
    float* base = a.partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
    for (int i = tid; i < NL_W * NL_C; i += CH_THREADS) base[NL_OFF_W1 + i] = sAcc[i];
    base[NL_OFF_B1 + tid] = sAcc[4096 + tid];
    if (tid == 0) base[NL_OFF_B3] = sAcc[4352];
}

// ---------------------------------------------------------------------------------------------
// Sum of the per-workgroup slabs for the chained family.  Slab contents: W1, b1, b3 = gradients (k_decoder_chain); W2 = RAW
// accumulators G[n][k] = sum_i m2(i,n) dsdf_i H1[i][k] and b2 = raw g[n] = sum_i m2(i,n) dsdf_i (k_decoder_wgrad2_x<natural>); W3
// unused.  Output: dW2 = w3_n G, db2 = w3_n g, dW3[n] = sum_k W2[n][k] G[n][k] + b2[n] g[n] (the identities in the file header).
// Blocks 0..255: row n of the W2 block (+ b2[n], W3[n]); blocks 256..: 256 elements each of the rest (W1, b1, b3).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_chain(const float* __restrict__ partials, int nslabs, const float* __restrict__ params,
                                                      float* __restrict__ out)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b < NL_W) {
        const int n = b, i = NL_OFF_W2 + n * NL_W + tid;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float* p = partials + i;
        int sl = 0;
        for (; sl + 3 < nslabs; sl += 4) {
            s0 += p[(size_t)sl * NL_DEC_PARAMS]; s1 += p[(size_t)(sl + 1) * NL_DEC_PARAMS];
            s2 += p[(size_t)(sl + 2) * NL_DEC_PARAMS]; s3 += p[(size_t)(sl + 3) * NL_DEC_PARAMS];
        }
        for (; sl < nslabs; ++sl) s0 += p[(size_t)sl * NL_DEC_PARAMS];
        const float G = (s0 + s1) + (s2 + s3);
        float gsum = 0.f;                                           // raw g[n]: every thread sums a strided share of the slabs
        for (int q = tid; q < nslabs; q += 256) gsum += partials[(size_t)q * NL_DEC_PARAMS + NL_OFF_B2 + n];
        float dot = params[i] * G;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { dot += __shfl_xor(dot, off); gsum += __shfl_xor(gsum, off); }
        __shared__ float rd[4], rg[4];
        if ((tid & 63) == 0) { rd[tid >> 6] = dot; rg[tid >> 6] = gsum; }
        __syncthreads();
        const float w3n = params[NL_OFF_W3 + n];
        out[i] = w3n * G;
        if (tid == 0) {
            const float gn = (rg[0] + rg[1]) + (rg[2] + rg[3]);
            out[NL_OFF_B2 + n] = w3n * gn;
            out[NL_OFF_W3 + n] = ((rd[0] + rd[1]) + (rd[2] + rd[3])) + params[NL_OFF_B2 + n] * gn;
        }
        return;
    }
    const int e = (b - NL_W) * 256 + tid;                           // W1 (4096), b1 (256), b3 (1)
    int i = -1;
    if (e < NL_OFF_W2) i = e;
    else if (e == NL_OFF_W2) i = NL_OFF_B3;
    if (i < 0) return;
    float s0 = 0.f, s1 = 0.f;
    const float* p = partials + i;
    int sl = 0;
    for (; sl + 1 < nslabs; sl += 2) { s0 += p[(size_t)sl * NL_DEC_PARAMS]; s1 += p[(size_t)(sl + 1) * NL_DEC_PARAMS]; }
    if (sl < nslabs) s0 += p[(size_t)sl * NL_DEC_PARAMS];
    out[i] = s0 + s1;
}

extern "C" {

int nl_decoder_chain_fwd_bwd(const void* loss_scalars, const float* X, const float* params, const float* ws, const int* s_ray,
                             const float* s_depth, const float* cos_gt, const float* gt_dist, float* sdf, float* dsdf, float* dX,
                             float* partials, unsigned* relu2_nat, int nslabs, int train_decoder, int six_products, int* counters,
                             void* dbg, void* stream)
{
    ChainArgs a;
    a.ls = (const NlLossScalars*)loss_scalars; a.P = 0; a.X = X; a.params = params; a.ws = ws; a.s_ray = s_ray; a.s_depth = s_depth;
    a.cos_gt = cos_gt; a.gt_dist = gt_dist; a.sdf = sdf; a.dsdf = dsdf; a.dX = dX; a.partials = partials; a.relu2_nat = relu2_nat;
    a.dcounters = (double*)(counters + NL_CNT_INTS);
    a.dbg = (long long*)dbg;
    const dim3 g(nslabs), b(CH_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (train_decoder) {
        if (six_products) hipLaunchKernelGGL((k_decoder_chain<2, 6>), g, b, 0, st, a);
        else              hipLaunchKernelGGL((k_decoder_chain<2, 9>), g, b, 0, st, a);
    } else {
        if (six_products) hipLaunchKernelGGL((k_decoder_chain<1, 6>), g, b, 0, st, a);
        else              hipLaunchKernelGGL((k_decoder_chain<1, 9>), g, b, 0, st, a);
    }
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_decoder_chain_forward(const float* X, const float* params, const float* ws, int P, float* sdf, int nblocks, int six_products,
                             void* stream)
{
    ChainArgs a;
    a.ls = nullptr; a.P = P; a.X = X; a.params = params; a.ws = ws; a.s_ray = nullptr; a.s_depth = nullptr; a.cos_gt = nullptr;
    a.gt_dist = nullptr; a.sdf = sdf; a.dsdf = nullptr; a.dX = nullptr; a.partials = nullptr; a.relu2_nat = nullptr; a.dcounters = nullptr;
    a.dbg = nullptr;
    const int need = nl_div_up(nl_div_up(P, 32), 4);
    const dim3 g(nblocks < need ? nblocks : need), b(CH_THREADS);
    if (six_products) hipLaunchKernelGGL((k_decoder_chain<0, 6>), g, b, 0, (hipStream_t)stream, a);
    else              hipLaunchKernelGGL((k_decoder_chain<0, 9>), g, b, 0, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_decoder_chain_reduce(const float* partials, int nslabs, const float* params, float* grad_out, void* stream)
{
    const int nb = NL_W + nl_div_up(NL_OFF_W2 + 1, 256);
    hipLaunchKernelGGL(k_reduce_chain, dim3(nb), dim3(256), 0, (hipStream_t)stream, partials, nslabs, params, grad_out);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

}  // extern "C"
