// nl_decoder_pair.hip -- the fused decoder kernel of nl_decoder.hip (same arithmetic: fp32 values, exact-product bf16 GEMMs) with TWO
// TILES IN FLIGHT per workgroup: gemm mode 5.
//
// nl_decoder.hip runs eight waves (two per SIMD) through one 64-sample tile in lock step: between its barriers both waves of a SIMD
// are in the same phase, so the matrix pipes idle whenever the phase is VALU / LDS work (operand-plane split, mask write, dH1
// store, K = 16 backward): 47.6 k cycles per tile against 30.7 k of matrix-pipe time.  Here the workgroup is two SETS of four waves
// (one per SIMD each), every set owns a 32-sample tile of its own, and a single token orders the sets' MFMA-heavy phases
//      C_a (forward GEMM)  C_b  F_a (dgrad GEMM)  F_b  C_a ...
// so that while one set holds the matrix pipes the other does its serial work on the same SIMDs - the overlap two independent
// workgroups per CU never settle into (profiles/experiments/README.md: they drift).  Sets synchronise among their four waves with
// counters in LDS (s_barrier spans the whole workgroup); a wave covers 32 rows x 64 columns, i.e. the same two accumulator tiles
// per GEMM as before, fed by ONE A fragment (rows) and two B streams (column blocks).
// Outputs, slabs and the ReLU words for the dW2 kernel have the formats of nl_decoder.hip (a set writes its half-words).
#include "nl_common.h"

#define PR_THREADS 512
#define PR_M 32                                         // samples per set tile
#define PR_LDH 257
#define PR_LDX 17
#define PR_STRIDE 264                                   // bf16 elements per plane / mask row (256 + 8 pad)
#define PR_PLANE_BYTES (PR_M * PR_STRIDE * 2)           // 16 896
#define PR_SET_BYTES (3 * PR_PLANE_BYTES)               // 50 688: three H1 planes; the mask tile aliases plane 0, the fp32 dH1 tile planes 1-2
#define PR_W2X_PLANE_BYTES (NL_W * NL_W * 2)
#define PR_WS_W2X_OFF (NL_W * NL_W)                     // floats into the decoder weight workspace (nl_optim.hip)
#define PR_WS_W2TX_OFF (PR_WS_W2X_OFF + 3 * NL_W * NL_W / 2)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define PR_RR(r) (((r) & 3) + 8 * ((r) >> 2))            // row of accumulator register r within a half-wave (+ 4 * lh)

struct PairArgs {
    const NlLossScalars* ls;
    const float* X; const float* params; const float* ws;
    const int* s_ray; const float* s_depth; const float* cos_gt; const float* gt_dist;
    float* sdf; float* dsdf; float* dX;
    float* partials;            // [gridDim.x][NL_DEC_PARAMS] (train)
    unsigned short* relu2_half; // relu2_mask of nl_decoder.hip seen as half-words: [64-sample tile][512 threads][2 sub-tiles]
    double* dcounters;
    long long* dbg;             // optional [16 tiles][16] stamps of set 0 / wave 0 (profiling aid)
};
#define PR_STAMP(slot)                                                                          \
    do { if (a.dbg && blockIdx.x == 0 && tid == 0 && tile_no < 16) a.dbg[tile_no * 16 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ uint4 pr_bload4(rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float pr_trunc(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
__device__ __forceinline__ int pr_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// ---- synchronisation inside the workgroup, all in LDS ------------------------------------------------------------------------
// set barrier: the four waves of a set (a counter per set; `gen` counts this wave's barriers)
__device__ __forceinline__ void pr_set_barrier(int* cnt, int& gen, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    ++gen;
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * gen) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// the MFMA token: wait for the turn `ticket`; after the phase every wave reports, the last of the set's four passes the turn on
__device__ __forceinline__ void pr_wait_turn(int* turn, int ticket)
{
    while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void pr_pass_turn(int* turn, int* arrived, int ticket, int lane)
{
    if (lane == 0) {
        const int n = __hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((n & 3) == 3) __hip_atomic_store(turn, ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(PR_THREADS, 2) void k_decoder_pair(PairArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char sP[2 * PR_SET_BYTES];
    __shared__ __attribute__((aligned(16))) float sXall[2 * 2 * PR_M * PR_LDX];
    __shared__ __attribute__((aligned(16))) float sW1[NL_W * NL_C];
    __shared__ float sSall[2 * 4 * PR_M];
    __shared__ float sdSall[8 * PR_M];
    __shared__ int s_sync[8];                            // [0], [1]: set barrier counters; [2]: turn; [3], [4]: arrivals of the sets

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = w >> 2, wv = w & 3, tis = tid & 255;  // set, wave in set, thread in set
    const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const int col0 = 64 * wv + l31, col1 = col0 + 32;
    unsigned char* sPs = sP + set * PR_SET_BYTES;         // this set's planes / mask tile / dH1 tile
    float* sD = reinterpret_cast<float*>(sPs + PR_PLANE_BYTES);
    float* sXs = sXall + set * (2 * PR_M * PR_LDX);
    float* sS = sSall + set * (4 * PR_M);
    float* sdS = sdSall + w * PR_M;
    int* cnt = s_sync + set; int* turn = s_sync + 2; int* arrived = s_sync + 3 + set;
    int gen = 0;
    if (tid < 8) s_sync[tid] = 0;

    const NlLossScalars ls = *a.ls;
    const int P = ls.P;
    const int ntiles64 = (P + 63) >> 6;
    const float* params = a.params;
    const rsrc_t rsW2X = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ws + PR_WS_W2X_OFF), 0, 3 * PR_W2X_PLANE_BYTES, 0x00020000);
    const rsrc_t rsW2TX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ws + PR_WS_W2TX_OFF), 0, 3 * PR_W2X_PLANE_BYTES, 0x00020000);
    const float b1c0 = params[NL_OFF_B1 + col0], b1c1 = params[NL_OFF_B1 + col1];
    const float b2c0 = params[NL_OFF_B2 + col0], b2c1 = params[NL_OFF_B2 + col1];
    const float w3c0 = params[NL_OFF_W3 + col0], w3c1 = params[NL_OFF_W3 + col1];
    const float b3 = params[NL_OFF_B3];
    const int voff = lane * 16;
    const int kt0 = (2 * wv) * 16 * 1024, kt1 = (2 * wv + 1) * 16 * 1024;      // this wave's two column tiles in the fragment-major planes

    f32x4 accW1[8];
    float aW3[2] = {0.f, 0.f}, aB2[2] = {0.f, 0.f}, aB1[2] = {0.f, 0.f}, aB3 = 0.f;
    double lossFs = 0.0, lossSdf = 0.0;
    if (TRAIN) {
#pragma unroll
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) accW1[t][r] = 0.f;
    }
    // inputs of a set tile: X slice (2 floats per thread of the set) and the loss inputs of row lane & 31 (every wave its own copy)
    const int xe = tis * 2, xi = xe >> 4, xc = xe & 15;
    float2 xv = make_float2(0.f, 0.f);
    float pz = 0.f, pd = 0.f;
    auto prefetch = [&](int t64) {                       // tile32 index = 2 t64 + set
        const int row0 = (2 * t64 + set) * PR_M;
        xv = make_float2(0.f, 0.f); pz = 0.f; pd = 0.f;
        if (t64 < ntiles64) {
            if (row0 + xi < P) xv = *reinterpret_cast<const float2*>(a.X + (size_t)(row0 + xi) * NL_C + xc);
            if (row0 + l31 < P) {
                const int ray = a.s_ray[row0 + l31];
                pz = a.s_depth[row0 + l31] * a.cos_gt[ray]; pd = a.gt_dist[ray];
            }
        }
    };
    prefetch(blockIdx.x);
    for (int i = tid; i < NL_W * NL_C; i += PR_THREADS) sW1[i] = params[NL_OFF_W1 + i];
    { float* sX = sXs; sX[xi * PR_LDX + xc] = xv.x; sX[xi * PR_LDX + xc + 1] = xv.y; }
    float cz = pz, cd = pd;
    prefetch(blockIdx.x + gridDim.x);
    __syncthreads();                                     // sW1, the sync words, both sets' first X tiles

    int tile_no = 0;
    for (int t64 = blockIdx.x; t64 < ntiles64; t64 += gridDim.x, ++tile_no) {
        const int row0 = (2 * t64 + set) * PR_M;
        float* sX = sXs + (tile_no & 1) * (PR_M * PR_LDX);
        PR_STAMP(0);
        // ---------------- B: H1 = relu(X W1^T + b1) for 32 rows x (col0, col1) -> three bf16 planes ----------------
        unsigned m1 = 0u;                                // bit r: H1[row(r)][col0] > 0, bit 16 + r: col1
        {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
            const float* xb = sX + pr_opaque(l31 * PR_LDX + lh);
            const float* wb0 = sW1 + pr_opaque(col0 * NL_C + lh);
            const float* wb1 = wb0 + 32 * NL_C;
#pragma unroll
            for (int kk = 0; kk < NL_C / 2; ++kk) {
                const float xa = xb[2 * kk];
                c0 = MFMA32(xa, wb0[2 * kk], c0); c1 = MFMA32(xa, wb1[2 * kk], c1);
            }
            unsigned short* pb = reinterpret_cast<unsigned short*>(sPs) + pr_opaque(4 * lh * PR_STRIDE + col0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float h = fmaxf((cb ? c1[r] + b1c1 : c0[r] + b1c0), 0.f);
                    m1 |= (h > 0.f) ? (1u << (16 * cb + r)) : 0u;
                    const float hi = pr_trunc(h), r1 = h - hi, mid = pr_trunc(r1), lo = r1 - mid;
                    unsigned short* d = pb + PR_RR(r) * PR_STRIDE + 32 * cb;
                    d[0] = (unsigned short)(__float_as_uint(hi) >> 16);
                    d[PR_PLANE_BYTES / 2] = (unsigned short)(__float_as_uint(mid) >> 16);
                    d[PR_PLANE_BYTES] = (unsigned short)(__float_as_uint(lo) >> 16);
                }
            }
        }
        pr_set_barrier(cnt, gen, lane);
        PR_STAMP(1);
        // ---------------- C: H2 = relu(H1 W2^T + b2), partial row sums of s = H2 w3 (the set's turn on the matrix pipes) ----------------
        f32x16 h0, h1;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
            // B fragments two k-steps ahead (ring of 3 x 2 column tiles x 3 planes), the first two stages before the turn
            uint4 bq[3][2][3];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    bq[s][0][p] = pr_bload4(rsW2TX, voff, p * PR_W2X_PLANE_BYTES + kt0 + s * 1024);
                    bq[s][1][p] = pr_bload4(rsW2TX, voff, p * PR_W2X_PLANE_BYTES + kt1 + s * 1024);
                }
            pr_wait_turn(turn, 4 * tile_no + set);
            PR_STAMP(2);
            const unsigned char* a0 = sPs + pr_opaque(l31 * (PR_STRIDE * 2) + 16 * lh);
            uint4 aq[2];
            aq[0] = *reinterpret_cast<const uint4*>(a0 + 2 * PR_PLANE_BYTES);
#pragma unroll
            for (int g = 0; g < 48; ++g) {               // group = (k-step s = g / 3, A plane pa = 2 - g % 3) x (3 B planes x 2 column tiles)
                const int s = g / 3;
                if (g % 3 == 0 && s + 2 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        bq[(s + 2) % 3][0][p] = pr_bload4(rsW2TX, voff, p * PR_W2X_PLANE_BYTES + kt0 + (s + 2) * 1024);
                        bq[(s + 2) % 3][1][p] = pr_bload4(rsW2TX, voff, p * PR_W2X_PLANE_BYTES + kt1 + (s + 2) * 1024);
                    }
                }
                if (g + 1 < 48) {
                    const int sn = (g + 1) / 3, pn = 2 - (g + 1) % 3;
                    aq[(g + 1) & 1] = *reinterpret_cast<const uint4*>(a0 + pn * PR_PLANE_BYTES + 32 * sn);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 fa = __builtin_bit_cast(bf16x8, aq[g & 1]);
#pragma unroll
                for (int pb = 2; pb >= 0; --pb) {
                    h0 = MFMA_BF16(fa, __builtin_bit_cast(bf16x8, bq[s % 3][0][pb]), h0);
                    h1 = MFMA_BF16(fa, __builtin_bit_cast(bf16x8, bq[s % 3][1][pb]), h1);
                }
            }
            pr_pass_turn(turn, arrived, 4 * tile_no + set, lane);
            PR_STAMP(3);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h0[r] = fmaxf(h0[r] + b2c0, 0.f); h1[r] = fmaxf(h1[r] + b2c1, 0.f);
                v[r] = h0[r] * w3c0 + h1[r] * w3c1;
            }
            // sum over the 32 lanes of each half-wave by recursive halving: lane l31 ends with the total of entry l31 & 15
            float v8[8], v4[4], v2[2];
            { const bool up = (l31 & 8) != 0;
#pragma unroll
              for (int i = 0; i < 8; ++i) v8[i] = (up ? v[i + 8] : v[i]) + __shfl_xor(up ? v[i] : v[i + 8], 8); }
            { const bool up = (l31 & 4) != 0;
#pragma unroll
              for (int i = 0; i < 4; ++i) v4[i] = (up ? v8[i + 4] : v8[i]) + __shfl_xor(up ? v8[i] : v8[i + 4], 4); }
            { const bool up = (l31 & 2) != 0;
#pragma unroll
              for (int i = 0; i < 2; ++i) v2[i] = (up ? v4[i + 2] : v4[i]) + __shfl_xor(up ? v4[i] : v4[i + 2], 2); }
            const bool up1 = (l31 & 1) != 0;
            float tot = (up1 ? v2[1] : v2[0]) + __shfl_xor(up1 ? v2[0] : v2[1], 1);
            tot += __shfl_xor(tot, 16);
            if (l31 < 16) sS[wv * PR_M + PR_RR(l31) + 4 * lh] = tot;
        }
        pr_set_barrier(cnt, gen, lane);
        PR_STAMP(4);
        // ---------------- D: sdf, loss gradient: every wave all 32 rows for itself (lane & 31 = row); wave 0 of the set owns the outputs ----------------
        {
            const int g = row0 + l31;
            float ds = 0.f;
            if (g < P) {
                float s = ((sS[l31] + sS[PR_M + l31]) + sS[2 * PR_M + l31]) + sS[3 * PR_M + l31];      // fixed order: reproducible
                s += b3;
                bool f, m;
                nl_loss_masks(cz, cd, ls.tau, ls.max_depth, &f, &m);
                float q1, q2;
                ds = nl_loss_grad(s, cz, cd, f, m, ls, &q1, &q2);
                if (wv == 0 && lh == 0) {
                    a.sdf[g] = s; a.dsdf[g] = ds;
                    lossFs += (double)q1; lossSdf += (double)q2;
                }
            }
            if (lh == 0) sdS[l31] = ds;
            if (TRAIN && wv == 0 && lh == 0) aB3 += ds;
            __builtin_amdgcn_wave_barrier();
        }
        // ---------------- E: the 0/1 mask tile (dgrad A operand), ReLU half-words for the dW2 kernel, dW3 / db2 partial sums ----------------
        {
            unsigned mw0 = 0u, mw1 = 0u;
            const float* dsb = sdS + pr_opaque(4 * lh);
            unsigned short* mb = reinterpret_cast<unsigned short*>(sPs) + pr_opaque(4 * lh * PR_STRIDE + col0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dr = dsb[PR_RR(r)];
                const bool on0 = h0[r] > 0.f, on1 = h1[r] > 0.f;
                mb[PR_RR(r) * PR_STRIDE] = on0 ? 0x3F80 : 0; mb[PR_RR(r) * PR_STRIDE + 32] = on1 ? 0x3F80 : 0;
                if (TRAIN) {
                    aW3[0] = fmaf(dr, h0[r], aW3[0]); aW3[1] = fmaf(dr, h1[r], aW3[1]);
                    aB2[0] += on0 ? dr * w3c0 : 0.f; aB2[1] += on1 ? dr * w3c1 : 0.f;
                    mw0 |= on0 ? (1u << r) : 0u; mw1 |= on1 ? (1u << r) : 0u;
                }
            }
            if (TRAIN) {                                 // nl_decoder.hip's word (64-sample tile, thread 64 (col / 32) + lane): this sub-tile's half
                unsigned short* mh = a.relu2_half + ((size_t)t64 * 512 + 64 * (2 * wv) + lane) * 2 + set;
                mh[0] = (unsigned short)mw0; mh[128] = (unsigned short)mw1;
            }
        }
        pr_set_barrier(cnt, gen, lane);
        PR_STAMP(5);
        // ---------------- F: dH1 = (mask (w3 W2)) * dsdf * [H1 > 0] (the set's second turn) ----------------
        f32x16 g0v, g1v;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { g0v[r] = 0.f; g1v[r] = 0.f; }
            uint4 bq[3][2][3];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    bq[s][0][p] = pr_bload4(rsW2X, voff, p * PR_W2X_PLANE_BYTES + kt0 + s * 1024);
                    bq[s][1][p] = pr_bload4(rsW2X, voff, p * PR_W2X_PLANE_BYTES + kt1 + s * 1024);
                }
            pr_wait_turn(turn, 4 * tile_no + 2 + set);
            PR_STAMP(6);
            const unsigned char* a0 = sPs + pr_opaque(l31 * (PR_STRIDE * 2) + 16 * lh);
            uint4 aq[2];
            aq[0] = *reinterpret_cast<const uint4*>(a0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 2 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        bq[(s + 2) % 3][0][p] = pr_bload4(rsW2X, voff, p * PR_W2X_PLANE_BYTES + kt0 + (s + 2) * 1024);
                        bq[(s + 2) % 3][1][p] = pr_bload4(rsW2X, voff, p * PR_W2X_PLANE_BYTES + kt1 + (s + 2) * 1024);
                    }
                }
                if (s + 1 < 16) aq[(s + 1) & 1] = *reinterpret_cast<const uint4*>(a0 + 32 * (s + 1));
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 fa = __builtin_bit_cast(bf16x8, aq[s & 1]);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    g0v = MFMA_BF16(fa, __builtin_bit_cast(bf16x8, bq[s % 3][0][p]), g0v);
                    g1v = MFMA_BF16(fa, __builtin_bit_cast(bf16x8, bq[s % 3][1][p]), g1v);
                }
            }
            pr_pass_turn(turn, arrived, 4 * tile_no + 2 + set, lane);
            PR_STAMP(7);
            const float* dsb = sdS + pr_opaque(4 * lh);
            float* db = sD + pr_opaque(4 * lh * PR_LDH + col0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dr = dsb[PR_RR(r)];
                g0v[r] *= ((m1 >> r) & 1u) ? dr : 0.f;
                g1v[r] *= ((m1 >> (16 + r)) & 1u) ? dr : 0.f;
                if (TRAIN) { aB1[0] += g0v[r]; aB1[1] += g1v[r]; }
                // ---------------- H: dH1 -> LDS (planes 1-2 of the set: the forward GEMM is through with them) ----------------
                db[PR_RR(r) * PR_LDH] = g0v[r]; db[PR_RR(r) * PR_LDH + 32] = g1v[r];
            }
        }
        pr_set_barrier(cnt, gen, lane);
        PR_STAMP(8);
        // ---------------- I: waves 0, 1 of the set: dX of 16 rows each; waves 2, 3: dW1 += dH1^T X for 128 units each ----------------
        if (wv < 2) {
            f32x4 cxa = {0.f, 0.f, 0.f, 0.f}, cxb = {0.f, 0.f, 0.f, 0.f};
            const float* ap = sD + pr_opaque((16 * wv + l15) * PR_LDH + lq);
            const float* bqw = sW1 + pr_opaque(lq * NL_C + l15);
            float aA[8], bA[8], aB[8], bB[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * i]; bA[i] = bqw[4 * i * NL_C]; }
#pragma unroll 1
            for (int q = 0; q < NL_W / 4; q += 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { aB[i] = ap[4 * (q + 8 + i)]; bB[i] = bqw[4 * (q + 8 + i) * NL_C]; }
#pragma unroll
                for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aA[i], bA[i], cxa); cxb = MFMA16(aA[i + 1], bA[i + 1], cxb); }
                if (q + 16 < NL_W / 4) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * (q + 16 + i)]; bA[i] = bqw[4 * (q + 16 + i) * NL_C]; }
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aB[i], bB[i], cxa); cxb = MFMA16(aB[i + 1], bB[i + 1], cxb); }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = row0 + 16 * wv + 4 * lq + r;
                if (g < P) a.dX[(size_t)g * NL_C + l15] = cxa[r] + cxb[r];
            }
        } else if (TRAIN) {
            const float* xr = sX + pr_opaque(lq * PR_LDX + l15);
            const float* dr = sD + pr_opaque(lq * PR_LDH + 128 * (wv - 2) + l15);
#pragma unroll
            for (int ii = 0; ii < PR_M / 4; ++ii) {
                const float xb = xr[4 * ii * PR_LDX];
#pragma unroll
                for (int t = 0; t < 8; ++t) accW1[t] = MFMA16(dr[4 * ii * PR_LDH + 16 * t], xb, accW1[t]);
            }
        }
        // ---------------- A (next tile): X -> the other buffer of the set; loads of the tile after it ----------------
        {
            float* sXn = sXs + ((tile_no + 1) & 1) * (PR_M * PR_LDX);
            sXn[xi * PR_LDX + xc] = xv.x; sXn[xi * PR_LDX + xc + 1] = xv.y;
            cz = pz; cd = pd;
            prefetch(t64 + 2 * gridDim.x);
        }
        pr_set_barrier(cnt, gen, lane);
        PR_STAMP(9);
    }

    // ---------------- loss sums ----------------
    if (wv == 0) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { lossFs += __shfl_xor(lossFs, off); lossSdf += __shfl_xor(lossSdf, off); }
        if (lane == 0 && (lossFs != 0.0 || lossSdf != 0.0)) { atomicAdd(&a.dcounters[NLD_FS_SQ], lossFs); atomicAdd(&a.dcounters[NLD_SDF_SQ], lossSdf); }
    }
    if (!TRAIN) return;
    // ---------------- weight-gradient slab of the workgroup: set 1 hands its accumulators over through LDS, set 0 adds and writes ----------------
    aW3[0] += __shfl_xor(aW3[0], 32); aW3[1] += __shfl_xor(aW3[1], 32);
    aB2[0] += __shfl_xor(aB2[0], 32); aB2[1] += __shfl_xor(aB2[1], 32);
    aB1[0] += __shfl_xor(aB1[0], 32); aB1[1] += __shfl_xor(aB1[1], 32);
    if (wv == 0) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) aB3 += __shfl_xor(aB3, off);
    }
    __syncthreads();                                     // both sets are through with their tiles: the plane region is free
    float* sx = reinterpret_cast<float*>(sP);            // [0, 768): W3 | b2 | b1 by column; [768, 4864): dW1; [4864]: db3
    if (set == 1) {
        if (lh == 0) {
            sx[col0] = aW3[0]; sx[col1] = aW3[1]; sx[256 + col0] = aB2[0]; sx[256 + col1] = aB2[1]; sx[512 + col0] = aB1[0]; sx[512 + col1] = aB1[1];
        }
        if (wv >= 2) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sx[768 + (128 * (wv - 2) + 16 * t + 4 * lq + r) * NL_C + l15] = accW1[t][r];
        }
        if (wv == 0 && lane == 0) sx[4864] = aB3;
    }
    __syncthreads();
    if (set == 0) {
        float* base = a.partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
        if (lh == 0) {
            base[NL_OFF_W3 + col0] = aW3[0] + sx[col0]; base[NL_OFF_W3 + col1] = aW3[1] + sx[col1];
            base[NL_OFF_B2 + col0] = aB2[0] + sx[256 + col0]; base[NL_OFF_B2 + col1] = aB2[1] + sx[256 + col1];
            base[NL_OFF_B1 + col0] = aB1[0] + sx[512 + col0]; base[NL_OFF_B1 + col1] = aB1[1] + sx[512 + col1];
        }
        if (wv >= 2) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = (128 * (wv - 2) + 16 * t + 4 * lq + r) * NL_C + l15;
                    base[NL_OFF_W1 + i] = accW1[t][r] + sx[768 + i];
                }
        }
        if (wv == 0 && lane == 0) base[NL_OFF_B3] = aB3 + sx[4864];
    }
}

extern "C" {

int nl_decoder_pair_fwd_bwd(const void* loss_scalars, const float* X, const float* params, const float* ws, const int* s_ray,
                            const float* s_depth, const float* cos_gt, const float* gt_dist, float* sdf, float* dsdf, float* dX,
                            float* partials, unsigned* relu2_mask, int nslabs, int train_decoder, int* counters, void* dbg, void* stream)
{
    PairArgs a;
    a.ls = (const NlLossScalars*)loss_scalars; a.X = X; a.params = params; a.ws = ws; a.s_ray = s_ray; a.s_depth = s_depth;
    a.cos_gt = cos_gt; a.gt_dist = gt_dist; a.sdf = sdf; a.dsdf = dsdf; a.dX = dX; a.partials = partials;
    a.relu2_half = reinterpret_cast<unsigned short*>(relu2_mask);
    a.dcounters = (double*)(counters + NL_CNT_INTS);
    a.dbg = (long long*)dbg;
    const dim3 g(nslabs), b(PR_THREADS);
    if (train_decoder) hipLaunchKernelGGL(k_decoder_pair<true>, g, b, 0, (hipStream_t)stream, a);
    else               hipLaunchKernelGGL(k_decoder_pair<false>, g, b, 0, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

}  // extern "C"
