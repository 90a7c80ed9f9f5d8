// nl_decoder.hip -- the SDF decoder (16 -> 256 -> 256 -> 1 ReLU MLP) forward + loss gradient + backward as ONE persistent
// kernel on the matrix cores of gfx950, plus the dW2 kernel and the forward-only kernel.
//
// Reference behaviour: src/variations/lidar.py:109-131 (forward), src/criterion.py:59-100 (loss), torch autograd for the
// backward.  All values and all accumulations are fp32.  The three 256 x 256 contractions run
// on the bf16 matrix cores as EXACT-PRODUCT formulations (gemm_x9, gemm_mask_x, k_decoder_wgrad2_x): an fp32
// operand is the exact sum of three bf16 terms, a {0,1} ReLU mask is a bf16 operand as it stands, every product the matrix
// core forms is exact, accumulation is fp32.  The plain fp32-MFMA versions (v_mfma_f32_32x32x2_f32: gemm256,
// k_decoder_wgrad2) stay selectable (nl_decoder_set_gemm_mode / nl_decoder_set_wgrad2_mode) and are cross-checked in
// tests/test_gpu_parity.py; the K = 16 layers always run on the fp32 matrix cores.
//
// Structure (MI355X-first, not how the reference does it - it runs 3 GEMMs + autograd):
//   * one 512-thread workgroup (8 waves, 2 per SIMD) per CU, persistent over 64-sample tiles, 5 barriers per tile;
//   * per tile everything stays on chip: X tile, H1 (three bf16 planes / fp32 tile), the ReLU-mask tile and dH1 live in
//     LDS; H2 never leaves registers;
//   * the loss gradient dL/dsdf needs only per-sample geometry + iteration-global scalars that are known before the decoder
//     runs (nl_geometry.hip), so forward, loss and backward fuse; every wave recomputes the 64 loss gradients of the tile
//     for itself (no barrier between the loss and the mask phase);
//   * W2 (256 KB fp32 > LDS) is streamed from L2 straight into MFMA B operands, pre-split into bf16 planes in MFMA-fragment
//     order by k_prepare_w2x (nl_optim.hip): one contiguous 1 KB buffer_load_dwordx4 per fragment, software-pipelined;
//   * weight gradients accumulate in registers across ALL tiles of the workgroup and are flushed once as a per-workgroup
//     partial slab - no atomics; nl_reduce_partials sums the slabs.  The big one, dW2 = dH2^T H1 (8 tiles of 32x32 per wave
//     = 128 accumulator registers), runs in its own persistent kernel so that neither kernel spills: it rebuilds H1 from X
//     (K = 16) and dH2 from dsdf + the 256-bit ReLU mask the first kernel saves per sample (32 B instead of a 1 KB row).
// FLOPs per sample (algorithmic): 3 * 2 * (16*256 + 256*256 + 256) = 419,328 (279,552 with a frozen decoder).
#include "nl_common.h"
#include <atomic>

#define DEC_M 64
#define DEC_THREADS 512
#define LDH 257
#define LDX 17

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA16_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// LDS carve (floats)
#define S_H1 0
#define S_D (S_H1 + DEC_M * LDH)
#define S_X (S_D + DEC_M * LDH)                         // two X tiles (double-buffered across tiles)
#define S_W1 (S_X + 2 * DEC_M * LDX)
#define S_S (S_W1 + NL_W * NL_C)
#define S_DS (S_S + 8 * DEC_M)                          // sS: per-wave partial row sums [8][64] (fixed-order reduce)
#define S_TOTAL (S_DS + 8 * DEC_M)                      // sdS: one copy of the tile's dL/dsdf per wave

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
};

// (STAMPS: a template argument of k_decoder - the production instantiations carry no stamp code: eleven branches and their scalar registers per tile)
#define DBG_STAMP(slot)                                                                          \
    do { if constexpr (STAMPS) { if (a.dbg && blockIdx.x == 0 && tid == 0 && tile_no < 16) a.dbg[tile_no * 16 + (slot)] = (long long)__builtin_readcyclecounter(); } } while (0)

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

// upper halves of two fp32 registers -> one register of two bf16 (lo = a, hi = b): truncation, used on values whose
// low 16 bits are already zero or are carried by the next split term
__device__ __forceinline__ unsigned pack_hi16(float a, float b)
{
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
// two fp32 values as three packed bf16 pairs: a = hi + mid + lo exactly (nl_split3_bf16's terms, by truncation)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned* hi, unsigned* mid, unsigned* lo)
{
    const float r0 = a - trunc_bf16(a), r1 = b - trunc_bf16(b);
    *hi = pack_hi16(a, b); *mid = pack_hi16(r0, r1); *lo = pack_hi16(r0 - trunc_bf16(r0), r1 - trunc_bf16(r1));
}
// layer 1 on the bf16 matrix cores: this lane's B fragments, W1[col][8 lh + e] (e = 0..7) as three bf16 terms
__device__ __forceinline__ void l1_w1_fragments(const float* __restrict__ params, int col, int lh, uint4 (&out)[3])
{
    const float4* wr = reinterpret_cast<const float4*>(params + NL_OFF_W1 + col * NL_C + 8 * lh);
    const float4 wa = wr[0], wb = wr[1];
    const float v[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    unsigned q[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3_pair(v[2 * e], v[2 * e + 1], &q[0][e], &q[1][e], &q[2][e]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[pl] = make_uint4(q[pl][0], q[pl][1], q[pl][2], q[pl][3]);
}

// ---- "f16 pair" arithmetic (gemm modes 4 / 5; nl_device_math.h nl_split2_f16): an fp32 operand, scaled by a power of two and saturated at
// the fp16 range by the caller, as hi = f16(x), lo = f16(x - hi), both round-to-nearest-even (v_cvt_pk_f16_f32); two values per call,
// packed like the MFMA fragments want them.  4 VALU instructions per pair against ~13 for the three-term bf16 split.
__device__ __forceinline__ void split2_pair_f16(float a, float b, unsigned* hi, unsigned* lo)
{
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const f32x2v v = {a, b};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    // residuals a - (float)h.lo16, b - (float)h.hi16: v_fma_mix_f32 reads the fp16 half as an operand (fma(h, -1, a): the exact difference, one
    // instruction instead of a conversion and a subtraction; the compiler does not select it by itself)
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(b));
    const f32x2v r = {r0, r1};
    *hi = h; *lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ float sat_f16(float x) { return __builtin_amdgcn_fmed3f(x, -NL_F16_MAX, NL_F16_MAX); }
// layer 1 as f16 pairs: this lane's B fragments, W1[col][8 lh + e] * 2^8 (e = 0..7)
__device__ __forceinline__ void l1_w1_fragments_f16(const float* __restrict__ params, int col, int lh, uint4 (&out)[2])
{
    const float4* wr = reinterpret_cast<const float4*>(params + NL_OFF_W1 + col * NL_C + 8 * lh);
    const float4 wa = wr[0], wb = wr[1];
    const float v[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    unsigned q[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_pair_f16(sat_f16(v[2 * e] * NL_F16_SW1), sat_f16(v[2 * e + 1] * NL_F16_SW1), &q[0][e], &q[1][e]);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) out[pl] = make_uint4(q[pl][0], q[pl][1], q[pl][2], q[pl][3]);
}

// LICM hoists every "base + constant" LDS address out of the persistent tile loop into its own VGPR (dozens of them);
// laundering the base through an empty asm inside the loop keeps ONE base register and lets the constants fold into the
// ds_read/ds_write offset fields.
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float launder_f(float x) { asm volatile("" : "+v"(x)); return x; }
#define D32_RR(r) (((r) & 3) + 8 * ((r) >> 2))          // row of accumulator register r within a half-wave (+ 4*lh)

// ---- dgrad on the bf16 matrix cores, exactly -------------------------------------------------------------------------
// dH1[i][k] = sum_j dH2[i][j] W2[j][k] with dH2[i][j] = m(i,j) dsdf_i w3_j (m = 0/1 ReLU mask of H2), hence
//     dH1[i][k] = dsdf_i * sum_j m(i,j) * (w3_j W2[j][k]).
// A = m is exact in bf16; B = w3_j W2[j][k] (one fp32 rounding, done once per optimiser step by k_prepare_w2x) is split
// into three bf16 terms whose sum is the fp32 value exactly.  Every product in the matrix core is exact, accumulation is
// fp32: the same arithmetic class as the fp32 MFMA GEMM at 3/16 of its pipe time.
// W2X layout (fragment-major, so every B fragment is one contiguous 1 KB buffer_load_dwordx4 per wave):
//     [plane p(3)][column tile kt(8)][k-step s(16)][lane(64)][8 bf16],  lane = 32*h + n holds j = 16 s + 8 h + e (e = 0..7)
//     of column k = 32 kt + n.
// The mask tile lives in LDS as bf16 [64 rows][264] (row stride 528 B: conflict-free ds_read_b128 A fragments).
#define SM_STRIDE 264                                   // bf16 elements per mask row (256 + 8 pad)
#define W2X_PLANE_BYTES (NL_W * NL_W * 2)
#define NL_DEC_WS_W2X_OFF (NL_W * NL_W)                 // floats: W2X follows W2T in the decoder weight workspace

__device__ __forceinline__ uint4 bload4(i32x4 rsrc, int voff_bytes, int soff_bytes)
{
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, 0));
}

// one 32x32x16 matrix-core instruction on packed 16-bit fragments: bf16 (exact three-term splits) or fp16 (two-term pairs)
template <bool F16>
__device__ __forceinline__ f32x16 mma16(const uint4& a, const uint4& b, const f32x16& c)
{
    if constexpr (F16) return MFMA_F16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c);
    else return MFMA_BF16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
}

#define MX_RING 6                                       // B fragments run MX_RING-1 k-steps (6 MFMAs = 192 pipe cycles each) ahead
#define MX_PRE 2                                        // of which this many are issued before the barrier (register budget)
// first MX_PRE stages of the B stream: independent of the tile, so issued BEFORE the barrier that publishes the mask
template <int NPL>                                      // NPL: operand planes of the B matrix (3 bf16 terms, or 2 with the fp16 pairs)
__device__ __forceinline__ void gemm_mask_x_prefetch(i32x4 rsX, int w, int lane, uint4 (&bq)[MX_RING][3])
{
    const int voff = lane * 16;
    const int kt_off = __builtin_amdgcn_readfirstlane(w) * 16 * 1024;
#pragma unroll
    for (int s = 0; s < MX_PRE; ++s)
#pragma unroll
        for (int p = 0; p < NPL; ++p) bq[s][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + kt_off + s * 1024);
}

template <bool PIN, bool F16 = false, int RING = MX_RING>   // RING <= MX_RING: B fragments RING - 1 k-steps ahead.  PIN: hold the software pipeline in place with scheduling barriers.  Without them
                                                        // the scheduler sinks every prefetch to just before its use (vmcnt(0) after each
                                                        // load); with them the trainable-decoder kernel, which is at the 256-VGPR limit,
                                                        // spills ~40 registers - so it is enabled where registers allow (frozen / forward)
__device__ __forceinline__ void gemm_mask_x(i32x4 rsX, int w, int lane, const unsigned char* sM, uint4 (&bq)[MX_RING][3], f32x16& c0, f32x16& c1)
{
    const int l31 = lane & 31, lh = lane >> 5;
    const int voff = lane * 16;
    const int kt_off = __builtin_amdgcn_readfirstlane(w) * 16 * 1024;     // this wave's column tile; provably wave-uniform, or every
                                                                          // load below becomes a waterfall loop over soffset
    const unsigned char* a0 = sM + opaque(l31 * (SM_STRIDE * 2) + 16 * lh);
    static_assert(RING <= MX_RING && RING - 1 >= MX_PRE, "ring depth");
    constexpr int NPL = F16 ? 2 : 3;
    uint4 aq[2][2];
#pragma unroll
    for (int s = MX_PRE; s < RING - 1; ++s)
#pragma unroll
        for (int p = 0; p < NPL; ++p) bq[s][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + kt_off + s * 1024);
    aq[0][0] = *reinterpret_cast<const uint4*>(a0); aq[0][1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s + RING - 1 < 16) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) bq[(s + RING - 1) % RING][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + kt_off + (s + RING - 1) * 1024);
        }
        if (s + 1 < 16) {
            aq[(s + 1) & 1][0] = *reinterpret_cast<const uint4*>(a0 + 32 * (s + 1));
            aq[(s + 1) & 1][1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2 + 32 * (s + 1));
        }
        if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int p = F16 ? NPL - 1 - i : i;           // (fp16 pairs: the low term first)
            c0 = mma16<F16>(aq[s & 1][0], bq[s % RING][p], c0); c1 = mma16<F16>(aq[s & 1][1], bq[s % RING][p], c1);
        }
    }
}

// Rolled variant for the trainable-decoder kernel (no scheduling barriers, so no spills at its 256-VGPR limit): the k-steps go
// through a rolled loop in pairs with two register buffers, like gemm256 - a load issued in one half of the body cannot be
// sunk below the MFMAs of that half because its consumers sit in the other half.  bq[0], bq[1] arrive preloaded.
template <bool F16 = false, int HS = 2>                 // HS: k-steps per register buffer - the B fragments of a buffer are requested HS k-steps (HS x 256 cycles with the SIMD's
                                                        // other wave) before their first use.  2 at the register limit of the bf16 kernels; 4 in the fp16-pair kernel since its ReLU bits
                                                        // left the scalar file (243 -> 199 registers): an L2 hit takes ~800 cycles, two k-steps did not cover it (phase F 4.9 k cycles
                                                        // against 2.8 k in the frozen kernel's pinned loop)
__device__ __forceinline__ void gemm_mask_x_rolled(i32x4 rsX, int w, int lane, const unsigned char* sM, uint4 (&bq)[MX_RING][3], f32x16& c0, f32x16& c1)
{
    static_assert(MX_PRE == 2, "the rolled loop consumes the two pre-barrier stages first");
    static_assert(HS == 2 || HS == 4, "16 k-steps in two buffers of HS");
    const int l31 = lane & 31, lh = lane >> 5;
    const int voff = lane * 16;
    const int kt_off = __builtin_amdgcn_readfirstlane(w) * 16 * 1024;
    const unsigned char* a0 = sM + opaque(l31 * (SM_STRIDE * 2) + 16 * lh);
    constexpr int NPL = F16 ? 2 : 3;
    uint4 bA[HS][3], bB[HS][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < NPL; ++p) bA[t][p] = bq[t][p];
#pragma unroll
    for (int t = 2; t < HS; ++t)
#pragma unroll
        for (int p = 0; p < NPL; ++p) bA[t][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + kt_off + t * 1024);
    auto steps = [&](const uint4 (&b)[HS][3], const unsigned char* ap) {
#pragma unroll
        for (int t = 0; t < HS; ++t) {
            const uint4 fa0 = *reinterpret_cast<const uint4*>(ap + 32 * t);
            const uint4 fa1 = *reinterpret_cast<const uint4*>(ap + 32 * SM_STRIDE * 2 + 32 * t);
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const int p = F16 ? NPL - 1 - i : i;       // (fp16 pairs: the low term first)
                c0 = mma16<F16>(fa0, b[t][p], c0); c1 = mma16<F16>(fa1, b[t][p], c1);
            }
        }
    };
#pragma unroll 1
    for (int s = 0; s < 16; s += 2 * HS) {
        const int so = kt_off + s * 1024;
#pragma unroll
        for (int t = 0; t < HS; ++t)
#pragma unroll
            for (int p = 0; p < NPL; ++p) bB[t][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + so + (HS + t) * 1024);
        steps(bA, a0 + 32 * s);
        if (s + 2 * HS < 16) {
#pragma unroll
            for (int t = 0; t < HS; ++t)
#pragma unroll
                for (int p = 0; p < NPL; ++p) bA[t][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + so + (2 * HS + t) * 1024);
        }
        steps(bB, a0 + 32 * (s + HS));
    }
}

// ---- forward GEMM on the bf16 matrix cores with exact products ---------------------------------------------------------
// H2pre[i][n] = sum_k H1[i][k] W2[n][k]: both operands are general fp32, so BOTH are split into three bf16 terms
// (x = x_hi + x_mid + x_lo exactly) and all nine partial products are formed: each bf16 x bf16 product is exact in fp32,
// their sum is the exact fp32 x fp32 product, accumulation is fp32 - again the arithmetic class of the fp32 MFMA GEMM,
// at 9/16 of its pipe time.  A planes: LDS, bf16 [3][64 rows][264]; B planes "W2TX": fragment-major like W2X with
// k = 16 s + 8 h + e (input index) and n = 32 nt + (lane & 31) (output column).
#define X_PLANE_ELEMS (DEC_M * SM_STRIDE)
#define X_PLANE_BYTES (X_PLANE_ELEMS * 2)
#define NL_DEC_WS_W2TX_OFF (NL_DEC_WS_W2X_OFF + 3 * NL_W * NL_W / 2)       // floats
#define NL_DEC_WS_W2H_OFF (NL_DEC_WS_W2TX_OFF + 3 * NL_W * NL_W / 2)      // fp16 pairs: dgrad planes (w3_j W2[j][k] * 2^10 = NL_F16_SG), two planes
#define NL_DEC_WS_W2TH_OFF (NL_DEC_WS_W2H_OFF + 2 * NL_W * NL_W / 2)      // forward planes (W2 * 2^8)
static_assert(NL_DEC_WS_W2TH_OFF + 2 * NL_W * NL_W / 2 == NL_DEC_WS_W1F_OFF, "W1F / W1X (nl_common.h) follow the W2 planes");

// bf16 mode, layer 1 on the bf16 matrix cores as well (exact 3 x 3 term products, nine K = 16 MFMAs per 32-row sub-tile instead of
// eight fp32 ones): W1 as B fragments [wave][plane][lane][16 B] and the X tile as three planes [64 rows][32 B], in LDS the bf16 mode
// leaves unused (between the H1 planes and the fp32 X tiles; the last X plane behind the carve)
#define XG_W1X_OFF (3 * X_PLANE_BYTES)
#define XG_W1X_BYTES (8 * 3 * 64 * 16)
#define XG_XP_BYTES (DEC_M * 32)
#define XG_XP01_OFF (XG_W1X_OFF + XG_W1X_BYTES)
#define XG_XP2_OFF (S_TOTAL * 4)
#define S_ALLOC (S_TOTAL + XG_XP_BYTES / 4)
static_assert(XG_XP01_OFF + 2 * XG_XP_BYTES <= S_X * 4, "layer-1 operands overlap the fp32 X tiles");
static_assert(S_ALLOC * 4 <= 163840, "LDS");

#define X9_RING 3                                       // B fragments 2 k-steps (18 MFMAs = 576 pipe cycles each) ahead
__device__ __forceinline__ void gemm_x9_prefetch(i32x4 rsX, int w, int lane, uint4 (&bq)[X9_RING][3])
{
    const int voff = lane * 16;
    const int nt_off = __builtin_amdgcn_readfirstlane(w) * 16 * 1024;
#pragma unroll
    for (int s = 0; s < X9_RING - 1; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[s][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + nt_off + s * 1024);
}

// NP = 9: all nine partial products (exact).  NP = 8 (the default): without lo x lo - |x_lo| < 2^-15 |x| for both factors, so the
// dropped term is below 2^-30 of the product, 2^-6 of one rounding of the fp32 accumulation that follows (bound checked in rational
// arithmetic on the host, tests/test_device_math_host.py) - at 8/9 of the matrix-pipe time.  NP = 6: without lo x lo, lo x mid, mid x lo - terms below 2^-24 of the product,
// i.e. below the rounding of the fp32 accumulation itself (measured on a 4096 x 256 x 256 case: rms error 7e-9 of sum|a||b|
// against 2.6e-8 for an fp32 GEMM's own rounding) - at two thirds of the matrix-pipe time.
template <bool PIN, int NP = 9>
__device__ __forceinline__ void gemm_x9(i32x4 rsX, int w, int lane, const unsigned char* sP, uint4 (&bq)[X9_RING][3], f32x16& c0, f32x16& c1)
{
    static_assert(NP == 9 || NP == 8 || NP == 6, "nine, eight or six partial products");
    const int l31 = lane & 31, lh = lane >> 5;
    const int voff = lane * 16;
    const int nt_off = __builtin_amdgcn_readfirstlane(w) * 16 * 1024;     // wave-uniform scalar offset (no waterfall loops)
    const unsigned char* a0 = sP + opaque(l31 * (SM_STRIDE * 2) + 16 * lh);
    constexpr int RING = X9_RING;
    // 48 groups of 6 MFMAs: group g = (k-step s = g / 3, A plane pa = 2 - g % 3) x (3 B planes x 2 row tiles).  The A fragments
    // of the NEXT group are read from LDS while this group's MFMAs run (192 pipe cycles > LDS latency): 16 registers of A
    // instead of 48 for a whole k-step ahead.
    uint4 aq[2][2];
    aq[0][0] = *reinterpret_cast<const uint4*>(a0 + 2 * X_PLANE_BYTES);
    aq[0][1] = *reinterpret_cast<const uint4*>(a0 + 2 * X_PLANE_BYTES + 32 * SM_STRIDE * 2);
#pragma unroll
    for (int g = 0; g < 48; ++g) {
        const int s = g / 3;
        // one memory instruction after every second MFMA (the group's three memory instructions spread over its six MFMAs) instead
        // of all of them above the group: bunched, they hold up the issue of the next MFMA (scripts/micro/chain_stream.hip)
        const bf16x8 fa0 = __builtin_bit_cast(bf16x8, aq[g & 1][0]), fa1 = __builtin_bit_cast(bf16x8, aq[g & 1][1]);
        const int sn = (g + 1) / 3, pn = 2 - (g + 1) % 3;
#pragma unroll
        for (int pb = 2; pb >= 0; --pb) {
            // A plane (2 - g % 3) x B plane pb (0 = hi, 1 = mid, 2 = lo).  NP = 6 keeps the index sums <= 2, NP = 8 drops lo x lo only
            if (!(NP == 6 && (2 - g % 3) + pb > 2) && !(NP == 8 && (2 - g % 3) + pb > 3)) {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bq[s % RING][pb]);
                c0 = MFMA_BF16(fa0, fb, c0); c1 = MFMA_BF16(fa1, fb, c1);
            }
            if (pb == 2 && g + 1 < 48) aq[(g + 1) & 1][0] = *reinterpret_cast<const uint4*>(a0 + pn * X_PLANE_BYTES + 32 * sn);
            if (pb == 1 && g + 1 < 48) aq[(g + 1) & 1][1] = *reinterpret_cast<const uint4*>(a0 + pn * X_PLANE_BYTES + 32 * SM_STRIDE * 2 + 32 * sn);
            if (pb == 0 && s + RING - 1 < 16)
                bq[(s + RING - 1) % RING][g % 3] = bload4(rsX, voff, (g % 3) * W2X_PLANE_BYTES + nt_off + (s + RING - 1) * 1024);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- forward GEMM as fp16 pairs (gemm modes 4 / 5) ------------------------------------------------------------------------------
// H1 * 2^4 and W2 * 2^8 as hi + lo (nl_split2_f16); per k-step and 32-row sub-tile the products lo x hi, hi x lo, hi x hi (NP = 3: what is
// dropped, lo x lo, is below 2^-22 of a product - under the rounding of the 256-deep fp32 accumulation) or all four (NP = 4): 6 or 8 matrix
// instructions per k-step against 16 of the eight-product bf16 split.  A planes: LDS, f16 [2][64 rows][264]; B planes "W2TH": fragment-major
// like W2TX.  The accumulators come out scaled by 2^12 (the caller's epilogue folds 2^-12 into its bias add: exact).
#define F16_RING 4                                      // B fragments 3 k-steps (6-8 MFMAs = 192-256 pipe cycles each) ahead
__device__ __forceinline__ void gemm_f16_prefetch(i32x4 rsH, int w, int lane, uint4 (&bq)[F16_RING][2])
{
    const int voff = lane * 16;
    const int nt_off = __builtin_amdgcn_readfirstlane(w) * 16 * 1024;
#pragma unroll
    for (int s = 0; s < F16_RING - 1; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) bq[s][p] = bload4(rsH, voff, p * W2X_PLANE_BYTES + nt_off + s * 1024);
}

template <bool PIN, int NP>
__device__ __forceinline__ void gemm_f16(i32x4 rsH, int w, int lane, const unsigned char* sP, uint4 (&bq)[F16_RING][2], f32x16& c0, f32x16& c1)
{
    static_assert(NP == 3 || NP == 4, "three or four partial products");
    const int l31 = lane & 31, lh = lane >> 5;
    const int voff = lane * 16;
    const int nt_off = __builtin_amdgcn_readfirstlane(w) * 16 * 1024;     // wave-uniform scalar offset (no waterfall loops)
    const unsigned char* a0 = sP + opaque(l31 * (SM_STRIDE * 2) + 16 * lh);
    constexpr int RING = F16_RING;
    // the A fragments of the NEXT k-step (hi / lo plane x two row tiles: four ds_read_b128) and the B fragments three k-steps ahead are
    // issued between this k-step's matrix instructions, one memory instruction per instruction pair
    uint4 aq[2][4];                                      // [k-step parity][hi rows 0-31, hi rows 32-63, lo rows 0-31, lo rows 32-63]
    aq[0][0] = *reinterpret_cast<const uint4*>(a0); aq[0][1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2);
    aq[0][2] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES); aq[0][3] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES + 32 * SM_STRIDE * 2);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        const uint4 bh = bq[s % RING][0], bl = bq[s % RING][1];
        if (NP == 4) { c0 = mma16<true>(aq[cur][2], bl, c0); c1 = mma16<true>(aq[cur][3], bl, c1); }
        c0 = mma16<true>(aq[cur][2], bh, c0); c1 = mma16<true>(aq[cur][3], bh, c1);
        if (s + 1 < 16) {
            aq[nxt][0] = *reinterpret_cast<const uint4*>(a0 + 32 * (s + 1));
            aq[nxt][1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2 + 32 * (s + 1));
        }
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        c0 = mma16<true>(aq[cur][0], bl, c0); c1 = mma16<true>(aq[cur][1], bl, c1);
        if (s + 1 < 16) {
            aq[nxt][2] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES + 32 * (s + 1));
            aq[nxt][3] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES + 32 * SM_STRIDE * 2 + 32 * (s + 1));
        }
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        c0 = mma16<true>(aq[cur][0], bh, c0); c1 = mma16<true>(aq[cur][1], bh, c1);
        if (s + RING - 1 < 16) {
#pragma unroll
            for (int p = 0; p < 2; ++p) bq[(s + RING - 1) % RING][p] = bload4(rsH, voff, p * W2X_PLANE_BYTES + nt_off + (s + RING - 1) * 1024);
        }
        if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
}

// value of the neighbouring lane (lane ^ 1) - the neighbouring output column of the 32x32 MFMA tile
__device__ __forceinline__ float dpp_swap1(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ unsigned dpp_swap1u(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }

// [v > 0] as the integer 0 / 1 in ONE instruction: the bit pattern clamped to [0, 1] as a signed integer (positive floats are positive
// integers, +0 is 0, everything with the sign bit is negative).  Written as an instruction because the compiler turns min(max(x, 0), 1) back into
// compare + select: a v_cmp puts the result in a scalar register PAIR, and the thirty-two ReLU bits of a lane became 64 live scalar registers
// (53 of them spilled to vector lanes: ~130 v_readlane / v_writelane per wave and tile) or thirty-two vector registers kept alive for a deferred
// compare (round 5: the H1 values from phase B to phase F).
__device__ __forceinline__ unsigned pos_bit(float v)
{
    unsigned b;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(b) : "v"(v));
    return b;
}

// maximum of a non-negative value over the wave: four DPP steps inside the 16-lane rows, then the four rows through scalar registers
// (no LDS round trips: __shfl_xor is a ds_bpermute, ~100 cycles each on a latency chain)
__device__ __forceinline__ float wave_max_nonneg(float v)
{
    int x = __float_as_int(v);                           // non-negative floats order like their bit patterns
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true));        // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true));       // row_half_mirror
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true));       // row_mirror
    const int m = max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
    return __int_as_float(m);
}

// Range watch of the fp16-pair arithmetic on a workgroup's FIRST tile (nl_common.h NL_SAT_H1 / NL_SAT_Q): ~100 instructions once per workgroup instead of
// one or two per element in every tile.  h1_clips: H1 * 2^4 = relu(acc * S1 + b1s) reaches 65504 for a layer-1 accumulator of this lane; q_clips: a dgrad
// accumulator of a row whose H1 is positive (the others are discarded) reaches +-65504.  Negated comparisons: a NaN counts.
__device__ __forceinline__ bool h1_clips(const f32x16& c0, const f32x16& c1, float b1s)
{
    constexpr float S1 = NL_F16_SH / (NL_F16_SX * NL_F16_SW1);
    float m = c0[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(fmaxf(c0[r], c1[r]), m);
    bool nan = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) nan |= (c0[r] != c0[r]) | (c1[r] != c1[r]);
    return nan || !(fmaf(m, S1, b1s) < NL_F16_MAX);
}
__device__ __forceinline__ bool q_clips(const f32x16& g0, const f32x16& g1, unsigned m1)
{
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bad |= ((m1 >> r) & 1u) && !(fabsf(g0[r]) < NL_F16_MAX);
        bad |= ((m1 >> (16 + r)) & 1u) && !(fabsf(g1[r]) < NL_F16_MAX);
    }
    return bad;
}

// H1 = relu(pre + b1) for this lane's column / 32 rows -> three bf16 planes in LDS (A operand of gemm_x9); returns the
// lane's 32 ReLU bits (bit r: row d32_row(r, lh), bit 16 + r: row 32 + d32_row(r, lh)) for the dgrad epilogue.
// Two neighbouring lanes hold neighbouring columns (k, k + 1) of the same rows, and a row of a plane is k-contiguous: the even lane
// takes the odd lane's value of row r, the odd lane the even lane's value of row r + 1 (one DPP move per value pair), so every lane
// stores (k, k + 1) of ONE row as a dword - half the LDS store instructions, and none of the two-lanes-per-bank-word conflicts of
// 16-bit stores.  The same bits land in the same places.
__device__ __forceinline__ unsigned store_h1_planes(unsigned short* sP, int col, int lh, const f32x16& c0, const f32x16& c1, float b1c)
{
    const bool odd = (col & 1) != 0;
    unsigned* pb = reinterpret_cast<unsigned*>(sP + opaque((4 * lh + (odd ? 1 : 0)) * SM_STRIDE + (col & ~1)));   // odd lanes: the odd rows
    // plane 2 lies beyond the 64 KB an LDS offset field reaches: its own base register instead of an address add per store
    unsigned* pb2 = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(sP) + opaque(((4 * lh + (odd ? 1 : 0)) * SM_STRIDE + (col & ~1)) * 2 + 2 * X_PLANE_BYTES));
    unsigned m1 = 0u;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float ha = fmaxf((sub ? c1[r] : c0[r]) + b1c, 0.f), hb = fmaxf((sub ? c1[r + 1] : c0[r + 1]) + b1c, 0.f);
            m1 |= ((ha > 0.f) ? (1u << (16 * sub + r)) : 0u) | ((hb > 0.f) ? (1u << (16 * sub + r + 1)) : 0u);
            const float got = dpp_swap1(odd ? ha : hb);            // even lane: the odd lane's row r; odd lane: the even lane's row r + 1
            const float lo_k = odd ? got : ha, hi_k = odd ? hb : got;      // columns (k, k + 1) of this lane's row
            unsigned q0, q1, q2;
            split3_pair(lo_k, hi_k, &q0, &q1, &q2);
            unsigned* d = pb + ((32 * sub + D32_RR(r)) * SM_STRIDE) / 2;
            d[0] = q0; d[X_PLANE_ELEMS / 2] = q1; pb2[((32 * sub + D32_RR(r)) * SM_STRIDE) / 2] = q2;
        }
    }
    return m1;
}

// the same for the fp16 pairs: H1 * 2^4 = relu(acc * 2^-10 + 16 b1) (acc = the layer-1 accumulator, scaled 2^14; power-of-two scaling commutes
// with the rounding of the add), saturated at the fp16 range by the same v_med3 that is the ReLU, as two planes hi / lo
__device__ __forceinline__ unsigned store_h1_planes_f16(unsigned short* sP, int col, int lh, const f32x16& c0, const f32x16& c1, float b1s)
{
    const bool odd = (col & 1) != 0;
    unsigned* pb = reinterpret_cast<unsigned*>(sP + opaque((4 * lh + (odd ? 1 : 0)) * SM_STRIDE + (col & ~1)));   // odd lanes: the odd rows
    constexpr float S1 = NL_F16_SH / (NL_F16_SX * NL_F16_SW1);
    unsigned m1 = 0u;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float ha = __builtin_amdgcn_fmed3f(fmaf(sub ? c1[r] : c0[r], S1, b1s), 0.f, NL_F16_MAX);
            const float hb = __builtin_amdgcn_fmed3f(fmaf(sub ? c1[r + 1] : c0[r + 1], S1, b1s), 0.f, NL_F16_MAX);
            m1 |= (pos_bit(ha) << (16 * sub + r)) | (pos_bit(hb) << (16 * sub + r + 1));
            const float got = dpp_swap1(odd ? ha : hb);            // even lane: the odd lane's row r; odd lane: the even lane's row r + 1
            const float lo_k = odd ? got : ha, hi_k = odd ? hb : got;      // columns (k, k + 1) of this lane's row
            unsigned q0, q1;
            split2_pair_f16(lo_k, hi_k, &q0, &q1);
            unsigned* d = pb + ((32 * sub + D32_RR(r)) * SM_STRIDE) / 2;
            d[0] = q0; d[X_PLANE_ELEMS / 2] = q1;
        }
    }
    return m1;
}

// For each of the 32 rows a half-wave holds (16 of h0 ++ 16 of h1), the sum over its 32 lanes of
// h[row] * w3c, by recursive halving: 31 exchanges instead of 160; lane l31 ends up with the total of
// list entry e = l31.  Register-lean: level 1 consumes h0/h1 directly (16 live values), then 8, 4, 2, 1.
// No LDS round trips (round 5; the ds_bpermute of __shfl_xor put five LDS latencies on the kernel's critical path): the 16-lane rows of a
// half-wave exchange through v_permlane16_swap (one instruction swaps what the two rows send each other - no selects at that level),
// the levels inside a row through DPP operands of the add itself: row_mirror (l <-> 15 - l: bit 3 flips), row_half_mirror (l <-> 7 - l inside
// each 8: bit 2 flips), quad_perm (xor 2, xor 1).  Any pairing that flips the level's lane bit halves correctly; which lanes a partial sum
// has visited differs from the xor butterfly, the set it covers at the end - all 32 - does not.
__device__ __forceinline__ float dpp_add(float keep, float send, int ctrl_tag)
{
    // keep + (the partner lane's `send`), the partner chosen by a DPP control; separate calls per control: the control is an immediate
    switch (ctrl_tag) {
    case 0: return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x140, 0xF, 0xF, true));   // row_mirror
    case 1: return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x141, 0xF, 0xF, true));   // row_half_mirror
    case 2: return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    default: return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    }
}
__device__ __forceinline__ float halfwave_rowsum(const f32x16& h0, const f32x16& h1, float w3c, int l31)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float v16[16], v8[8], v4[4], v2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // lanes 0-15 keep the h0 entry and get the partner row's, lanes 16-31 the h1 entry: after the swap both operands sit in the lane that sums them
        const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h0[i] * w3c), __float_as_uint(h1[i] * w3c), false, false);
        v16[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const bool up = (l31 & 8) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = dpp_add(up ? v16[i + 8] : v16[i], up ? v16[i] : v16[i + 8], 0);
    }
    {
        const bool up = (l31 & 4) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v4[i] = dpp_add(up ? v8[i + 4] : v8[i], up ? v8[i] : v8[i + 4], 1);
    }
    {
        const bool up = (l31 & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) v2[i] = dpp_add(up ? v4[i + 2] : v4[i], up ? v4[i] : v4[i + 2], 2);
    }
    const bool up = (l31 & 1) != 0;
    return dpp_add(up ? v2[1] : v2[0], up ? v2[0] : v2[1], 3);
}

template <bool TRAIN, bool XG, int NP = 9, bool STAMPS = false>     // NP: partial products of the forward GEMM - 9 / 8 / 6: three-term bf16 splits (gemm_x9), 3 / 4: fp16 pairs (gemm_f16); XG: the 256-deep GEMMs on the 16-bit
                                                         // matrix cores; STAMPS: per-phase cycle stamps of workgroup 0 (scripts/phase_probe.py: nl_decoder_set_debug_buffer)
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decoder(DecArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[S_ALLOC];
    // XG: the three bf16 planes of H1 occupy the first 101 KB.  Once the forward GEMM has consumed them, the bf16 mask
    // tile (phases E/F) aliases plane 0 and the fp32 dH1 tile (phases H/I) aliases planes 1-2: F reads and H writes
    // never overlap, so no barrier separates them.
    float* sH1 = lds + S_H1; float* sD = XG ? lds + X_PLANE_BYTES / 4 : lds + S_D; float* sW1 = lds + S_W1;
    float* sS = lds + S_S;

    constexpr bool F16 = XG && NP <= 4;                   // fp16 pairs: every 16-bit operand plane below is one of two (hi, lo) instead of one of three
    // F16: the dgrad accumulators are 2^12 x dH1 / dsdf (the scale of the W2H planes); dH1 goes through phases H / I with that factor and
    // it is taken out where the results leave (dX store, the dW1 / db1 slab) - powers of two: exact
    constexpr float DHS = F16 ? 1.0f / NL_F16_SG : 1.0f;
    // F16, phases H / I (round 5): Q[i][k] = [H1 > 0] * (dgrad accumulator) = 2^10 dH1[i][k] / dsdf_i goes on as an fp16 pair -
    //   dX[i][c]  = dsdf_i 2^-18 * sum_k Q[i][k] (2^8 W1[k][c])               (Q planes in LDS, 16x16x32 matrix instructions, waves 0-3)
    //   dW1[k][c] += 2^-14 / sigma_t * sum_i Q[i][k] U[i][c],  U = sigma_t dsdf_i * 2^4 X[i][c]   (A fragments = the lane's own accumulator registers, no LDS)
    //   db1[k]    += 2^-10 / sigma_t * sum_i Q[i][k] U[i][16], U[i][16] = sigma_t dsdf_i          (a 17th column of the same matrix instruction)
    // sigma_t = the power of two that puts the TILE's largest |dsdf| in [8, 16).  U lives behind the (single-buffered) fp32 X tile.
    constexpr int U_CHUNK = 17 * 16, U_PLANE = 8 * U_CHUNK;               // bytes: 8 (k-step, lane half) chunks of 17 fragments per plane
    constexpr int U_OFF = S_X * 4 + DEC_M * LDX * 4;
    static_assert(2 * U_PLANE <= DEC_M * LDX * 4, "the U planes fit the second X buffer");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const int col = 32 * w + l31;                 // this lane's output column in 32x32 tiles
    float* sdS = lds + S_DS + DEC_M * w;          // this WAVE's copy of the tile's dL/dsdf (written and read by the same wave)
    const NlLossScalars ls = *a.ls;
    const int P = ls.P;
    const int ntiles = (P + DEC_M - 1) / DEC_M;

    const i32x4 rsW2 = make_w_rsrc(a.params + NL_OFF_W2), rsW2T = make_w_rsrc(a.W2T);
    const i32x4 rsW2X = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W2X_OFF), 0, 3 * W2X_PLANE_BYTES, 0x00020000);
    const i32x4 rsW2TX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W2TX_OFF), 0, 3 * W2X_PLANE_BYTES, 0x00020000);
    const i32x4 rsW2H = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W2H_OFF), 0, 2 * W2X_PLANE_BYTES, 0x00020000);
    const i32x4 rsW2TH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W2TH_OFF), 0, 2 * W2X_PLANE_BYTES, 0x00020000);
    const float b1c = a.params[NL_OFF_B1 + col], b2c = a.params[NL_OFF_B2 + col], w3c = a.params[NL_OFF_W3 + col];
    const float b3 = a.params[NL_OFF_B3];

    // persistent weight-gradient accumulators (dW2 lives in k_decoder_wgrad2)
    f32x4 accW1[4];
    f32x16 accW1h;                                        // F16: this wave's 32 rows of dW1 (lanes 0-15: the channel) and of db1 (lane 16), 2^-14 / 2^-10 taken out per tile
    float aW3 = 0.f, aB2 = 0.f, aB1 = 0.f, aB3 = 0.f, dsMax = 0.f;
    double lossFs = 0.0, lossSdf = 0.0;
    if (TRAIN) {
#pragma unroll
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) accW1[t][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accW1h[r] = 0.f;
    }
    // software prefetch of the next tiles' inputs: the X slice (2 floats per thread) and the loss inputs of row `lane`
    // (every wave keeps its own copy: the loss gradient is recomputed per wave, which removes a workgroup barrier).  The loss inputs are a
    // two-level chain - sample -> its ray -> the ray's cos / range - and a load that waits for its address stalls the wave for a memory
    // latency at the end of every tile (round 5: two such waits per tile in front of the tile's last barrier).  So the chain is spread over
    // two tiles: a tile issues the RAY id of the tile three ahead and, with the id fetched a tile ago, the depth / cos / range loads of the
    // tile two ahead; the product depth * cos is formed where the values are consumed, a whole tile after their loads were issued.
    const int xe = tid * 2, xi = xe >> 4, xc = xe & 15;
    float2 xv = make_float2(0.f, 0.f);
    float pdep = 0.f, pcos = 0.f, pd = 0.f;
    int pray = -1;
    auto ray_of = [&](int tile) {
        const int row0 = tile * DEC_M;
        return (tile < ntiles && row0 + lane < P) ? a.s_ray[row0 + lane] : -1;
    };
    auto prefetch_ray = [&](int tile) { pray = ray_of(tile); };
    auto prefetch = [&](int tile) {                       // `pray` holds the ray id of this tile's row (prefetch_ray, a tile earlier)
        const int row0 = tile * DEC_M;
        xv = make_float2(0.f, 0.f); pdep = 0.f; pcos = 0.f; pd = 0.f;
        if (tile < ntiles) {
            if (row0 + xi < P) xv = *reinterpret_cast<const float2*>(a.X + (size_t)(row0 + xi) * NL_C + xc);
            if (pray >= 0) { pdep = a.s_depth[row0 + lane]; pcos = a.cos_gt[pray]; pd = a.gt_dist[pray]; }
        }
    };
    const int pray_second = ray_of(blockIdx.x + gridDim.x);       // (the first two tiles' ray ids leave together: one latency, not two)
    prefetch_ray(blockIdx.x);
    prefetch(blockIdx.x);
    // (W1 -> LDS under the first tile's input loads: the loss inputs are a two-level dependent chain; at the live shapes a workgroup
    //  sees two tiles in all and the kernel's start is on the critical path of the step)
    unsigned char* const ldsb = reinterpret_cast<unsigned char*>(lds);
    if (F16) {
        // dX's B operand: W1 * 2^8 as fp16-pair fragments of the 16x16x32 matrix instruction, [k-step s(8)][plane(2)][lane(64)][16 B]:
        // lane (c = lane & 15, q = lane >> 4) holds k = 32 s + 8 q + e (e = 0..7) of channel c
        const int s_ = tid >> 6, c_ = lane & 15, q_ = lane >> 4;
        unsigned qh[4], ql[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k_ = 32 * s_ + 8 * q_ + 2 * e;
            split2_pair_f16(sat_f16(a.params[NL_OFF_W1 + k_ * NL_C + c_] * NL_F16_SW1), sat_f16(a.params[NL_OFF_W1 + (k_ + 1) * NL_C + c_] * NL_F16_SW1), &qh[e], &ql[e]);
        }
        uint4* dstw = reinterpret_cast<uint4*>(ldsb + S_W1 * 4 + (s_ * 2 * 64 + lane) * 16);
        dstw[0] = make_uint4(qh[0], qh[1], qh[2], qh[3]); dstw[64] = make_uint4(ql[0], ql[1], ql[2], ql[3]);
    } else {
        for (int i = tid; i < NL_W * NL_C; i += DEC_THREADS) sW1[i] = a.params[NL_OFF_W1 + i];
    }
    // X tile of the next phase B: fp32 (dW1 of the trainable decoder reads it in phase I) and, in the bf16 mode, three bf16 planes
    unsigned xw = 0u;                                   // F16: range watch (nl_common.h nl_range_check)
    auto stage_x = [&](float* sXf) {
        if (!XG || TRAIN) { sXf[xi * LDX + xc] = xv.x; sXf[xi * LDX + xc + 1] = xv.y; }      // (train: dW1 needs the fp32 values)
        if (F16) xw = nl_xw_update(xw, xv.x, xv.y);
        if (F16) {
            unsigned q0, q1;
            split2_pair_f16(sat_f16(xv.x * NL_F16_SX), sat_f16(xv.y * NL_F16_SX), &q0, &q1);
            const int o = opaque(xi * 32 + 2 * xc);
            *reinterpret_cast<unsigned*>(ldsb + XG_XP01_OFF + o) = q0;
            *reinterpret_cast<unsigned*>(ldsb + XG_XP01_OFF + XG_XP_BYTES + o) = q1;
        } else if (XG) {
            unsigned q0, q1, q2;
            split3_pair(xv.x, xv.y, &q0, &q1, &q2);
            const int o = opaque(xi * 32 + 2 * xc);
            *reinterpret_cast<unsigned*>(ldsb + XG_XP01_OFF + o) = q0;
            *reinterpret_cast<unsigned*>(ldsb + XG_XP01_OFF + XG_XP_BYTES + o) = q1;
            *reinterpret_cast<unsigned*>(ldsb + XG_XP2_OFF + o) = q2;
        }
    };
    if (F16) {  // this lane's B fragments of layer 1, parked in LDS (the kernel has no registers to spare)
        uint4 wq[2];
        l1_w1_fragments_f16(a.params, col, lh, wq);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<uint4*>(ldsb + XG_W1X_OFF + ((w * 3 + pl) * 64 + lane) * 16) = wq[pl];
    } else if (XG) {
        uint4 wq[3];
        l1_w1_fragments(a.params, col, lh, wq);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(ldsb + XG_W1X_OFF + ((w * 3 + pl) * 64 + lane) * 16) = wq[pl];
    }
    stage_x(lds + S_X);                                  // phase A of the first tile: X -> LDS buffer 0
    float cz = pdep * pcos, cd = pd;
    pray = pray_second;
    prefetch(blockIdx.x + gridDim.x);
    prefetch_ray(blockIdx.x + 2 * gridDim.x);
    __syncthreads();

    int tile_no = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++tile_no) {
        const int row0 = tile * DEC_M;
        float* sX = lds + S_X + (F16 ? 0 : (tile_no & 1) * (DEC_M * LDX));      // (F16: one fp32 X buffer - it is read in phase E only - the U planes behind it)
        DBG_STAMP(0);
        DBG_STAMP(1);
        // ---------------- B: H1 = relu(X W1^T + b1) ----------------
        unsigned m1 = 0u;                               // XG: this lane's 32 ReLU bits of H1
        uint4 bq9[X9_RING][3];
        uint4 bqh[F16_RING][2];
        {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
            if (F16) {
                const unsigned char* xq = ldsb + opaque(l31 * 32 + 16 * lh);
                const unsigned char* wq = ldsb + opaque(XG_W1X_OFF + (w * 3 * 64 + lane) * 16);
                uint4 xa0[2], xa1[2], wf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    xa0[pl] = *reinterpret_cast<const uint4*>(xq + XG_XP01_OFF + pl * XG_XP_BYTES);
                    xa1[pl] = *reinterpret_cast<const uint4*>(xq + XG_XP01_OFF + pl * XG_XP_BYTES + 32 * 32);
                    wf[pl] = *reinterpret_cast<const uint4*>(wq + pl * 64 * 16);
                }
#pragma unroll
                for (int pa = 1; pa >= 0; --pa)               // all four products, the smallest first (K = 16: two dozen pipe cycles)
#pragma unroll
                    for (int pq = 1; pq >= 0; --pq) { c0 = mma16<true>(xa0[pa], wf[pq], c0); c1 = mma16<true>(xa1[pa], wf[pq], c1); }
            } else if (XG) {
                const unsigned char* xq = ldsb + opaque(l31 * 32 + 16 * lh);
                const unsigned char* wq = ldsb + opaque(XG_W1X_OFF + (w * 3 * 64 + lane) * 16);
                bf16x8 xa0[3], xa1[3], wf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const int po = pl < 2 ? XG_XP01_OFF + pl * XG_XP_BYTES : XG_XP2_OFF;
                    xa0[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xq + po));
                    xa1[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xq + po + 32 * 32));
                    wf[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wq + pl * 64 * 16));
                }
#pragma unroll
                for (int pa = 2; pa >= 0; --pa)               // smallest terms first
#pragma unroll
                    for (int pq = 2; pq >= 0; --pq) { c0 = MFMA_BF16(xa0[pa], wf[pq], c0); c1 = MFMA_BF16(xa1[pa], wf[pq], c1); }
            } else {
                const float* xb = sX + opaque(l31 * LDX + lh);
                const float* wb = sW1 + opaque(col * NL_C + lh);
#pragma unroll
                for (int kk = 0; kk < NL_C / 2; ++kk) {
                    const float bw = wb[2 * kk];
                    c0 = MFMA32(xb[2 * kk], bw, c0); c1 = MFMA32(xb[32 * LDX + 2 * kk], bw, c1);
                }
            }
            if (F16) {
                gemm_f16_prefetch(rsW2TH, w, lane, bqh);         // W2 planes of the first k-steps: in flight across the barrier
                if (tile_no == 0 && h1_clips(c0, c1, b1c * NL_F16_SH)) xw = nl_xw_mark(xw, NL_SAT_H1);
                m1 = store_h1_planes_f16(reinterpret_cast<unsigned short*>(lds), col, lh, c0, c1, b1c * NL_F16_SH);
            } else if (XG) {
                gemm_x9_prefetch(rsW2TX, w, lane, bq9);          // W2 planes of the first k-steps: in flight across the barrier
                m1 = store_h1_planes(reinterpret_cast<unsigned short*>(lds), col, lh, c0, c1, b1c);
            } else {
                float* hb = sH1 + opaque(4 * lh * LDH + col);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hb[D32_RR(r) * LDH] = fmaxf(c0[r] + b1c, 0.f);
                    hb[(32 + D32_RR(r)) * LDH] = fmaxf(c1[r] + b1c, 0.f);
                }
            }
        }
        nl_lds_barrier();
        DBG_STAMP(2);
        // ---------------- C: H2 = relu(H1 W2^T + b2), s = H2 w3 + b3 ----------------
        f32x16 h0, h1;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
            if constexpr (F16) gemm_f16<true, NP>(rsW2TH, w, lane, reinterpret_cast<const unsigned char*>(lds), bqh, h0, h1);
            else if (XG) gemm_x9<true, F16 ? 9 : NP>(rsW2TX, w, lane, reinterpret_cast<const unsigned char*>(lds), bq9, h0, h1);
            else    gemm256(rsW2T, (lh * NL_W + col) * 4, sH1 + l31 * LDH + lh, sH1 + (32 + l31) * LDH + lh, h0, h1);
            DBG_STAMP(3);
            constexpr float S2 = 1.0f / (NL_F16_SH * NL_F16_SW2);     // fp16 pairs: the accumulators are 2^12 x H1 W2^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h0[r] = fmaxf(F16 ? fmaf(h0[r], S2, b2c) : h0[r] + b2c, 0.f); h1[r] = fmaxf(F16 ? fmaf(h1[r], S2, b2c) : h1[r] + b2c, 0.f);
            }
            const float tot = halfwave_rowsum(h0, h1, w3c, l31);               // entry e = l31 of [h0 rows | h1 rows]
            sS[w * DEC_M + (l31 >> 4) * 32 + d32_row(l31 & 15, lh)] = tot;     // this wave's 32 columns of 64 distinct rows
        }
        nl_lds_barrier();
        DBG_STAMP(4);
        // ---------------- D: sdf, loss gradient (criterion.py): every wave computes all 64 rows (lane = row) for itself;
        //                  wave 0 owns the global outputs and the loss sums ----------------
        {
            const int g = row0 + lane;
            float ds = 0.f;
            if (g < P) {
                float s = sS[lane];                             // fixed summation order: run-to-run reproducible sdf
#pragma unroll
                for (int ww = 1; ww < 8; ++ww) s += sS[ww * DEC_M + lane];
                s += b3;
                bool f, m;
                nl_loss_masks(cz, cd, ls.tau, ls.max_depth, &f, &m);
                float q1, q2;
                ds = nl_loss_grad(s, cz, cd, f, m, ls, &q1, &q2);
                if (w == 0) {
                    a.sdf[g] = s; a.dsdf[g] = ds;
                    lossFs += (double)q1; lossSdf += (double)q2;
                }
            }
            sdS[lane] = ds;
            if (TRAIN && w == 0) { aB3 += ds; dsMax = fmaxf(dsMax, fabsf(ds)); }
            __builtin_amdgcn_wave_barrier();                    // same-wave LDS write -> read: in order, keep the compiler from reordering
        }
        DBG_STAMP(5);
        float inv_sigma = 1.0f;                          // F16 train: 1 / sigma_t of this tile
        if (F16 && TRAIN) {
            const float mx = wave_max_nonneg(fabsf(sdS[lane]));
            int e = 0;
            if (mx > 0.f) { (void)frexpf(mx, &e); e = e < -100 ? -100 : e; }        // mx = m 2^e, m in [0.5, 1): sigma_t = 2^(4 - e)
            const float sigma = ldexpf(1.0f, 4 - e);
            inv_sigma = ldexpf(1.0f, e - 4);
            if (tid < 2 * 8 * 17) {
                // half a B fragment of dW1's matrix instructions per thread: (k-step t, lane half hh, column c) = 8 sample rows of U, in the order the
                // A fragments - accumulator registers 8 (t & 1) .. + 7 of sub-tile t >> 1 - hold their rows; thread parity = which four of the eight
                const int f = tid >> 1, half = tid & 1, t = f / 34, hh = (f / 17) & 1, c = f % 17;
                const float* xr = sX + opaque(c < NL_C ? c : 0);
                const int row = 16 * (t & 1) + 4 * hh + 32 * (t >> 1) + 8 * half;       // rows of elements 4 half .. 4 half + 3: consecutive
                float d[4], u[4];
#pragma unroll
                for (int z = 0; z < 4; ++z) { d[z] = sdS[row + z] * sigma; u[z] = xr[(row + z) * LDX]; }
#pragma unroll
                for (int z = 0; z < 4; ++z) u[z] = c < NL_C ? sat_f16(d[z] * 16.0f * u[z]) : d[z];
                unsigned uh[2], ul[2];
                split2_pair_f16(u[0], u[1], &uh[0], &ul[0]); split2_pair_f16(u[2], u[3], &uh[1], &ul[1]);
                unsigned char* ud = ldsb + opaque(U_OFF + (t * 2 + hh) * U_CHUNK + c * 16 + 8 * half);
                *reinterpret_cast<uint2*>(ud) = make_uint2(uh[0], uh[1]);
                *reinterpret_cast<uint2*>(ud + U_PLANE) = make_uint2(ul[0], ul[1]);
            }
        }
        // ---------------- E: dH2 = ds * w3 * [H2 > 0] -> LDS ----------------
        uint4 bqm[MX_RING][3];
        if (F16) gemm_mask_x_prefetch<2>(rsW2H, w, lane, bqm);
        else if (XG) gemm_mask_x_prefetch<3>(rsW2X, w, lane, bqm);
        {
            unsigned mw = 0u;                       // this lane's 32 ReLU bits: bit r = h0[r] > 0, bit 16+r = h1[r] > 0
            const float* dsb = sdS + opaque(4 * lh);
            float* db = sD + opaque(4 * lh * LDH + col);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ds0 = dsb[D32_RR(r)], ds1 = dsb[32 + D32_RR(r)];
                if constexpr (!XG) {
                    const bool on0 = h0[r] > 0.f, on1 = h1[r] > 0.f;
                    db[D32_RR(r) * LDH] = on0 ? ds0 * w3c : 0.f; db[(32 + D32_RR(r)) * LDH] = on1 ? ds1 * w3c : 0.f;
                }
                // the ReLU bits as integers 0 / 1 (pos_bit: no compare, no scalar lane masks)
                const unsigned on0 = pos_bit(h0[r]), on1 = pos_bit(h1[r]);
                // db2[col] = w3[col] * sum_i [H2 > 0] dsdf_i: the factor is applied once, at the flush; ds * 1 + acc and ds * 0 + acc are the adds they replace, bit for bit
                if (TRAIN) { aW3 = fmaf(ds0, h0[r], aW3); aW3 = fmaf(ds1, h1[r], aW3); aB2 = fmaf(ds0, (float)on0, aB2); aB2 = fmaf(ds1, (float)on1, aB2); }
                if (XG || TRAIN) mw |= (on0 << r) | (on1 << (16 + r));
            }
            if (XG) {
                // the 0/1 mask itself, as bf16, is the dgrad A operand.  Neighbouring lanes = neighbouring columns of the same rows: with
                // the neighbour's 32 bits (one DPP move) a lane stores columns (k, k + 1) of one row as a dword - the even lane the even
                // accumulator rows, the odd lane the odd ones: 16 dword stores instead of 32 half-word stores
                const bool odd = (col & 1) != 0;
                const unsigned nb = dpp_swap1u(mw);
                const unsigned lo_bits = odd ? nb : mw, hi_bits = odd ? mw : nb;        // columns k and k + 1
                unsigned* mb = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(lds) + opaque((4 * lh + (odd ? 1 : 0)) * SM_STRIDE + (col & ~1)));
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const unsigned lb = (odd ? lo_bits >> 1 : lo_bits) >> (16 * sub + r), hb = (odd ? hi_bits >> 1 : hi_bits) >> (16 * sub + r);
                        mb[((32 * sub + D32_RR(r)) * SM_STRIDE) / 2] = ((lb & 1u) ? (F16 ? 0x3C00u : 0x3F80u) : 0u) | ((hb & 1u) ? (F16 ? 0x3C000000u : 0x3F800000u) : 0u);     // 1.0 as fp16 / bf16
                    }
            }
            // one word per thread, thread-major per tile: k_decoder_wgrad2's thread (same wave/lane) reads it back
            if (TRAIN) a.relu2_mask[(size_t)tile * DEC_THREADS + tid] = mw;
        }
        nl_lds_barrier();
        DBG_STAMP(6);
        // ---------------- F: dH1 = (dH2 W2) * [H1 > 0]  (kept in registers) ----------------
        f32x16 g0v, g1v;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { g0v[r] = 0.f; g1v[r] = 0.f; }
            if (F16 && TRAIN) gemm_mask_x_rolled<true, 4>(rsW2H, w, lane, reinterpret_cast<const unsigned char*>(lds), bqm, g0v, g1v);
            else if (F16)    gemm_mask_x<true, true>(rsW2H, w, lane, reinterpret_cast<const unsigned char*>(lds), bqm, g0v, g1v);     // (pinned: the trainable kernel has the registers since its ReLU bits left the scalar file, 243 -> 199)
            else if (XG && TRAIN) gemm_mask_x_rolled(rsW2X, w, lane, reinterpret_cast<const unsigned char*>(lds), bqm, g0v, g1v);
            else if (XG)     gemm_mask_x<true>(rsW2X, w, lane, reinterpret_cast<const unsigned char*>(lds), bqm, g0v, g1v);
            else    gemm256(rsW2, (lh * NL_W + col) * 4, sD + l31 * LDH + lh, sD + (32 + l31) * LDH + lh, g0v, g1v);
            DBG_STAMP(7);
            if constexpr (F16) { if (tile_no == 0 && q_clips(g0v, g1v, m1)) xw = nl_xw_mark(xw, NL_SAT_Q); }
            if constexpr (!F16) {
            const float* dsb = sdS + opaque(4 * lh);
            const float* hb = sH1 + opaque(4 * lh * LDH + col);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (XG) {
                    const float d0 = dsb[D32_RR(r)], d1 = dsb[32 + D32_RR(r)];           // unconditional loads: no exec-mask branches
                    // the ReLU bit as an all-ones / zero word (one v_bfe_i32) ANDed onto the factor: no compare, no branch
                    g0v[r] *= __uint_as_float(__float_as_uint(d0) & (unsigned)__builtin_amdgcn_sbfe((int)m1, r, 1));
                    g1v[r] *= __uint_as_float(__float_as_uint(d1) & (unsigned)__builtin_amdgcn_sbfe((int)m1, 16 + r, 1));
                } else {
                    g0v[r] = hb[D32_RR(r) * LDH] > 0.f ? g0v[r] : 0.f;
                    g1v[r] = hb[(32 + D32_RR(r)) * LDH] > 0.f ? g1v[r] : 0.f;
                }
                if (TRAIN) aB1 += g0v[r] + g1v[r];
            }
            }
        }
        if (!XG) nl_lds_barrier();                       // fp32 path: dH1 overwrites the dH2 tile other waves may still be reading
        DBG_STAMP(8);
        if constexpr (F16) {
            // ---------------- H (fp16 pairs): Q = [H1 > 0] * accumulator, saturated, as an fp16 pair.  A lane's packed (row, row + 1) words of its
            // own column ARE dW1's A fragments; exchanged with the neighbouring column's (one DPP move + one byte permute per word) they are the
            // (k, k + 1) words of one row of the Q planes, dX's A operand ----------------
            // (train) dW1 / db1 right here, k-step by k-step: the three matrix instructions of a k-step are issued as soon as its eight registers are
            // converted - they run in the matrix pipe under the conversion of the next k-step, and no fragment outlives its k-step (the U planes were
            // published by the barrier in front of phase F)
            f32x16 tw;
            if (TRAIN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tw[r] = 0.f;
            }
            {
                const bool odd = (col & 1) != 0;
                const unsigned sel = odd ? 0x03020706u : 0x05040100u;       // odd: (neighbour.hi16, own.hi16) = row r + 1; even: (own.lo16, neighbour.lo16) = row r
                unsigned* pq = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(lds) + opaque(X_PLANE_ELEMS + (4 * lh + (odd ? 1 : 0)) * SM_STRIDE + (col & ~1)));
                const unsigned char* ub = ldsb + opaque(U_OFF + lh * U_CHUNK + (l31 < 17 ? l31 : 16) * 16);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int sub = t >> 1;
                    uint4 qh, ql, bh, bl;
                    if (TRAIN) { bh = *reinterpret_cast<const uint4*>(ub + 2 * t * U_CHUNK); bl = *reinterpret_cast<const uint4*>(ub + 2 * t * U_CHUNK + U_PLANE); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 8 * (t & 1) + 2 * e;
                        // saturate (one v_med3), then the ReLU bit as an all-ones / zero word ANDed on (v_bfe_i32 + v_and): no compare, no branch
                        const float qa = __uint_as_float(__float_as_uint(sat_f16(sub ? g1v[r] : g0v[r])) & (unsigned)__builtin_amdgcn_sbfe((int)m1, 16 * sub + r, 1));
                        const float qb = __uint_as_float(__float_as_uint(sat_f16(sub ? g1v[r + 1] : g0v[r + 1])) & (unsigned)__builtin_amdgcn_sbfe((int)m1, 16 * sub + r + 1, 1));
                        unsigned ph, pl;
                        split2_pair_f16(qa, qb, &ph, &pl);
                        (&qh.x)[e] = ph; (&ql.x)[e] = pl;
                        unsigned* d = pq + ((32 * sub + D32_RR(r)) * SM_STRIDE) / 2;
                        d[0] = __builtin_amdgcn_perm(dpp_swap1u(ph), ph, sel);
                        d[X_PLANE_ELEMS / 2] = __builtin_amdgcn_perm(dpp_swap1u(pl), pl, sel);
                    }
                    if (TRAIN) { tw = mma16<true>(ql, bh, tw); tw = mma16<true>(qh, bl, tw); tw = mma16<true>(qh, bh, tw); }
                }
            }
            if (TRAIN) {
                const float sc = inv_sigma * (l31 == 16 ? 1.0f / NL_F16_SG : 1.0f / (NL_F16_SG * 16.0f));
#pragma unroll
                for (int r = 0; r < 16; ++r) accW1h[r] = fmaf(tw[r], sc, accW1h[r]);
            }
            nl_lds_barrier();
            DBG_STAMP(9);
            // ---------------- I (fp16 pairs): waves 0-3: dX of 16 rows each ----------------
            if (w < 4) {
                // two accumulator chains (k-step parity), the fragments of the next k-step in flight under this one's three instructions: a plain
                // unrolled loop has every ds_read sunk to just before its first use, i.e. one LDS latency per k-step on a dependent chain
                f32x4 cxa = {0.f, 0.f, 0.f, 0.f}, cxb = {0.f, 0.f, 0.f, 0.f};
                const unsigned char* ap = ldsb + opaque(X_PLANE_BYTES + (16 * w + l15) * (SM_STRIDE * 2) + 16 * lq);
                const unsigned char* bp = ldsb + opaque(S_W1 * 4 + lane * 16);
                uint4 fa[2][2], fb[2][2];                 // [k-step parity][hi, lo]
                auto fetch = [&](int s8, int buf) {
                    fa[buf][0] = *reinterpret_cast<const uint4*>(ap + 64 * s8); fa[buf][1] = *reinterpret_cast<const uint4*>(ap + X_PLANE_BYTES + 64 * s8);
                    fb[buf][0] = *reinterpret_cast<const uint4*>(bp + s8 * 2048); fb[buf][1] = *reinterpret_cast<const uint4*>(bp + s8 * 2048 + 1024);
                };
                fetch(0, 0);
                const float4 dsv = *reinterpret_cast<const float4*>(sdS + opaque(16 * w + 4 * lq));      // dsdf of this lane's four rows: one read, under the loop
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int cur = s8 & 1;
                    if (s8 + 1 < 8) fetch(s8 + 1, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const f16x8 ah = __builtin_bit_cast(f16x8, fa[cur][0]), al = __builtin_bit_cast(f16x8, fa[cur][1]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, fb[cur][0]), bl = __builtin_bit_cast(f16x8, fb[cur][1]);
                    if (cur) { cxb = MFMA16_F16(al, bh, cxb); cxb = MFMA16_F16(ah, bl, cxb); cxb = MFMA16_F16(ah, bh, cxb); }
                    else     { cxa = MFMA16_F16(al, bh, cxa); cxa = MFMA16_F16(ah, bl, cxa); cxa = MFMA16_F16(ah, bh, cxa); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4 cx = cxa + cxb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = row0 + 16 * w + 4 * lq + r;
                    if (g < P) a.dX[(size_t)g * NL_C + l15] = cx[r] * ((&dsv.x)[r] * (1.0f / (NL_F16_SG * NL_F16_SW1)));
                }
            }
        } else {
        // ---------------- H: dH1 -> LDS ----------------
        {
            float* db = sD + opaque(4 * lh * LDH + col);
#pragma unroll
            for (int r = 0; r < 16; ++r) { db[D32_RR(r) * LDH] = g0v[r]; db[(32 + D32_RR(r)) * LDH] = g1v[r]; }
        }
        nl_lds_barrier();
        DBG_STAMP(9);
        // ---------------- I: waves 0-3: dX[16 rows each] = dH1 W1 -> global ; waves 4-7: dW1 += dH1^T X ----------------
        if (w < 4) {
            // 64 MFMA16 in two accumulator chains; operands double-buffered 8 k-steps ahead through a rolled loop (an unrolled
            // loop lets the scheduler sink every ds_read to just before its MFMA and the chain stalls on LDS latency)
            f32x4 cxa = {0.f, 0.f, 0.f, 0.f}, cxb = {0.f, 0.f, 0.f, 0.f};
            const float* ap = sD + opaque((16 * w + l15) * LDH + lq);
            const float* bq = sW1 + opaque(lq * NL_C + l15);
            float aA[8], bA[8], aB[8], bB[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * i]; bA[i] = bq[4 * i * NL_C]; }
#pragma unroll 1
            for (int q = 0; q < NL_W / 4; q += 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { aB[i] = ap[4 * (q + 8 + i)]; bB[i] = bq[4 * (q + 8 + i) * NL_C]; }
#pragma unroll
                for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aA[i], bA[i], cxa); cxb = MFMA16(aA[i + 1], bA[i + 1], cxb); }
                if (q + 16 < NL_W / 4) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * (q + 16 + i)]; bA[i] = bq[4 * (q + 16 + i) * NL_C]; }
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aB[i], bB[i], cxa); cxb = MFMA16(aB[i + 1], bB[i + 1], cxb); }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = row0 + 16 * w + 4 * lq + r;
                if (g < P) a.dX[(size_t)g * NL_C + l15] = cxa[r] + cxb[r];
            }
        } else if (TRAIN) {
            const float* xr = sX + opaque(lq * LDX + l15);
            const float* dr = sD + opaque(lq * LDH + 64 * (w - 4) + l15);
            float xA[2], dA[2][4], xB[2], dB[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                xA[u] = xr[4 * u * LDX];
#pragma unroll
                for (int t = 0; t < 4; ++t) dA[u][t] = dr[4 * u * LDH + 16 * t];
            }
#pragma unroll 1
            for (int ii = 0; ii < DEC_M / 4; ii += 4) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    xB[u] = xr[4 * (ii + 2 + u) * LDX];
#pragma unroll
                    for (int t = 0; t < 4; ++t) dB[u][t] = dr[4 * (ii + 2 + u) * LDH + 16 * t];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) accW1[t] = MFMA16(dA[u][t], xA[u], accW1[t]);
                if (ii + 4 < DEC_M / 4) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        xA[u] = xr[4 * (ii + 4 + u) * LDX];
#pragma unroll
                        for (int t = 0; t < 4; ++t) dA[u][t] = dr[4 * (ii + 4 + u) * LDH + 16 * t];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) accW1[t] = MFMA16(dB[u][t], xB[u], accW1[t]);
            }
        }
        }
        // ---------------- A (next tile): X -> the other LDS buffer; issue the loads of the tile after it ----------------
        {
            stage_x(lds + S_X + (F16 ? 0 : ((tile_no + 1) & 1) * (DEC_M * LDX)));
            cz = pdep * pcos; cd = pd;
            prefetch(tile + 2 * gridDim.x);
            prefetch_ray(tile + 3 * gridDim.x);
        }
        nl_lds_barrier();
        DBG_STAMP(10);
    }

    if constexpr (F16) nl_range_check(a.W2T, xw, TRAIN);       // did an operand of the fp16-pair arithmetic leave its range?  (sticky status word of the weight workspace)
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
        if constexpr (F16) {
            // lane c < 16: dW1[32 w + row][c]; lane 16: db1[32 w + row]  (row of accumulator register r: D32_RR(r) + 4 lh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * w + D32_RR(r) + 4 * lh;
                if (l31 < NL_C) base[NL_OFF_W1 + k * NL_C + l31] = accW1h[r];
                else if (l31 == NL_C) base[NL_OFF_B1 + k] = accW1h[r];
            }
        } else if (w >= 4) {
            const int hb = 64 * (w - 4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    base[NL_OFF_W1 + (hb + 16 * t + 4 * lq + r) * NL_C + l15] = accW1[t][r];
        }
        aW3 += __shfl_xor(aW3, 32); aB2 += __shfl_xor(aB2, 32); aB1 += __shfl_xor(aB1, 32);
        if (lh == 0) { base[NL_OFF_W3 + col] = aW3; base[NL_OFF_B2 + col] = aB2 * w3c; if (!F16) base[NL_OFF_B1 + col] = aB1; }
        if (tid < 64) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { aB3 += __shfl_xor(aB3, off); dsMax = fmaxf(dsMax, __shfl_xor(dsMax, off)); }
            if (tid == 0) {
                base[NL_OFF_B3] = aB3;
                // max |dL/dsdf| of the launch (non-negative floats order like their bit patterns): the scale of the fp16-pair dW2 kernel
                if (dsMax > 0.f) atomicMax(const_cast<unsigned*>(&a.ls->ds_max_bits), __float_as_uint(dsMax));
            }
        }
    }
}

// =============================================================================================================================
// k_decoder2 (round 6): the fp16-pair decoder as TWO INDEPENDENT 4-wave workgroups per CU.
//
// Why: k_decoder runs 8 waves (2 per SIMD) through five barrier-separated phases per tile, so the two waves of a SIMD are always in the
// SAME phase - its ~1 300 VALU instructions per wave and tile (operand splits, ReLU bits, row sums) never run under the other wave's
// matrix instructions, and the matrix pipe is busy 47 % of the time (profiles/r05_z_decoder_phase_cycles.txt: 23.9 k cycles per tile,
// 11.9 k of them matrix-pipe time).  Two workgroups that share no barrier drift apart: while one is in a VALU phase the other is in a
// GEMM.  Same arithmetic per element as k_decoder<.., true, 3 / 4> (sdf, dsdf and the ReLU masks are bit-identical; dX and the weight
// gradients differ by summation order only).
//
// What had to change to fit two workgroups into one CU's 160 KB of LDS (79 360 B each):
//   * a wave owns TWO 32-column tiles (ct = 2 w, 2 w + 1): the GEMMs run on four accumulators per wave and share every A fragment
//     between the two column tiles (half the LDS reads per matrix instruction); the per-element code runs per column tile with the
//     lane map of k_decoder, so the saved ReLU words have k_decoder's layout (k_decoder_wgrad2* read them unchanged);
//   * two 33 KB plane regions instead of three: H1 hi / lo in phase C; the 0/1 mask tile (P0) in phases E / F; Q = [H1 > 0] x dgrad
//     accumulator goes to P1 in HALF tiles (32 rows: hi in rows 0-31, lo in rows 32-63 of the region), and only into the wave's OWN 64
//     columns: dX[i][c] = sum_k Q[i][k] W1[k][c] is split over k by wave - a wave multiplies its own 64 columns of Q (written and read
//     back by the same wave: no workgroup barrier between F, H and I) into a partial [64][16] tile, the four partial tiles are summed in a
//     fixed order after the tile's last barrier;
//   * W1's two operand forms (layer-1 B fragments, dX B fragments: 16 registers each) stay in registers; U aliases the fp16 X planes.
// Barriers per tile: 5, as before - but a barrier now stalls 4 waves while the CU's other 4 keep issuing.
// =============================================================================================================================
#define D2_THREADS 256
#define D2_P1 X_PLANE_BYTES
#define D2_XF (2 * X_PLANE_BYTES)                       // fp32 X tile [64][17] (train: U = sigma dsdf 16 X)
#define D2_XU (D2_XF + DEC_M * LDX * 4)                 // fp16 X planes [2][64][32 B] (phase B)  /  U planes (phases D - H)
#define D2_U_CHUNK (17 * 16)
#define D2_U_PLANE (8 * D2_U_CHUNK)
#define D2_SS (D2_XU + 2 * D2_U_PLANE)                  // [8 column tiles][64] partial row sums
#define D2_SDS (D2_SS + 8 * DEC_M * 4)                  // [4 waves][64] dL/dsdf, one copy per wave
#define D2_LOSS (D2_SDS + 4 * DEC_M * 4)                // [64 lanes of wave 0][2] fp64 loss sums (in LDS: four registers less across the tile loop)
#define D2_TOTAL (D2_LOSS + DEC_M * 16)
static_assert(2 * XG_XP_BYTES <= 2 * D2_U_PLANE, "the fp16 X planes fit the U region");
static_assert(2 * D2_TOTAL <= 163840, "two workgroups per CU");

#ifndef F2_RING
#define F2_RING 3                                       // forward: B fragments 3 k-steps (12-16 matrix instructions each) ahead
#endif
__device__ __forceinline__ void gemm_f16_2ct_prefetch(i32x4 rsH, int w, int lane, uint4 (&bq)[F2_RING][2][2])
{
    const int voff = lane * 16;
    const int nt_off = __builtin_amdgcn_readfirstlane(w) * 2 * 16 * 1024;
#pragma unroll
    for (int s = 0; s < F2_RING - 1; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) bq[s][j][p] = bload4(rsH, voff, p * W2X_PLANE_BYTES + nt_off + j * 16 * 1024 + s * 1024);
}

// H2pre of this wave's two column tiles x two row tiles: every accumulator sees k_decoder's sequence (lo x lo with NP = 4, lo x hi, hi x lo, hi x hi per
// k-step); the four accumulators interleave, the A fragments (four ds_read_b128 per k-step) serve both column tiles
template <int NP>
__device__ __forceinline__ void gemm_f16_2ct(i32x4 rsH, int w, int lane, const unsigned char* sP, uint4 (&bq)[F2_RING][2][2], f32x16 (&c)[2][2])
{
    static_assert(NP == 3 || NP == 4, "three or four partial products");
    const int l31 = lane & 31, lh = lane >> 5;
    const int voff = lane * 16;
    const int nt_off = __builtin_amdgcn_readfirstlane(w) * 2 * 16 * 1024;
    const unsigned char* a0 = sP + opaque(l31 * (SM_STRIDE * 2) + 16 * lh);
    constexpr int RING = F2_RING;
    // A fragments single-buffered (16 registers, not 32 - the kernel sits at the register limit): the lo pair of the next k-step is requested once this
    // k-step's lo products are issued (its registers are free from there on), the hi pair after the last product; the next k-step opens with the
    // lo products, so the hi pair has four matrix instructions to arrive
    uint4 ahi[2], alo[2];
    ahi[0] = *reinterpret_cast<const uint4*>(a0); ahi[1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2);
    alo[0] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES); alo[1] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES + 32 * SM_STRIDE * 2);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (NP == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { c[j][0] = mma16<true>(alo[0], bq[s % RING][j][1], c[j][0]); c[j][1] = mma16<true>(alo[1], bq[s % RING][j][1], c[j][1]); }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { c[j][0] = mma16<true>(alo[0], bq[s % RING][j][0], c[j][0]); c[j][1] = mma16<true>(alo[1], bq[s % RING][j][0], c[j][1]); }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 16) {
            alo[0] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES + 32 * (s + 1));
            alo[1] = *reinterpret_cast<const uint4*>(a0 + X_PLANE_BYTES + 32 * SM_STRIDE * 2 + 32 * (s + 1));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { c[j][0] = mma16<true>(ahi[0], bq[s % RING][j][1], c[j][0]); c[j][1] = mma16<true>(ahi[1], bq[s % RING][j][1], c[j][1]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) { c[j][0] = mma16<true>(ahi[0], bq[s % RING][j][0], c[j][0]); c[j][1] = mma16<true>(ahi[1], bq[s % RING][j][0], c[j][1]); }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 16) {
            ahi[0] = *reinterpret_cast<const uint4*>(a0 + 32 * (s + 1));
            ahi[1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2 + 32 * (s + 1));
        }
        if (s + RING - 1 < 16) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    bq[(s + RING - 1) % RING][j][p] = bload4(rsH, voff, p * W2X_PLANE_BYTES + nt_off + j * 16 * 1024 + (s + RING - 1) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifndef M2_RING
#define M2_RING 3                                       // dgrad: B fragments 3 k-steps (8 matrix instructions each) ahead
#endif
#define M2_PRE 2                                        // of which this many are requested before the barrier that publishes the mask tile
__device__ __forceinline__ void gemm_mask_2ct_prefetch(i32x4 rsX, int w, int lane, uint4 (&bq)[M2_RING][2][2])
{
    const int voff = lane * 16;
    const int kt_off = __builtin_amdgcn_readfirstlane(w) * 2 * 16 * 1024;
#pragma unroll
    for (int s = 0; s < M2_PRE; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) bq[s][j][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + kt_off + j * 16 * 1024 + s * 1024);
}

// dgrad accumulators (2^10 dH1 / dsdf before the ReLU of H1) of the wave's two column tiles: 0/1 mask x (lo, hi) planes of w3_j W2[j][k], the low term first
template <typename Late>                               // late(): issued two k-steps before the loop ends, when a ring stage's registers have become free (k_decoder2: dX's W1 fragments)
__device__ __forceinline__ void gemm_mask_2ct(i32x4 rsX, int w, int lane, const unsigned char* sM, uint4 (&bq)[M2_RING][2][2], f32x16 (&g)[2][2], Late late)
{
    const int l31 = lane & 31, lh = lane >> 5;
    const int voff = lane * 16;
    const int kt_off = __builtin_amdgcn_readfirstlane(w) * 2 * 16 * 1024;
    const unsigned char* a0 = sM + opaque(l31 * (SM_STRIDE * 2) + 16 * lh);
    constexpr int RING = M2_RING;
    static_assert(RING - 1 >= M2_PRE, "ring depth");
    uint4 aq[2][2];
#pragma unroll
    for (int s = M2_PRE; s < RING - 1; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) bq[s][j][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + kt_off + j * 16 * 1024 + s * 1024);
    aq[0][0] = *reinterpret_cast<const uint4*>(a0); aq[0][1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s + RING - 1 < 16) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    bq[(s + RING - 1) % RING][j][p] = bload4(rsX, voff, p * W2X_PLANE_BYTES + kt_off + j * 16 * 1024 + (s + RING - 1) * 1024);
        }
        if (s + 1 < 16) {
            aq[(s + 1) & 1][0] = *reinterpret_cast<const uint4*>(a0 + 32 * (s + 1));
            aq[(s + 1) & 1][1] = *reinterpret_cast<const uint4*>(a0 + 32 * SM_STRIDE * 2 + 32 * (s + 1));
        }
        if (s == 14) late();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 1; p >= 0; --p)
#pragma unroll
            for (int j = 0; j < 2; ++j) { g[j][0] = mma16<true>(aq[s & 1][0], bq[s % RING][j][p], g[j][0]); g[j][1] = mma16<true>(aq[s & 1][1], bq[s % RING][j][p], g[j][1]); }
    }
}

// lane ids of a phase, re-derived from a LAUNDERED thread id: every LDS address of the kernel is a function of the lane id, i.e. loop-invariant, and the
// compiler keeps each of the ~20 distinct ones in a register for the whole persistent loop (or spills and reloads it - a scratch reload queues behind the
// weight stream's loads in flight) where two or three integer instructions per phase recompute it
#define D2_IDS()                                                                                                            \
    const int tid = opaque((int)threadIdx.x), lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5,               \
              l15 = lane & 15, lq = lane >> 4, xi = tid >> 2, xc = (tid & 3) * 4;                                           \
    float* const sdS = reinterpret_cast<float*>(ldsb + D2_SDS) + DEC_M * w;      /* this WAVE's copy of the tile's dL/dsdf */ \
    (void)lane; (void)w; (void)l31; (void)lh; (void)l15; (void)lq; (void)xi; (void)xc; (void)sdS

// wave priority by phase (A/B aid, -DD2_PRIO=n): n > 0 raises a wave's priority to n while it is in a conversion / epilogue phase and drops it to 0 inside the GEMM loops
#ifndef D2_PRIO
#define D2_PRIO 4
#endif
// n = 4: the CU's two workgroups (blocks b and b + gridDim.x / 2 start together on one CU) take turns, tile by tile, at being the one the arbiter prefers when both
// want the same pipe: with equal priorities the OLDER workgroup always wins and finishes its 33 tiles ~20 % earlier - the kernel then ends with half-empty CUs
#define D2_PRIO_VALU() do { if (D2_PRIO == 4) { if (prio_alt) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); } else if (D2_PRIO) __builtin_amdgcn_s_setprio(D2_PRIO); } while (0)
#define D2_PRIO_MFMA() do { if (D2_PRIO == 4) { if (prio_alt) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } else if (D2_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)

template <bool TRAIN, int NP, bool STAMPS = false>
__global__ __launch_bounds__(D2_THREADS, 2) void k_decoder2(DecArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char ldsb[D2_TOTAL];
    float* const sXf = reinterpret_cast<float*>(ldsb + D2_XF);
    float* const sS = reinterpret_cast<float*>(ldsb + D2_SS);
    const NlLossScalars ls = *a.ls;
    const int P = ls.P;
    const int ntiles = (P + DEC_M - 1) / DEC_M;
    // STAMPS: one record per workgroup behind the 256 phase stamps - [wall clock (100 MHz) at start, at end, shader cycles at start, at end, HW_ID, XCC_ID, tiles, -]:
    // which CU a workgroup ran on and with whom, and the clock the kernel held (scripts/decoder_layout_probe.py)
    long long wg_t0 = 0, wg_c0 = 0;
    if constexpr (STAMPS) { if (a.dbg && threadIdx.x == 0) { wg_t0 = (long long)wall_clock64(); wg_c0 = (long long)__builtin_readcyclecounter(); } }

    const i32x4 rsW2H = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W2H_OFF), 0, 2 * W2X_PLANE_BYTES, 0x00020000);
    const i32x4 rsW2TH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W2TH_OFF), 0, 2 * W2X_PLANE_BYTES, 0x00020000);
    // W1 * 2^8 as fp16 pairs in its two operand forms comes from the weight workspace (planes W1F / W1X, rebuilt with the W2 planes after every
    // optimiser step: nl_optim.hip) - 32 registers per lane would not fit next to four accumulators, and there is no LDS left:
    //   w1f[j][plane]: layer 1's B fragments of column tile 2 w + j, fetched at the end of a tile for the next one (in flight across the barrier)
    //   w1x[ks][plane]: dX's B fragments of this wave's 64 columns (k-steps 2 w, 2 w + 1 of W1X), fetched behind the dgrad loop
    const i32x4 rsW1F = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W1F_OFF), 0, 2 * NL_W * NL_C * 2, 0x00020000);
    const i32x4 rsW1X = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2T + NL_DEC_WS_W1X_OFF), 0, 2 * NL_W * NL_C * 2, 0x00020000);
    const i32x4 rsPar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.params), 0, NL_DEC_PARAMS * 4, 0x00020000);
    const float b3 = a.params[NL_OFF_B3];
    uint4 w1f[2][2];
    auto fetch_w1f = [&](int lane_, int w_) {
        const int so = __builtin_amdgcn_readfirstlane(w_) * 2 * 2 * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w1f[j][pl] = bload4(rsW1F, lane_ * 16, so + (j * 2 + pl) * 1024);
    };

    f32x16 accW1h[2];                                     // this wave's 2 x 32 rows of dW1 (lanes 0-15: the channel) and of db1 (lane 16)
    float aW3[2] = {0.f, 0.f}, aB2[2] = {0.f, 0.f}, aB3 = 0.f, dsMax = 0.f;
    double* const sLoss = reinterpret_cast<double*>(ldsb + D2_LOSS);
    if (threadIdx.x < 2 * DEC_M) sLoss[threadIdx.x] = 0.0;
    if (TRAIN) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW1h[j][r] = 0.f;
    }
    // next tile's inputs (the loss inputs are a two-level chain - sample -> its ray -> the ray's cos / range - spread over two tiles); a thread stages four values of X
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    float pdep = 0.f, pcos = 0.f, pd = 0.f;
    int pray = -1;
    auto ray_of = [&](int tile, int lane_) {
        const int row0 = tile * DEC_M;
        return (tile < ntiles && row0 + lane_ < P) ? a.s_ray[row0 + lane_] : -1;
    };
    auto prefetch = [&](int tile, int lane_, int xi_, int xc_) {      // `pray` holds the ray id of this tile's row (fetched a tile earlier)
        const int row0 = tile * DEC_M;
        xv = make_float4(0.f, 0.f, 0.f, 0.f); pdep = 0.f; pcos = 0.f; pd = 0.f;
        if (tile < ntiles) {
            if (row0 + xi_ < P) xv = *reinterpret_cast<const float4*>(a.X + (size_t)(row0 + xi_) * NL_C + xc_);
            if (pray >= 0) { pdep = a.s_depth[row0 + lane_]; pcos = a.cos_gt[pray]; pd = a.gt_dist[pray]; }
        }
    };
    unsigned xw = 0u;                                     // range watch (nl_common.h nl_range_check)
    auto stage_x = [&](int xi_, int xc_) {
        xw = nl_xw_update(nl_xw_update(xw, xv.x, xv.y), xv.z, xv.w);
        if (TRAIN) { float* d = sXf + opaque(xi_ * LDX + xc_); d[0] = xv.x; d[1] = xv.y; d[2] = xv.z; d[3] = xv.w; }
        unsigned h0, l0, h1, l1;
        split2_pair_f16(sat_f16(xv.x * NL_F16_SX), sat_f16(xv.y * NL_F16_SX), &h0, &l0);
        split2_pair_f16(sat_f16(xv.z * NL_F16_SX), sat_f16(xv.w * NL_F16_SX), &h1, &l1);
        unsigned char* d = ldsb + opaque(D2_XU + xi_ * 32 + 2 * xc_);
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + XG_XP_BYTES) = make_uint2(l0, l1);
    };
    // the first tile's inputs; from then on a tile requests the NEXT tile's inputs at the start of its phase F (two GEMM / conversion phases = several us in
    // front of their use at the tile's end) and the ray ids of the tile after that one: the prefetch registers are dead from the tile's start to phase F,
    // i.e. during the forward GEMM, where the kernel is at the register limit
    int pray_next;
    float cz, cd;
    {
        D2_IDS();
        pray_next = ray_of(blockIdx.x + gridDim.x, lane);
        pray = ray_of(blockIdx.x, lane);
        prefetch(blockIdx.x, lane, xi, xc);
        fetch_w1f(lane, w);
        stage_x(xi, xc);
        cz = pdep * pcos; cd = pd;
    }
    __syncthreads();

    bool prio_alt = blockIdx.x >= gridDim.x / 2;
#ifdef D2_STAGGER                                       // A/B aid: the CU's second workgroup starts D2_STAGGER cycles late (half a tile: anti-phase from the start)
    if (blockIdx.x >= gridDim.x / 2) { const long long t0 = __builtin_readcyclecounter(); while (__builtin_readcyclecounter() - t0 < D2_STAGGER) __builtin_amdgcn_s_sleep(8); }
#endif
    D2_PRIO_VALU();
    int tile_no = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++tile_no) {
        const int row0 = tile * DEC_M;
        prio_alt = ((tile_no + (blockIdx.x >= gridDim.x / 2 ? 1 : 0)) & 1) != 0;
        const int tid = threadIdx.x;                    // (the stamps' `tid == 0`; the phases below re-derive their lane ids)
        DBG_STAMP(0);
        DBG_STAMP(1);
        // ---------------- B: H1 = relu(X W1^T + b1) of both column tiles -> the hi / lo planes (P0, P1) ----------------
        unsigned m1[2];
        uint4 bqh[F2_RING][2][2];
        {
            D2_IDS();
            const unsigned char* xq = ldsb + opaque(D2_XU + l31 * 32 + 16 * lh);
            uint4 xa0[2], xa1[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                xa0[pl] = *reinterpret_cast<const uint4*>(xq + pl * XG_XP_BYTES);
                xa1[pl] = *reinterpret_cast<const uint4*>(xq + pl * XG_XP_BYTES + 32 * 32);
            }
            float b1s[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b1s[j] = bload(rsPar, (32 * (2 * w + j) + l31) * 4, NL_OFF_B1 * 4) * NL_F16_SH;
            gemm_f16_2ct_prefetch(rsW2TH, w, lane, bqh);       // W2 planes of the first k-steps: in flight across the barrier
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c0, c1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
                for (int pa = 1; pa >= 0; --pa)
#pragma unroll
                    for (int pq = 1; pq >= 0; --pq) { c0 = mma16<true>(xa0[pa], w1f[j][pq], c0); c1 = mma16<true>(xa1[pa], w1f[j][pq], c1); }
                if (tile_no == 0 && h1_clips(c0, c1, b1s[j])) xw = nl_xw_mark(xw, NL_SAT_H1);
                m1[j] = store_h1_planes_f16(reinterpret_cast<unsigned short*>(ldsb), 32 * (2 * w + j) + l31, lh, c0, c1, b1s[j]);
            }
        }
        nl_lds_barrier();
        DBG_STAMP(2);
        // ---------------- C: H2 = relu(H1 W2^T + b2), partial row sums of H2 w3 ----------------
        f32x16 h[2][2];
        {
            D2_IDS();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { h[j][0][r] = 0.f; h[j][1][r] = 0.f; }
            float b2c[2], w3c[2];                          // (requested here, used behind the loop)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b2c[j] = bload(rsPar, (32 * (2 * w + j) + l31) * 4, NL_OFF_B2 * 4); w3c[j] = bload(rsPar, (32 * (2 * w + j) + l31) * 4, NL_OFF_W3 * 4);
            }
            D2_PRIO_MFMA();
            gemm_f16_2ct<NP>(rsW2TH, w, lane, ldsb, bqh, h);
            D2_PRIO_VALU();
            DBG_STAMP(3);
            constexpr float S2 = 1.0f / (NL_F16_SH * NL_F16_SW2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { h[j][0][r] = fmaxf(fmaf(h[j][0][r], S2, b2c[j]), 0.f); h[j][1][r] = fmaxf(fmaf(h[j][1][r], S2, b2c[j]), 0.f); }
                const float tot = halfwave_rowsum(h[j][0], h[j][1], w3c[j], l31);
                sS[opaque((2 * w + j) * DEC_M + (l31 >> 4) * 32 + d32_row(l31 & 15, lh))] = tot;
            }
        }
        nl_lds_barrier();
        DBG_STAMP(4);
        // ---------------- D: sdf, loss gradient: every wave computes all 64 rows (lane = row) for itself; wave 0 owns the outputs ----------------
        float inv_sigma = 1.0f;
        uint4 bqm[M2_RING][2][2];
        {
            D2_IDS();
            {
                const int g = row0 + lane;
                float ds = 0.f;
                if (g < P) {
                    float s = sS[lane];                             // fixed summation order over the 8 column tiles (k_decoder's)
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) s += sS[ww * DEC_M + lane];
                    s += b3;
                    bool f, m;
                    nl_loss_masks(cz, cd, ls.tau, ls.max_depth, &f, &m);
                    float q1, q2;
                    // (the loss gradient's uniform factors are products of launch constants: hoisted out of the tile loop they are two more vector registers
                    //  held - here: spilled - for the whole kernel; laundering the first factors recomputes the two products per tile, same order, same bits)
                    NlLossScalars lsd = ls;
                    lsd.fs_weight = launder_f(ls.fs_weight); lsd.sdf_weight = launder_f(ls.sdf_weight);
                    ds = nl_loss_grad(s, cz, cd, f, m, lsd, &q1, &q2);
                    if (w == 0) {
                        a.sdf[g] = s; a.dsdf[g] = ds;
                        sLoss[2 * lane] += (double)q1; sLoss[2 * lane + 1] += (double)q2;       // (this lane's own slots: plain read-modify-write)
                    }
                }
                sdS[lane] = ds;
                if (TRAIN && w == 0) { aB3 += ds; dsMax = fmaxf(dsMax, fabsf(ds)); }
                __builtin_amdgcn_wave_barrier();
            }
            DBG_STAMP(5);
            if (TRAIN) {
                const float mx = wave_max_nonneg(fabsf(sdS[lane]));
                int e = 0;
                if (mx > 0.f) { (void)frexpf(mx, &e); e = e < -100 ? -100 : e; }
                const float sigma = ldexpf(1.0f, 4 - e);
                inv_sigma = ldexpf(1.0f, e - 4);
                // U planes (k_decoder's layout: [plane][(k-step t, lane half hh) chunk of 17 fragments][column c][8 rows], a thread = half a fragment = 4 rows): the 256
                // half fragments of the 16 channels are one per thread - no loop -, and the 17th column (sigma dsdf itself: db1) is written row by row by the wave's
                // lanes, which hold one dsdf each (wave w: rows 16 w .. 16 w + 15)
                {
                    const int c = tid & 15, f = tid >> 4, half = f & 1, hh = (f >> 1) & 1, t = f >> 2;      // f = (t, hh, half): 16 values
                    const int row = 16 * (t & 1) + 4 * hh + 32 * (t >> 1) + 8 * half;
                    const float4 dv = *reinterpret_cast<const float4*>(sdS + opaque(row));
                    const float* xr = sXf + opaque(row * LDX + c);
                    float u[4];
                    u[0] = sat_f16(dv.x * sigma * 16.0f * xr[0]); u[1] = sat_f16(dv.y * sigma * 16.0f * xr[LDX]);
                    u[2] = sat_f16(dv.z * sigma * 16.0f * xr[2 * LDX]); u[3] = sat_f16(dv.w * sigma * 16.0f * xr[3 * LDX]);
                    unsigned uh[2], ul[2];
                    split2_pair_f16(u[0], u[1], &uh[0], &ul[0]); split2_pair_f16(u[2], u[3], &uh[1], &ul[1]);
                    unsigned char* ud = ldsb + opaque(D2_XU + (t * 2 + hh) * D2_U_CHUNK + c * 16 + 8 * half);
                    *reinterpret_cast<uint2*>(ud) = make_uint2(uh[0], uh[1]);
                    *reinterpret_cast<uint2*>(ud + D2_U_PLANE) = make_uint2(ul[0], ul[1]);
                }
                if ((lane >> 4) == w) {                      // lanes 16 w .. 16 w + 15 of wave w: row = lane
                    // row = 16 (t & 1) + 4 hh + 32 (t >> 1) + 8 half + z  <=>  t = 2 (row >> 5) + ((row >> 4) & 1), half = (row >> 3) & 1, hh = (row >> 2) & 1, z = row & 3
                    const int row = lane, t = 2 * (row >> 5) + ((row >> 4) & 1), half = (row >> 3) & 1, hh = (row >> 2) & 1, z = row & 3;
                    const float d = sdS[row] * sigma;
                    const _Float16 dh = (_Float16)d, dl = (_Float16)(d - (float)dh);
                    _Float16* ud = reinterpret_cast<_Float16*>(ldsb + opaque(D2_XU + (t * 2 + hh) * D2_U_CHUNK + NL_C * 16 + 8 * half + 2 * z));
                    ud[0] = dh; ud[D2_U_PLANE / 2] = dl;
                }
            }
            DBG_STAMP(11);
            // ---------------- E: the 0/1 mask tile of H2 -> P0; db2 / dW3 sums; the saved ReLU words ----------------
            gemm_mask_2ct_prefetch(rsW2H, w, lane, bqm);
            DBG_STAMP(12);
            // the 32 dsdf values of this lane's rows (rows D32_RR(r) + 4 lh of both sub-tiles: eight runs of four consecutive floats), read ONCE for both column
            // tiles as eight ds_read_b128 - sixteen ds_read2_b32 per column tile issued next to their use put an LDS latency in front of every row pair
            float dsv[2][16];
            if (TRAIN) {
                const float4* dq = reinterpret_cast<const float4*>(sdS + opaque(4 * lh));
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = dq[8 * sub + 2 * q];               // floats 32 sub + 8 q .. + 3 (+ 4 lh)
                        dsv[sub][4 * q] = v.x; dsv[sub][4 * q + 1] = v.y; dsv[sub][4 * q + 2] = v.z; dsv[sub][4 * q + 3] = v.w;
                    }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = 32 * (2 * w + j) + l31;
                unsigned mw = 0u;
                // dW3 / db2 sums of the tile in four independent chains per column tile (rows of the two sub-tiles, even / odd register): one chain per sum is a
                // 64-deep dependent sequence of FMAs (~8 cycles each with nothing to issue in between); the four partial sums are folded in a fixed order
                float pW[4] = {0.f, 0.f, 0.f, 0.f}, pB[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ds0 = dsv[0][r], ds1 = dsv[1][r];
                    // (laundered: the ReLU bits are pure functions of H2, and the compiler otherwise forms all 64 of them in phase C's epilogue and carries them -
                    //  one register each - next to the 64 values of H2 across phase D)
                    const float hv0 = launder_f(h[j][0][r]), hv1 = launder_f(h[j][1][r]);
                    const unsigned on0 = pos_bit(hv0), on1 = pos_bit(hv1);
                    if (TRAIN) {
                        pW[r & 1] = fmaf(ds0, hv0, pW[r & 1]); pW[2 + (r & 1)] = fmaf(ds1, hv1, pW[2 + (r & 1)]);
                        pB[r & 1] = fmaf(ds0, (float)on0, pB[r & 1]); pB[2 + (r & 1)] = fmaf(ds1, (float)on1, pB[2 + (r & 1)]);     // (ds * 1 + acc / ds * 0 + acc: the adds they replace, bit for bit)
                    }
                    mw |= (on0 << r) | (on1 << (16 + r));
                }
                // (the sums are used at the loop's back edge only: without a use HERE the machine sinker moves the chains - and with them the lifetime of all 64
                //  values of H2, 64 ReLU bits and 64 dsdf reads - to the end of the tile: ~290 spilled registers)
                if (TRAIN) { aW3[j] = launder_f(aW3[j] + ((pW[0] + pW[1]) + (pW[2] + pW[3]))); aB2[j] = launder_f(aB2[j] + ((pB[0] + pB[1]) + (pB[2] + pB[3]))); }
                if (j == 0) { DBG_STAMP(13); } else { DBG_STAMP(14); }
                const bool odd = (col & 1) != 0;
                const unsigned nb = dpp_swap1u(mw);
                const unsigned lo_bits = odd ? nb : mw, hi_bits = odd ? mw : nb;
                unsigned* mb = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(ldsb) + opaque((4 * lh + (odd ? 1 : 0)) * SM_STRIDE + (col & ~1)));
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const unsigned lb = (odd ? lo_bits >> 1 : lo_bits) >> (16 * sub + r), hb = (odd ? hi_bits >> 1 : hi_bits) >> (16 * sub + r);
                        mb[((32 * sub + D32_RR(r)) * SM_STRIDE) / 2] = ((lb & 1u) ? 0x3C00u : 0u) | ((hb & 1u) ? 0x3C000000u : 0u);
                    }
                if (TRAIN) a.relu2_mask[(size_t)tile * DEC_THREADS + (2 * w + j) * 64 + lane] = mw;     // k_decoder's (tile, thread) layout
            }
        }
        nl_lds_barrier();
        DBG_STAMP(6);
        // ---------------- F: dgrad accumulators of both column tiles (registers) ----------------
        f32x16 g[2][2];
        uint4 w1x[2][2];
        {
            D2_IDS();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { g[j][0][r] = 0.f; g[j][1][r] = 0.f; }
            pray = pray_next;
            prefetch(tile + gridDim.x, lane, xi, xc);
            pray_next = ray_of(tile + 2 * gridDim.x, lane);
            D2_PRIO_MFMA();
            // (dX's W1 fragments are requested two k-steps before the dgrad loop ends - behind the loop their L2 latency, ~900 cycles, was exposed on every tile; in
            //  front of it they are 16 more registers through the loop, which spills)
            gemm_mask_2ct(rsW2H, w, lane, ldsb, bqm, g, [&]() {
                const int so = __builtin_amdgcn_readfirstlane(w) * 2 * 2 * 1024;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) w1x[ks][pl] = bload4(rsW1X, lane * 16, so + (ks * 2 + pl) * 1024);
            });
            D2_PRIO_VALU();
            DBG_STAMP(7);
            if (tile_no == 0 && (q_clips(g[0][0], g[0][1], m1[0]) || q_clips(g[1][0], g[1][1], m1[1]))) xw = nl_xw_mark(xw, NL_SAT_Q);
        }
        DBG_STAMP(8);
        // ---------------- H / I: Q = [H1 > 0] x accumulator as an fp16 pair; dW1 / db1 from the lane's own registers (k_decoder's phase H); the wave's
        //                  own 64 columns of Q through its private slice of P1, half a tile at a time, into its partial dX ----------------
        {
            D2_IDS();
            f32x4 cx[2][2];                                // [sub][16-row tile] partial dX of rows 32 sub + 16 mt + 4 lq + r, channel l15
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) cx[sub][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned char* ub = ldsb + opaque(D2_XU + lh * D2_U_CHUNK + (l31 < 17 ? l31 : 16) * 16);
            const unsigned char* qa = ldsb + opaque(D2_P1 + l15 * (SM_STRIDE * 2) + 128 * w + 16 * lq);
            const float sc = inv_sigma * (l31 == 16 ? 1.0f / NL_F16_SG : 1.0f / (NL_F16_SG * 16.0f));
            // column tile j = k-step j of the wave's 64 columns; a half tile (sub) of its Q goes through columns [32 sub, 32 sub + 32) of the wave's slice
            // (hi: rows 0-31, lo: rows 32-63), so the two half tiles of a column tile never wait for each other's reads
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = 32 * (2 * w + j) + l31;
                const bool odd = (col & 1) != 0;
                const unsigned sel = odd ? 0x03020706u : 0x05040100u;
                f32x16 tw;
                if (TRAIN) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) tw[r] = 0.f;
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    unsigned* pq = reinterpret_cast<unsigned*>(ldsb + opaque(D2_P1 + ((4 * lh + (odd ? 1 : 0)) * SM_STRIDE + 64 * w + 32 * sub + (l31 & ~1)) * 2));
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        const int t = 2 * sub + t2;
                        uint4 qh, ql, bh, bl;
                        if (TRAIN) { bh = *reinterpret_cast<const uint4*>(ub + 2 * t * D2_U_CHUNK); bl = *reinterpret_cast<const uint4*>(ub + 2 * t * D2_U_CHUNK + D2_U_PLANE); }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 8 * t2 + 2 * e;
                            const float va = __uint_as_float(__float_as_uint(sat_f16(g[j][sub][r])) & (unsigned)__builtin_amdgcn_sbfe((int)m1[j], 16 * sub + r, 1));
                            const float vb = __uint_as_float(__float_as_uint(sat_f16(g[j][sub][r + 1])) & (unsigned)__builtin_amdgcn_sbfe((int)m1[j], 16 * sub + r + 1, 1));
                            unsigned ph, pl;
                            split2_pair_f16(va, vb, &ph, &pl);
                            (&qh.x)[e] = ph; (&ql.x)[e] = pl;
                            unsigned* d = pq + (D32_RR(r) * SM_STRIDE) / 2;           // hi: rows 0-31 of the region; lo: rows 32-63
                            d[0] = __builtin_amdgcn_perm(dpp_swap1u(ph), ph, sel);
                            d[(32 * SM_STRIDE) / 2] = __builtin_amdgcn_perm(dpp_swap1u(pl), pl, sel);
                        }
                        if (TRAIN) { tw = mma16<true>(ql, bh, tw); tw = mma16<true>(qh, bl, tw); tw = mma16<true>(qh, bh, tw); }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");       // same wave: LDS is in order - this keeps the compiler from moving the reads up
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
                    const f16x8 wh = __builtin_bit_cast(f16x8, w1x[j][0]), wl = __builtin_bit_cast(f16x8, w1x[j][1]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(qa + 16 * mt * (SM_STRIDE * 2) + 64 * sub));
                        const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(qa + (32 + 16 * mt) * (SM_STRIDE * 2) + 64 * sub));
                        cx[sub][mt] = MFMA16_F16(al, wh, cx[sub][mt]); cx[sub][mt] = MFMA16_F16(ah, wl, cx[sub][mt]); cx[sub][mt] = MFMA16_F16(ah, wh, cx[sub][mt]);
                    }
                }
                if (TRAIN) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) accW1h[j][r] = launder_f(fmaf(tw[r], sc, accW1h[j][r]));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            // the wave's partial dX tile -> its slice of P1 (the Q values there have been consumed by this wave's own reads)
            float* px = reinterpret_cast<float*>(ldsb + opaque(D2_P1 + (4 * lq) * (SM_STRIDE * 2) + 128 * w + 4 * l15));
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) px[((32 * sub + 16 * mt + r) * (SM_STRIDE * 2)) / 4] = cx[sub][mt][r];
        }
        DBG_STAMP(9);
        nl_lds_barrier();
        // ---------------- dX = dsdf_i 2^-18 x (the four partial tiles, waves 0..3 in order); X of the next tile -> LDS ----------------
        {
            D2_IDS();
            const unsigned char* pr = ldsb + opaque(D2_P1 + xi * (SM_STRIDE * 2) + 4 * xc);
            float4 s = *reinterpret_cast<const float4*>(pr);
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const float4 v = *reinterpret_cast<const float4*>(pr + 128 * ww);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const float dsc = sdS[xi] * (1.0f / (NL_F16_SG * NL_F16_SW1));
            if (row0 + xi < P) *reinterpret_cast<float4*>(a.dX + (size_t)(row0 + xi) * NL_C + xc) = make_float4(s.x * dsc, s.y * dsc, s.z * dsc, s.w * dsc);
            stage_x(xi, xc);
            cz = pdep * pcos; cd = pd;
            fetch_w1f(lane, w);
        }
        nl_lds_barrier();
        DBG_STAMP(10);
    }

    if constexpr (STAMPS) {
        if (a.dbg && threadIdx.x == 0) {
            long long* rec = a.dbg + 256 + 8 * (long long)blockIdx.x;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            rec[0] = wg_t0; rec[1] = (long long)wall_clock64(); rec[2] = wg_c0; rec[3] = (long long)__builtin_readcyclecounter();
            rec[4] = hw; rec[5] = xcc; rec[6] = tile_no; rec[7] = 0;
        }
    }
    D2_IDS();
    nl_range_check(a.W2T, xw, TRAIN);                     // did an operand of the fp16-pair arithmetic leave its range?  (sticky status word of the weight workspace)
    // ---------------- loss sums ----------------
    if (tid < 64) {
        double lossFs = sLoss[2 * lane], lossSdf = sLoss[2 * lane + 1];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lossFs += __shfl_xor(lossFs, off); lossSdf += __shfl_xor(lossSdf, off); }
        if (tid == 0 && (lossFs != 0.0 || lossSdf != 0.0)) {
            atomicAdd(&a.dcounters[NLD_FS_SQ], lossFs); atomicAdd(&a.dcounters[NLD_SDF_SQ], lossSdf);
        }
    }
    // ---------------- flush weight-gradient partials (slab blockIdx.x: W1, b1, b2, W3, b3 - dW2 is k_decoder_wgrad2*'s) ----------------
    if (TRAIN) {
        float* base = a.partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * (2 * w + j) + D32_RR(r) + 4 * lh;
                if (l31 < NL_C) base[NL_OFF_W1 + k * NL_C + l31] = accW1h[j][r];
                else if (l31 == NL_C) base[NL_OFF_B1 + k] = accW1h[j][r];
            }
            aW3[j] += __shfl_xor(aW3[j], 32); aB2[j] += __shfl_xor(aB2[j], 32);
            if (lh == 0) { const int col = 32 * (2 * w + j) + l31; base[NL_OFF_W3 + col] = aW3[j]; base[NL_OFF_B2 + col] = aB2[j] * a.params[NL_OFF_W3 + col]; }
        }
        if (tid < 64) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { aB3 += __shfl_xor(aB3, off); dsMax = fmaxf(dsMax, __shfl_xor(dsMax, off)); }
            if (tid == 0) {
                base[NL_OFF_B3] = aB3;
                if (dsMax > 0.f) atomicMax(const_cast<unsigned*>(&a.ls->ds_max_bits), __float_as_uint(dsMax));
            }
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
                                                                    const unsigned* __restrict__ relu2_mask, float* __restrict__ partials)
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
            pmk = relu2_mask[(size_t)tile * DEC_THREADS + tid];
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
// dW2 on the bf16 matrix cores, EXACTLY.  dW2[j][k] = sum_i dH2[i][j] H1[i][k] with dH2[i][j] = m(i,j) * dsdf_i * w3_j
// and m the 0/1 ReLU mask of H2, so
//     dW2[j][k] = w3_j * sum_i  m(i,j) * (dsdf_i * H1[i][k]).
// The A operand m is exactly representable in bf16, and the fp32 B operand v = dsdf_i * H1[i][k] is split into three
// bf16 terms v = v_hi + v_mid + v_lo (8 + 8 + 8 significand bits: the split is exact), so every product the matrix
// core forms is exact and the accumulation is fp32 - the same arithmetic class as the fp32 MFMA kernel above (a
// different summation order, nothing else), at 3/16 of its matrix-pipe time (v_mfma_f32_32x32x16_bf16 retires
// 16 k-steps in 32 cycles against 2 in 64).
//
// Layout per 64-sample tile (K = the 64 samples).  The k-slot order inside the tile is chosen so that both operands
// come straight out of their producers' register layouts: slot = 32*sub + 16*lh + r for the sample row
// 32*sub + d32_row(r, lh) (sub = 32-row sub-tile; lh, r = lane half and accumulator register of the 32x32 MFMA that
// produced it).  k_decoder's saved mask word of (column j, lane half lh) then holds slots [16*lh, 16*lh+16) of sub 0
// in its low half and of sub 1 in its high half: an A fragment (8 slots of one j) is one BYTE of a mask word,
// expanded to 8 bf16 by a 256-entry LDS table.  H1 is rebuilt from X (K = 16, fp32 MFMA) in the same register layout,
// so a lane packs its 16 rows of one column into 32 contiguous bytes per plane: sB[plane][column k][slot], row
// stride 144 B (conflict-free ds_read_b128 / ds_write_b128).
// Wave (wj = w>>1, wk = w&1) owns dW2 rows [64 wj, +64) x columns [128 wk, +128): 2 x 4 accumulators; one B
// fragment (ds_read_b128) feeds two MFMAs.
// ---------------------------------------------------------------------------------------------
#define WX_STRIDE 144                                   // bytes per column row of a plane: 64 slots x 2 B + 16 pad
#define WX_PLANE (NL_W * WX_STRIDE)
#define WX_OFF_LUT (3 * WX_PLANE)
#define WX_OFF_MASK (WX_OFF_LUT + 256 * 16)
#define WX_OFF_X (WX_OFF_MASK + 2 * DEC_THREADS * 4)
#define WXX_STRIDE 48                                   // bytes per sample row of an X plane: 16 bf16 + 16 pad (conflict-free ds_read_b128)
#define WXX_PLANE (DEC_M * WXX_STRIDE)
#define WX_OFF_DS (WX_OFF_X + 2 * 3 * WXX_PLANE)
#define WX_TOTAL (WX_OFF_DS + 2 * DEC_M * 4)


template <bool F16>                                     // F16 (wgrad2 mode 2): the fp32 operand as an fp16 PAIR instead of three bf16 terms - see below
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decoder_wgrad2_x(const NlLossScalars* __restrict__ lsp, const float* __restrict__ X,
                                                                      const float* __restrict__ params, const float* __restrict__ dsdf,
                                                                      const unsigned* __restrict__ relu2_mask, float* __restrict__ partials)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[WX_TOTAL];
    unsigned char* sB = smem;
    uint4* sLut = reinterpret_cast<uint4*>(smem + WX_OFF_LUT);
    unsigned* sMask = reinterpret_cast<unsigned*>(smem + WX_OFF_MASK);
    unsigned char* sXp = smem + WX_OFF_X;                // X tile as three bf16 planes [parity][plane][64 rows][48 B]
    float* sdS = reinterpret_cast<float*>(smem + WX_OFF_DS);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int col = 32 * w + l31;                        // producer role: H1 column of this lane
    const int wj = w >> 1, wk = w & 1;                   // consumer role: dW2 block of this wave
    const int P = lsp->P;
    const int ntiles = (P + DEC_M - 1) / DEC_M;
    const float b1c = params[NL_OFF_B1 + col];
    // layer 1 on the bf16 matrix cores, exact products like the 256-deep GEMMs: X and W1 as three bf16 terms each, nine K = 16
    // MFMAs per half tile (288 pipe cycles) instead of eight fp32 ones (512).  B fragments (W1 row of this lane's column, k = 8 lh + e)
    // stay in registers for the whole kernel.
    // F16: the operands as fp16 pairs, like k_decoder<.., 3 / 4>: H1 * 2^4 from the same four layer-1 products in the same order (the H1 the
    // forward pass saw, bit for bit), and v = (dsdf_i * sigma / 16) * (16 H1[i][k]) split in two.  sigma is a power of two that puts the
    // launch's largest |dsdf| in [8, 16) (NlLossScalars.ds_max_bits, raised by k_decoder): with H1 < 4094 no v leaves fp16's range, the
    // samples that carry the gradient sit high in it, and what the low term of a small contribution loses is absolute - 2^-25 against
    // summands of order one.  mask x 2 planes: 64 matrix instructions per tile and wave against 96.
    constexpr int NPL = F16 ? 2 : 3;
    uint4 w1p[3];
    if (F16) { uint4 w1h[2]; l1_w1_fragments_f16(params, col, lh, w1h); w1p[0] = w1h[0]; w1p[1] = w1h[1]; }
    else l1_w1_fragments(params, col, lh, w1p);
    float ds_scale = 1.0f, out_scale = 1.0f;
    if (F16) {
        const float dsmax = __uint_as_float(lsp->ds_max_bits);
        int e = 0;
        if (dsmax > 0.f) { (void)frexpf(dsmax, &e); e = e < -100 ? -100 : e; }      // dsmax = m 2^e, m in [0.5, 1)
        ds_scale = ldexpf(1.0f, 4 - e) * (1.0f / NL_F16_SH);     // dsdf * ds_scale * (16 H1) = v * sigma, sigma = 2^(4 - e)
        out_scale = ldexpf(1.0f, e - 4);
    }
    if (tid < 256) {                                     // byte -> 8 bf16 / fp16 (1.0 where the bit is set)
        constexpr unsigned ONE = F16 ? 0x3C00u : 0x3F80u;
        uint4 e;
        e.x = ((tid & 1) ? ONE : 0u) | ((tid & 2) ? ONE << 16 : 0u);
        e.y = ((tid & 4) ? ONE : 0u) | ((tid & 8) ? ONE << 16 : 0u);
        e.z = ((tid & 16) ? ONE : 0u) | ((tid & 32) ? ONE << 16 : 0u);
        e.w = ((tid & 64) ? ONE : 0u) | ((tid & 128) ? ONE << 16 : 0u);
        sLut[tid] = e;
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a_][t][r] = 0.f;

    const int xe = tid * 2, xi = xe >> 4, xc = xe & 15;
    float2 xv = make_float2(0.f, 0.f); float pds = 0.f; unsigned pmk = 0u;
    auto prefetch = [&](int tile) {
        const int row0 = tile * DEC_M;
        xv = make_float2(0.f, 0.f); pds = 0.f; pmk = 0u;
        if (tile < ntiles) {
            if (row0 + xi < P) xv = *reinterpret_cast<const float2*>(X + (size_t)(row0 + xi) * NL_C + xc);
            if (tid < DEC_M && row0 + tid < P) pds = dsdf[row0 + tid];
            pmk = relu2_mask[(size_t)tile * DEC_THREADS + tid];
        }
    };
    // inputs of a tile -> LDS buffers of parity `pb` (X tile, dsdf and mask words are double-buffered by tile parity)
    auto stage_inputs = [&](int pb) {
        unsigned* sx = reinterpret_cast<unsigned*>(sXp + pb * (3 * WXX_PLANE) + xi * WXX_STRIDE + 2 * xc);
        if (F16) split2_pair_f16(sat_f16(xv.x * NL_F16_SX), sat_f16(xv.y * NL_F16_SX), &sx[0], &sx[WXX_PLANE / 4]);
        else split3_pair(xv.x, xv.y, &sx[0], &sx[WXX_PLANE / 4], &sx[2 * WXX_PLANE / 4]);
        if (tid < DEC_M) sdS[pb * DEC_M + tid] = F16 ? pds * ds_scale : pds;
        sMask[pb * DEC_THREADS + tid] = pmk;
    };
    // producer of one 32-sample half tile: v = dsdf_i * relu(X W1^T + b1)[i][col], split into 3 bf16 planes, k-slot order
    auto produce = [&](int pb, int sub) {
        const unsigned char* sx = sXp + pb * (3 * WXX_PLANE) + opaque((32 * sub + l31) * WXX_STRIDE + 16 * lh);
        const float* ds = sdS + pb * DEC_M + opaque(32 * sub + 4 * lh);
        f32x16 c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = 0.f;
        {
            uint4 xa[3];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) xa[pl] = *reinterpret_cast<const uint4*>(sx + pl * WXX_PLANE);
#pragma unroll
            for (int pa = NPL - 1; pa >= 0; --pa)            // smallest terms first
#pragma unroll
                for (int pq = NPL - 1; pq >= 0; --pq) c0 = mma16<F16>(xa[pa], w1p[pq], c0);
        }
        unsigned hi[8], mid[8], lo[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (F16) {
                constexpr float S1 = NL_F16_SH / (NL_F16_SX * NL_F16_SW1);
                const float h0 = __builtin_amdgcn_fmed3f(fmaf(c0[2 * q], S1, b1c * NL_F16_SH), 0.f, NL_F16_MAX);
                const float h1 = __builtin_amdgcn_fmed3f(fmaf(c0[2 * q + 1], S1, b1c * NL_F16_SH), 0.f, NL_F16_MAX);
                split2_pair_f16(h0 * ds[D32_RR(2 * q)], h1 * ds[D32_RR(2 * q + 1)], &hi[q], &lo[q]);
            } else {
                const float h0 = fmaxf(c0[2 * q] + b1c, 0.f), h1 = fmaxf(c0[2 * q + 1] + b1c, 0.f);
                const float v0 = h0 * ds[D32_RR(2 * q)], v1 = h1 * ds[D32_RR(2 * q + 1)];
                hi[q] = pack_hi16(v0, v1);
                const float r0 = v0 - trunc_bf16(v0), r1 = v1 - trunc_bf16(v1);
                mid[q] = pack_hi16(r0, r1);
                const float s0 = r0 - trunc_bf16(r0), s1 = r1 - trunc_bf16(r1);
                lo[q] = pack_hi16(s0, s1);
            }
        }
        unsigned char* dst = sB + opaque(col * WX_STRIDE + 32 * lh + 64 * sub);
        uint4* d0 = reinterpret_cast<uint4*>(dst);
        d0[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); d0[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
        if (!F16) {
            uint4* d1 = reinterpret_cast<uint4*>(dst + WX_PLANE);
            d1[0] = make_uint4(mid[0], mid[1], mid[2], mid[3]); d1[1] = make_uint4(mid[4], mid[5], mid[6], mid[7]);
        }
        uint4* d2 = reinterpret_cast<uint4*>(dst + (NPL - 1) * WX_PLANE);
        d2[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]); d2[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
    };
    // consumer of one half tile: 2 k-steps of 16 slots x 3 planes x 4 column tiles, each B fragment feeding both row tiles.
    // 6 groups (k-step, plane) of 8 MFMAs; the 4 B fragments of the NEXT group are read while this group's MFMAs run.
    auto consume = [&](int pb, int sub) {
        const unsigned* mk = sMask + pb * DEC_THREADS + l31;
        unsigned mwd[2][2];                              // [row tile jt][producer lane half]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) { mwd[jt][0] = mk[(2 * wj + jt) * 64]; mwd[jt][1] = mk[(2 * wj + jt) * 64 + 32]; }
        const unsigned char* bsrc = sB + opaque((128 * wk + l31) * WX_STRIDE + 16 * lh + 64 * sub);
        uint4 bfr[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bfr[0][kt] = *reinterpret_cast<const uint4*>(bsrc + (F16 ? WX_PLANE : 0) + 32 * kt * WX_STRIDE);
        // (plane order of a k-step: hi, mid, lo with the bf16 terms; lo, hi with the fp16 pair)
        uint4 af[2];
#pragma unroll
        for (int gq = 0; gq < 2 * NPL; ++gq) {
            const int s2 = gq / NPL, p3 = gq % NPL;      // k-step inside the half tile: producer lane half s2
            if (p3 == 0) {
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    const unsigned byte = (mwd[jt][s2] >> (16 * sub + 8 * lh)) & 0xFFu;
                    af[jt] = sLut[byte];
                }
            }
            if (gq + 1 < 2 * NPL) {
                const int sn = (gq + 1) / NPL, pn = F16 ? 1 - (gq + 1) % NPL : (gq + 1) % NPL;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    bfr[(gq + 1) & 1][kt] = *reinterpret_cast<const uint4*>(bsrc + pn * WX_PLANE + 32 * kt * WX_STRIDE + 32 * sn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                acc[0][kt] = mma16<F16>(af[0], bfr[gq & 1][kt], acc[0][kt]); acc[1][kt] = mma16<F16>(af[1], bfr[gq & 1][kt], acc[1][kt]);
            }
        }
    };

    // Pipeline over 32-sample half tiles (the two halves of a tile's k-slots are separate regions of sB): in every barrier
    // interval the workgroup produces the operand planes of the NEXT half tile and runs the MFMAs of the current one, so the
    // split/LDS-store work of one wave runs under the other waves' matrix work.
    prefetch(blockIdx.x);
    stage_inputs(0);
    prefetch(blockIdx.x + gridDim.x);
    __syncthreads();
    if (blockIdx.x < ntiles) produce(0, 0);
    __syncthreads();
    int par = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
        // step A: inputs of the next tile -> the other buffers; planes of this tile's second half; MFMAs of its first half
        stage_inputs(par ^ 1);
        prefetch(tile + 2 * gridDim.x);
        produce(par, 1);
        consume(par, 0);
        nl_lds_barrier();
        // step B: planes of the next tile's first half (its inputs were published by the barrier above); MFMAs of this tile's second half
        if (tile + gridDim.x < ntiles) produce(par ^ 1, 0);
        consume(par, 1);
        nl_lds_barrier();
    }
    float* base = partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 64 * wj + 32 * jt + d32_row(r, lh);
            const float w3j = F16 ? params[NL_OFF_W3 + j] * out_scale : params[NL_OFF_W3 + j];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) base[NL_OFF_W2 + j * NL_W + 128 * wk + 32 * kt + l31] = w3j * acc[jt][kt][r];
        }
}

// ---------------------------------------------------------------------------------------------
// forward-only variant for dense SDF queries (mesh-time get_scores, render_helpers.py:96-153) and
// tests: sdf = decoder(X).
// ---------------------------------------------------------------------------------------------
template <bool XG, int NP = 9>
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decoder_fwd(const float* __restrict__ X, const float* __restrict__ params,
                                                                 const float* __restrict__ W2T, int P, float* __restrict__ sdf)
{
    constexpr bool F16 = XG && NP <= 4;                   // fp16 pairs (gemm modes 4 / 5): two planes of every 16-bit operand
    constexpr int H1_FLOATS = XG ? 3 * X_PLANE_BYTES / 4 : DEC_M * LDH;
    constexpr int XS_FLOATS = XG ? 3 * XG_XP_BYTES / 4 : DEC_M * LDX;      // bf16 mode: the X tile as three bf16 planes [64][32 B]
    __shared__ __attribute__((aligned(16))) float lds[H1_FLOATS + XS_FLOATS + 8 * DEC_M];
    float* sH1 = lds; float* sX = lds + H1_FLOATS; float* sS = sX + XS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int col = 32 * w + l31;
    const float* W1 = params + NL_OFF_W1;
    const i32x4 rsW2T = make_w_rsrc(W2T);
    const i32x4 rsW2TX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2T + NL_DEC_WS_W2TX_OFF), 0, 3 * W2X_PLANE_BYTES, 0x00020000);
    const i32x4 rsW2TH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2T + NL_DEC_WS_W2TH_OFF), 0, 2 * W2X_PLANE_BYTES, 0x00020000);
    const float b1c = params[NL_OFF_B1 + col], b2c = params[NL_OFF_B2 + col], w3c = params[NL_OFF_W3 + col], b3 = params[NL_OFF_B3];
    float w1r[NL_C / 2];
    uint4 w1p[3], w1h[2];
    if (F16) l1_w1_fragments_f16(params, col, lh, w1h);
    else if (XG) l1_w1_fragments(params, col, lh, w1p);
    else {
#pragma unroll
        for (int kk = 0; kk < NL_C / 2; ++kk) w1r[kk] = W1[col * NL_C + 2 * kk + lh];
    }
    const int ntiles = (P + DEC_M - 1) / DEC_M;
    unsigned xw = 0u;                                     // F16: range watch (nl_common.h nl_range_check)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * DEC_M;
        {
            const int e = tid * 2, i = e >> 4, c = e & 15;
            float2 v = make_float2(0.f, 0.f);
            if (row0 + i < P) v = *reinterpret_cast<const float2*>(X + (size_t)(row0 + i) * NL_C + c);
            if (F16) xw = nl_xw_update(xw, v.x, v.y);
            if (F16) {
                unsigned* d = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(sX) + i * 32 + 2 * c);
                split2_pair_f16(sat_f16(v.x * NL_F16_SX), sat_f16(v.y * NL_F16_SX), &d[0], &d[XG_XP_BYTES / 4]);
            } else if (XG) {
                unsigned* d = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(sX) + i * 32 + 2 * c);
                split3_pair(v.x, v.y, &d[0], &d[XG_XP_BYTES / 4], &d[2 * XG_XP_BYTES / 4]);
            } else { sX[i * LDX + c] = v.x; sX[i * LDX + c + 1] = v.y; }
        }
        __syncthreads();
        uint4 bq9[X9_RING][3];
        uint4 bqh[F16_RING][2];
        {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
            if (F16) {                                    // the same four products in the same order as k_decoder's phase B
                const unsigned char* xq = reinterpret_cast<const unsigned char*>(sX) + opaque(l31 * 32 + 16 * lh);
                uint4 xa0[2], xa1[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    xa0[pl] = *reinterpret_cast<const uint4*>(xq + pl * XG_XP_BYTES);
                    xa1[pl] = *reinterpret_cast<const uint4*>(xq + pl * XG_XP_BYTES + 32 * 32);
                }
#pragma unroll
                for (int pa = 1; pa >= 0; --pa)
#pragma unroll
                    for (int pq = 1; pq >= 0; --pq) { c0 = mma16<true>(xa0[pa], w1h[pq], c0); c1 = mma16<true>(xa1[pa], w1h[pq], c1); }
            } else if (XG) {                              // the same nine products in the same order as k_decoder's phase B
                const unsigned char* xq = reinterpret_cast<const unsigned char*>(sX) + opaque(l31 * 32 + 16 * lh);
                bf16x8 xa0[3], xa1[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    xa0[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xq + pl * XG_XP_BYTES));
                    xa1[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xq + pl * XG_XP_BYTES + 32 * 32));
                }
#pragma unroll
                for (int pa = 2; pa >= 0; --pa)
#pragma unroll
                    for (int pq = 2; pq >= 0; --pq) {
                        const bf16x8 wf = __builtin_bit_cast(bf16x8, w1p[pq]);
                        c0 = MFMA_BF16(xa0[pa], wf, c0); c1 = MFMA_BF16(xa1[pa], wf, c1);
                    }
            } else {
                const float* xb = sX + opaque(l31 * LDX + lh);
#pragma unroll
                for (int kk = 0; kk < NL_C / 2; ++kk) { c0 = MFMA32(xb[2 * kk], w1r[kk], c0); c1 = MFMA32(xb[32 * LDX + 2 * kk], w1r[kk], c1); }
            }
            if (F16) {
                gemm_f16_prefetch(rsW2TH, w, lane, bqh);
                if (tile == (int)blockIdx.x && h1_clips(c0, c1, b1c * NL_F16_SH)) xw = nl_xw_mark(xw, NL_SAT_H1);
                store_h1_planes_f16(reinterpret_cast<unsigned short*>(lds), col, lh, c0, c1, b1c * NL_F16_SH);
            } else if (XG) {
                gemm_x9_prefetch(rsW2TX, w, lane, bq9);
                store_h1_planes(reinterpret_cast<unsigned short*>(lds), col, lh, c0, c1, b1c);
            } else {
                float* hb = sH1 + opaque(4 * lh * LDH + col);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hb[D32_RR(r) * LDH] = fmaxf(c0[r] + b1c, 0.f); hb[(32 + D32_RR(r)) * LDH] = fmaxf(c1[r] + b1c, 0.f);
                }
            }
        }
        __syncthreads();
        {   // identical arithmetic (and summation order) to k_decoder's phase C/D: forward-only sdf == fused-kernel sdf
            f32x16 h0, h1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
            if constexpr (F16) gemm_f16<true, NP>(rsW2TH, w, lane, reinterpret_cast<const unsigned char*>(lds), bqh, h0, h1);
            else if (XG) gemm_x9<true, F16 ? 9 : NP>(rsW2TX, w, lane, reinterpret_cast<const unsigned char*>(lds), bq9, h0, h1);
            else    gemm256(rsW2T, (lh * NL_W + col) * 4, sH1 + l31 * LDH + lh, sH1 + (32 + l31) * LDH + lh, h0, h1);
            constexpr float S2 = 1.0f / (NL_F16_SH * NL_F16_SW2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h0[r] = fmaxf(F16 ? fmaf(h0[r], S2, b2c) : h0[r] + b2c, 0.f); h1[r] = fmaxf(F16 ? fmaf(h1[r], S2, b2c) : h1[r] + b2c, 0.f);
            }
            const float tot = halfwave_rowsum(h0, h1, w3c, l31);
            sS[w * DEC_M + (l31 >> 4) * 32 + d32_row(l31 & 15, lh)] = tot;
        }
        __syncthreads();
        if (tid < DEC_M && row0 + tid < P) {
            float sv = sS[tid];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) sv += sS[ww * DEC_M + tid];
            sdf[row0 + tid] = sv + b3;
        }
        // (sS is rewritten only after two more barriers; sX / sH1 after one)
    }
    if constexpr (F16) nl_range_check(W2T, xw, false);
}

// sum per-workgroup partial slabs: out[i] = sum_b partials[b][i].  HBM-bound (nslabs x n floats in): 64 columns per block,
// 4 slab groups per column, 4 independent accumulators per thread (16 loads in flight), fixed combination order.
// Columns [lo2, hi2) are summed over the first nslabs2 slabs only (the dW2 region when the fused decoder kernel ran on more - smaller -
// workgroups than the dW2 kernel: decoder_slab_plan below); a workgroup's 64 columns lie on one side of a region bound or take the
// branch per lane.
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ partials, int nslabs_all, int n, float* __restrict__ out,
                                                         int lo2, int hi2, int nslabs2)
{
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    const int nslabs = (i >= lo2 && i < hi2) ? nslabs2 : nslabs_all;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        const float* p = partials + i;
        int b = g;
        for (; b + 12 < nslabs; b += 16) {
            s0 += p[(size_t)b * n]; s1 += p[(size_t)(b + 4) * n]; s2 += p[(size_t)(b + 8) * n]; s3 += p[(size_t)(b + 12) * n];
        }
        for (; b < nslabs; b += 4) s0 += p[(size_t)b * n];
    }
    red[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < n) out[i] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
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

// process-global A/B / profiling state (include/nerfloam_hip_debug.h): relaxed atomics - a setter racing a launch is not a data race
static std::atomic<long long*> g_dec_dbg{nullptr};
static std::atomic<int> g_gemm_mode{4};             // 0: fp32 MFMA GEMMs; bf16 MFMA on the exact-product formulations (gemm_x9 / gemm_mask_x) with
                                         // 1: all nine forward products (exact), 3: eight (without lo x lo), 2: six; fp16 pairs (gemm_f16) with
                                         // 4: three of the four forward products (THE DEFAULT), 5: all four
static std::atomic<int> g_dec_layout{0};            // fp16-pair fused decoder kernel: 0 = by slab count (two 4-wave workgroups per CU - k_decoder2 - when the caller provides
                                         // >= 2 slabs per CU, nl_decoder_grid_hint), 1 = one 8-wave workgroup per CU (k_decoder), 2 = k_decoder2 whatever the slab count
static std::atomic<int> g_wgrad2_mode{2};           // 0: fp32 MFMA (k_decoder_wgrad2), 1: exact 0/1-mask x 3-term bf16 split, 2: 0/1-mask x fp16 pair (k_decoder_wgrad2_x; the default)

extern "C" {

/* profiling aid: device buffer of 256 int64 receiving per-phase s_memtime stamps (NULL disables) */
int nl_decoder_set_debug_buffer(void* dbg) { g_dec_dbg = (long long*)dbg; return NL_OK; }

/* dW2 kernel: 0 = fp32 matrix cores, 1 = bf16 matrix cores on the exact {0,1}-mask x 3-term-split formulation (default).
 * Same arithmetic class (exact products, fp32 accumulation); selectable for A/B measurements and cross-checks. */
int nl_decoder_set_wgrad2_mode(int mode) { if (mode < 0 || mode > 2) return NL_ERR_INVALID_ARG; g_wgrad2_mode = mode; return NL_OK; }
int nl_decoder_get_wgrad2_mode(void) { return g_wgrad2_mode; }
/* the two 256-deep GEMMs of the fused decoder kernels (forward H1 W2^T, dgrad dH2 W2): 0 = fp32 matrix cores; 1, 2, 3 = bf16
 * matrix cores on exact-product formulations: forward = both operands split into three bf16 terms, dgrad = {0,1} ReLU mask x
 * three-term split of w3_j W2[j][k], fp32 accumulation in both.  1 = all nine forward products (exact); 3 = eight, without lo x lo
 * (below 2^-30 of a product: the default); 2 = six (the dropped ones are below 2^-24 of a product: opt-in).  Process-wide
 * DEFAULTS: the *_m entry points and NlIterDesc.kernel_modes take the selection per call. */
int nl_decoder_set_gemm_mode(int mode) { if (mode < 0 || mode > 5) return NL_ERR_INVALID_ARG; g_gemm_mode = mode; return NL_OK; }
int nl_decoder_get_gemm_mode(void) { return g_gemm_mode; }
/* workgroup layout of the fp16-pair fused decoder kernel (gemm modes 4 / 5): 0 = by slab count, 1 = one 8-wave workgroup per CU, 2 = two 4-wave workgroups per CU */
int nl_decoder_set_layout(int layout) { if (layout < 0 || layout > 2) return NL_ERR_INVALID_ARG; g_dec_layout = layout; return NL_OK; }
int nl_decoder_get_layout(void) { return g_dec_layout; }
/* NL_KERNEL_LAYOUT of an iteration over n_rays rays when neither the caller nor the process default chose one (csrc/nl_common.h: the launch-shape table):
 * 1 up to NL_RAYS_DECODER_SPLIT rays - there the iteration is a latency chain and the 8-wave workgroup finishes a tile sooner -, else 0 (by slab count) */
int nl_decoder_layout_for(int n_rays) { return g_dec_layout.load(std::memory_order_relaxed) == 0 && n_rays <= NL_RAYS_DECODER_SPLIT ? 1 : 0; }

}  // extern "C"

struct DecModes { int gemm, wgrad2, layout; };
// kernel_modes = NL_KERNEL_MODES(gemm_mode, wgrad2_mode) | NL_KERNEL_LAYOUT(layout) of include/nerfloam_hip.h; a zero field = the process default
static bool resolve_modes(int kernel_modes, DecModes* m)
{
    const int g = (kernel_modes & 0xFF) - 1, w = ((kernel_modes >> 8) & 0xFF) - 1, l = (kernel_modes >> 16) & 3;
    if (g > 5 || w > 2 || l > 2 || (kernel_modes >> 18) != 0) return false;
    m->gemm = g < 0 ? g_gemm_mode.load(std::memory_order_relaxed) : g;
    m->wgrad2 = w < 0 ? g_wgrad2_mode.load(std::memory_order_relaxed) : w;
    m->layout = l == 0 ? g_dec_layout.load(std::memory_order_relaxed) : l;
    return true;
}

// compute units of the current device (cached per device id: hipGetDeviceProperties costs ~100 us)
static int cu_count()
{
    static std::atomic<int> cached[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    const int slot = dev & 15;
    int n = cached[slot].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cached[slot].store(n, std::memory_order_relaxed);
    return n;
}

// Who writes which slab (`nslabs` = the slabs the caller's `partials` holds, also the upper bound of every persistent grid):
//   * the 512-thread persistent kernels (k_decoder, k_decoder_wgrad2*, k_decoder_fwd) fit one workgroup per CU: grid = min(nslabs, CUs);
//   * k_decoder2 (256 threads, two workgroups per CU) takes all nslabs; it is chosen for the fp16-pair arithmetic when the caller
//     provides at least two slabs per CU (nl_decoder_grid_hint's count) or asks for it (layout 2).
// nl_decoder_fwd_bwd_m, nl_decoder_wgrad2_m and nl_decoder_reduce_m derive the same plan from the same two arguments.
struct SlabPlan { bool split; int grid_fused, grid512; };
static SlabPlan decoder_slab_plan(const DecModes& km, int nslabs)
{
    const int cus = cu_count();
    SlabPlan p;
    p.grid512 = nslabs < cus ? nslabs : cus;
    p.split = (km.gemm == 4 || km.gemm == 5) && km.layout != 1 && (km.layout == 2 || nslabs >= 2 * cus);
    p.grid_fused = p.split ? nslabs : p.grid512;
    return p;
}

extern "C" {

/* slabs to provide in `partials` (and the grid of the fused decoder kernel): two per compute unit of the current device */
int nl_decoder_grid_hint(void) { return 2 * cu_count(); }

int nl_decoder_fwd_bwd_m(const void* loss_scalars, const float* X, const float* params, const float* W2T,
                         const int* s_ray, const float* s_depth, const float* cos_gt, const float* gt_dist,
                         float* sdf, float* dsdf, float* dX, float* partials, unsigned* relu2_mask, int nslabs, int train_decoder,
                         int* counters, int kernel_modes, void* stream)
{
    DecModes km;
    if (!resolve_modes(kernel_modes, &km)) return NL_ERR_INVALID_ARG;
    if (!loss_scalars || !X || !params || !W2T || !s_ray || !s_depth || !cos_gt || !gt_dist || !sdf || !dsdf || !dX || !counters)
        return NL_ERR_INVALID_ARG;
    if (nslabs <= 0 || (train_decoder && (!partials || !relu2_mask))) return NL_ERR_INVALID_ARG;
    DecArgs a;
    a.ls = (const NlLossScalars*)loss_scalars; a.X = X; a.params = params; a.W2T = W2T; a.s_ray = s_ray; a.s_depth = s_depth;
    a.cos_gt = cos_gt; a.gt_dist = gt_dist; a.sdf = sdf; a.dsdf = dsdf; a.dX = dX; a.partials = partials;
    a.relu2_mask = relu2_mask;
    a.dcounters = (double*)(counters + NL_CNT_INTS);
    a.dbg = g_dec_dbg;
    const SlabPlan plan = decoder_slab_plan(km, nslabs);
    if (plan.split) {
        const dim3 g2(plan.grid_fused), b2(D2_THREADS);
        if (a.dbg) {
            if (train_decoder) hipLaunchKernelGGL((k_decoder2<true, 3, true>), g2, b2, 0, (hipStream_t)stream, a);
            else               hipLaunchKernelGGL((k_decoder2<false, 3, true>), g2, b2, 0, (hipStream_t)stream, a);
        } else if (km.gemm == 4) {
            if (train_decoder) hipLaunchKernelGGL((k_decoder2<true, 3>), g2, b2, 0, (hipStream_t)stream, a);
            else               hipLaunchKernelGGL((k_decoder2<false, 3>), g2, b2, 0, (hipStream_t)stream, a);
        } else {
            if (train_decoder) hipLaunchKernelGGL((k_decoder2<true, 4>), g2, b2, 0, (hipStream_t)stream, a);
            else               hipLaunchKernelGGL((k_decoder2<false, 4>), g2, b2, 0, (hipStream_t)stream, a);
        }
        NL_LAUNCH_CHECK();
        return NL_OK;
    }
    const dim3 g(plan.grid512), b(DEC_THREADS);
    if (a.dbg && (km.gemm == 4 || km.gemm == 3)) {           // phase probe: the stamped instantiations of the default and of the exact-product arithmetic
        if (km.gemm == 4) {
            if (train_decoder) hipLaunchKernelGGL((k_decoder<true, true, 3, true>), g, b, 0, (hipStream_t)stream, a);
            else               hipLaunchKernelGGL((k_decoder<false, true, 3, true>), g, b, 0, (hipStream_t)stream, a);
        } else {
            if (train_decoder) hipLaunchKernelGGL((k_decoder<true, true, 8, true>), g, b, 0, (hipStream_t)stream, a);
            else               hipLaunchKernelGGL((k_decoder<false, true, 8, true>), g, b, 0, (hipStream_t)stream, a);
        }
    } else if (km.gemm == 4) {
        if (train_decoder) hipLaunchKernelGGL((k_decoder<true, true, 3>), g, b, 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL((k_decoder<false, true, 3>), g, b, 0, (hipStream_t)stream, a);
    } else if (km.gemm == 5) {
        if (train_decoder) hipLaunchKernelGGL((k_decoder<true, true, 4>), g, b, 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL((k_decoder<false, true, 4>), g, b, 0, (hipStream_t)stream, a);
    } else if (km.gemm == 3) {
        if (train_decoder) hipLaunchKernelGGL((k_decoder<true, true, 8>), g, b, 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL((k_decoder<false, true, 8>), g, b, 0, (hipStream_t)stream, a);
    } else if (km.gemm == 2) {
        if (train_decoder) hipLaunchKernelGGL((k_decoder<true, true, 6>), g, b, 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL((k_decoder<false, true, 6>), g, b, 0, (hipStream_t)stream, a);
    } else if (km.gemm == 1) {
        if (train_decoder) hipLaunchKernelGGL((k_decoder<true, true>), g, b, 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL((k_decoder<false, true>), g, b, 0, (hipStream_t)stream, a);
    } else {
        if (train_decoder) hipLaunchKernelGGL((k_decoder<true, false>), g, b, 0, (hipStream_t)stream, a);
        else               hipLaunchKernelGGL((k_decoder<false, false>), g, b, 0, (hipStream_t)stream, a);
    }
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_decoder_fwd_bwd(const void* loss_scalars, const float* X, const float* params, const float* W2T,
                       const int* s_ray, const float* s_depth, const float* cos_gt, const float* gt_dist,
                       float* sdf, float* dsdf, float* dX, float* partials, unsigned* relu2_mask, int nslabs, int train_decoder,
                       int* counters, void* stream)
{
    return nl_decoder_fwd_bwd_m(loss_scalars, X, params, W2T, s_ray, s_depth, cos_gt, gt_dist, sdf, dsdf, dX, partials, relu2_mask, nslabs,
                                train_decoder, counters, 0, stream);
}

// dW2 slab of the decoder weight gradient (second persistent kernel; needs nl_decoder_fwd_bwd's dsdf + relu2_mask)
int nl_decoder_wgrad2_m(const void* loss_scalars, const float* X, const float* params, const float* dsdf, const unsigned* relu2_mask,
                        float* partials, int nslabs, int kernel_modes, void* stream)
{
    DecModes km;
    if (!resolve_modes(kernel_modes, &km)) return NL_ERR_INVALID_ARG;
    if (!loss_scalars || !X || !params || !dsdf || !relu2_mask || !partials || nslabs <= 0) return NL_ERR_INVALID_ARG;
    nslabs = decoder_slab_plan(km, nslabs).grid512;
    if (km.wgrad2 == 2)
        hipLaunchKernelGGL(k_decoder_wgrad2_x<true>, dim3(nslabs), dim3(DEC_THREADS), 0, (hipStream_t)stream, (const NlLossScalars*)loss_scalars, X,
                           params, dsdf, relu2_mask, partials);
    else if (km.wgrad2 == 1)
        hipLaunchKernelGGL(k_decoder_wgrad2_x<false>, dim3(nslabs), dim3(DEC_THREADS), 0, (hipStream_t)stream, (const NlLossScalars*)loss_scalars, X,
                           params, dsdf, relu2_mask, partials);
    else
        hipLaunchKernelGGL(k_decoder_wgrad2, dim3(nslabs), dim3(DEC_THREADS), 0, (hipStream_t)stream, (const NlLossScalars*)loss_scalars, X,
                           params, dsdf, relu2_mask, partials);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_decoder_wgrad2(const void* loss_scalars, const float* X, const float* params, const float* dsdf, const unsigned* relu2_mask,
                      float* partials, int nslabs, void* stream)
{
    return nl_decoder_wgrad2_m(loss_scalars, X, params, dsdf, relu2_mask, partials, nslabs, 0, stream);
}

int nl_decoder_forward_m(const float* X, const float* params, const float* W2T, int P, float* sdf, int nblocks, int kernel_modes, void* stream)
{
    DecModes km;
    if (!resolve_modes(kernel_modes, &km)) return NL_ERR_INVALID_ARG;
    if (!X || !params || !W2T || !sdf || P < 0 || nblocks <= 0) return NL_ERR_INVALID_ARG;
    if (P == 0) return NL_OK;
    nblocks = decoder_slab_plan(km, nblocks).grid512;
    if (km.gemm == 4) hipLaunchKernelGGL((k_decoder_fwd<true, 3>), dim3(nblocks), dim3(DEC_THREADS), 0, (hipStream_t)stream, X, params, W2T, P, sdf);
    else if (km.gemm == 5) hipLaunchKernelGGL((k_decoder_fwd<true, 4>), dim3(nblocks), dim3(DEC_THREADS), 0, (hipStream_t)stream, X, params, W2T, P, sdf);
    else if (km.gemm == 3) hipLaunchKernelGGL((k_decoder_fwd<true, 8>), dim3(nblocks), dim3(DEC_THREADS), 0, (hipStream_t)stream, X, params, W2T, P, sdf);
    else if (km.gemm == 2) hipLaunchKernelGGL((k_decoder_fwd<true, 6>), dim3(nblocks), dim3(DEC_THREADS), 0, (hipStream_t)stream, X, params, W2T, P, sdf);
    else if (km.gemm == 1) hipLaunchKernelGGL(k_decoder_fwd<true>, dim3(nblocks), dim3(DEC_THREADS), 0, (hipStream_t)stream, X, params, W2T, P, sdf);
    else                  hipLaunchKernelGGL(k_decoder_fwd<false>, dim3(nblocks), dim3(DEC_THREADS), 0, (hipStream_t)stream, X, params, W2T, P, sdf);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_decoder_forward(const float* X, const float* params, const float* W2T, int P, float* sdf, int nblocks, void* stream)
{
    return nl_decoder_forward_m(X, params, W2T, P, sdf, nblocks, 0, stream);
}

int nl_reduce_partials(const float* partials, int nslabs, int n, float* out, void* stream)
{
    if (!partials || !out || nslabs <= 0 || n <= 0) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nl_div_up(n, 64)), dim3(256), 0, (hipStream_t)stream, partials, nslabs, n, out, 0, 0, 0);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* sum of the per-workgroup weight-gradient slabs of nl_decoder_fwd_bwd + nl_decoder_wgrad2 into the decoder gradient */
int nl_decoder_reduce_m(const float* partials, int nslabs, const float* params, float* grad_out, int kernel_modes, void* stream)
{
    DecModes km;
    if (!resolve_modes(kernel_modes, &km)) return NL_ERR_INVALID_ARG;
    if (!partials || !params || !grad_out || nslabs <= 0) return NL_ERR_INVALID_ARG;
    const SlabPlan plan = decoder_slab_plan(km, nslabs);       // dW2's columns come from the dW2 kernel's slabs, the rest from the fused kernel's
    hipLaunchKernelGGL(k_reduce_partials, dim3(nl_div_up(NL_DEC_PARAMS, 64)), dim3(256), 0, (hipStream_t)stream, partials, plan.grid_fused, NL_DEC_PARAMS,
                       grad_out, NL_OFF_W2, NL_OFF_B2, plan.grid512);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_decoder_reduce(const float* partials, int nslabs, const float* params, float* grad_out, void* stream)
{
    return nl_decoder_reduce_m(partials, nslabs, params, grad_out, 0, stream);
}

int nl_mfma_selftest(const float* A32, const float* B32, float* D32, const float* A16, const float* B16, float* D16, void* stream)
{
    hipLaunchKernelGGL(k_mfma_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, A32, B32, D32, A16, B16, D16);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

}  // extern "C"
