// nl_decoder_chain.hip -- the SDF decoder as a REGISTER-CHAINED kernel: the second kernel family of the decoder (gemm mode 3).
//
// Same arithmetic contract as nl_decoder.hip (fp32 values, fp32 accumulation, the 256-deep contractions on the bf16 matrix cores
// as exact-product splits; reference: src/variations/lidar.py:109-131, src/criterion.py:59-100, autograd), different dataflow:
//
//   * nl_decoder.hip keeps the activations of a 64-sample tile in LDS, gives every wave 32 output COLUMNS, and synchronises the
//     eight waves of a workgroup five times per tile; between the barriers the matrix pipes idle while the waves split H1 into
//     operand planes, store 2-byte elements, reduce row sums with 31 shuffles, ...: 47.6 k cycles per tile against 30.7 k of MFMA.
//   * here a wave owns its SAMPLES end to end (64 per pass: two 32-sample sub-tiles).  All GEMMs are issued TRANSPOSED
//     (D^T = B^T A^T: the weights are the A operand, the activations the B operand), so an accumulator tile holds, per lane, one
//     sample and 16 hidden units - and those 16 registers ARE two B-operand fragments of the next layer's MFMA (nl_chain_slot in
//     nl_device_math.h: the weight planes are stored in that slot order by k_prepare_w2a).  H1, H2, the ReLU masks and dH2 never
//     leave the register file; the row sum of the output layer is 16 in-lane FMAs and one shuffle; dL/dsdf is a lane constant.
//     The only activation round trip through LDS is the 32x32 dH1 tile a wave transposes for itself to contract over samples
//     (dX, dW1: K = 16 layers on the fp32 matrix cores, 16x16x4).
//   * one wave per SIMD (256-thread workgroups, 512 registers per lane: three operand planes of H1 are 192 of them).
//   * THE WEIGHT STREAM is what bounds this dataflow (scripts/micro/chain_stream.hip, profiles/experiments): a wave that streams its
//     own A fragments occupies the texture-addresser path for 16 cycles per 1 KB fragment, four waves per CU load every fragment,
//     and at 32 samples per wave a fragment feeds only 9 (forward) or 3 (dgrad) MFMAs.  So the four waves of a workgroup run in
//     lock step and SHARE the planes through LDS: the workgroup walks 24 weight stages per pass (8 output tiles of the forward
//     GEMM for sub-tile a, the same 8 for sub-tile b, 8 tiles of the dgrad over BOTH sub-tiles = 6 MFMAs per fragment); a stage
//     is 48 fragments = 48 KB, double-buffered; every wave loads a quarter of the NEXT stage into registers and writes it to LDS
//     (one fragment per k-step), one barrier per stage.  Memory instructions are interleaved ONE PER MFMA (sched_group_barrier):
//     issued in groups they hold up the issue of the next MFMA (micro-benchmark: 39.5 -> 35.4 cycles per MFMA, floor 34.3).
//
// Weight gradients: dW1 / db1 accumulate per wave over all its tiles; between tiles they are parked in L2 (the W2 region of the
// workgroup's own slab, which the dW2 kernel only writes afterwards) - ten registers per stage instead of eighty for the whole
// kernel - and leave once per workgroup.  dW2 stays in its own kernel (k_decoder_wgrad2_x, "natural" mask format below), which also
// accumulates g[n] = sum_i m2(i,n) dsdf_i, and two identities give the rest without keeping H2:
//     db2[n] = w3_n g[n],    dW3[n] = sum_i dsdf_i h2[i][n] = sum_k W2[n][k] G[n][k] + b2[n] g[n],   G = dW2 / w3 (raw accumulators)
// (h2 = m2 (H1 W2^T + b2)); nl_decoder_reduce applies them while summing the per-workgroup slabs.
#include "nl_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#define CH_THREADS 256
#define CH_W1_STRIDE 20                                 // floats per W1 row in LDS: conflict-free ds_read_b128 of half a row per lane
#define CH_T_STRIDE 36                                  // floats per sample row of a wave's dH1 transposition tile
#define CH_ACC_FLOATS (16 * 64 * 4 + 16 * 64)            // per wave: dW1 tiles [kt][ks][lane][4] + db1 partials [kt][ks][lane]
#define CH_PLANE_BYTES (NL_W * NL_W * 2)
#define CH_WS_W2A_OFF (NL_W * NL_W + 6 * NL_W * NL_W / 2)     // floats into the decoder weight workspace (nl_optim.hip); W2XA follows W2A
#define CH_STAGE_BYTES (16 * 1024)                      // one plane's share of a weight stage: 16 k-steps x 1 KB
// scheduling classes of __builtin_amdgcn_sched_group_barrier
#define SG_MFMA 0x008
#define SG_VMEM_RD 0x020
#define SG_DS_RD 0x100
#define SG_DS_WR 0x200

struct ChainArgs {
    const NlLossScalars* ls;    // NULL: forward only over P samples
    int P;
    const float* X; const float* params; const float* ws;
    const int* s_ray; const float* s_depth; const float* cos_gt; const float* gt_dist;
    float* sdf; float* dsdf; float* dX;
    float* partials;            // [gridDim.x][NL_DEC_PARAMS]: W1, b1, b3 regions (train); the W2 region is this kernel's scratch
    unsigned* relu2_nat;        // [ceil(P/32)][256]: bit b of word (tile, unit) = ReLU of H2[32 tile + b][unit] (train)
    double* dcounters;
    long long* dbg;             // optional [16 passes][16] shader-clock stamps of workgroup 0 / wave 0 (profiling aid)
};
#define CH_STAMP(slot)                                                                          \
    do { if (a.dbg && blockIdx.x == 0 && tid == 0 && pass_no < 16) a.dbg[pass_no * 16 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ uint4 ch_bload4(rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ unsigned ch_pack_hi16(float a, float b)
{
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);      // (lo = bf16 of a, hi = bf16 of b), truncating
}
__device__ __forceinline__ float ch_trunc(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
// Long-lived prefetched values are pinned to the accumulation half of the register file: a value whose uses all want an AGPR is
// loaded there directly and stays there; left to itself the allocator spills such values to scratch right after the load - i.e. it
// waits for the load (HBM latency) on the spot, which is exactly what a prefetch must not do.
__device__ __forceinline__ float ch_from_acc(float x) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(x)); return v; }
__device__ __forceinline__ void ch_keep_in_acc(float x) { asm volatile("" ::"a"(x)); }
// MODE 0: forward only (sdf).  1: forward + loss gradient + dgrad + dX (frozen decoder: tracking, mapping after freeze_frame).
// 2: 1 + decoder weight gradients (dW1, db1, db3 here; the ReLU words of H2 for the dW2 kernel).
// NP: partial products of the forward GEMM (9 = exact, 6 = without lo x lo, lo x mid, mid x lo: gemm mode 2's arithmetic)
template <int MODE, int NP>
__global__ __launch_bounds__(CH_THREADS, 1) void k_decoder_chain(ChainArgs a)
{
    __shared__ __attribute__((aligned(16))) uint4 sRing[2][48][64];       // two weight stages: fragment f = 3 s + p of the stage, lane-major
    __shared__ __attribute__((aligned(16))) float sW1[NL_W * CH_W1_STRIDE];
    __shared__ __attribute__((aligned(16))) float sTab[3 * 256];          // b1 | b2 | w3 in accumulator order [tile][lh][r]
    __shared__ __attribute__((aligned(16))) uint4 sLut[256];              // byte -> 8 bf16 (1.0 where the bit is set)
    __shared__ __attribute__((aligned(16))) float sT[4 * 2 * 32 * CH_T_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    NlLossScalars ls;
    int P = a.P;
    if (MODE >= 1) { ls = *a.ls; P = ls.P; }
    const float b3 = params[NL_OFF_B3];
    // one resource over W2A | W2XA (adjacent in the workspace): a run-time choice between two resources would become a waterfall loop
    const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ws + CH_WS_W2A_OFF), 0, 6 * CH_PLANE_BYTES, 0x00020000);
    const int voff = lane * 16;
    float* sTw = sT + w * (2 * 32 * CH_T_STRIDE);

    // ---- the weight stages: NST per pass, walked cyclically.  Stage q < 16: forward output tile q & 7 (planes W2A); q >= 16: dgrad tile
    //      q - 16 (planes W2XA).  This wave moves fragments 12 w .. 12 w + 11 of a stage: one per k-step, written to LDS four k-steps later.
    constexpr int NST = MODE == 0 ? 16 : 24;
    uint4 st[5];
    auto stage_soff = [&](int q) { return q >= 16 ? 3 * CH_PLANE_BYTES + (q - 16) * CH_STAGE_BYTES : (q & 7) * CH_STAGE_BYTES; };
    auto fill_load = [&](int q, int i) {
        const int f = 12 * w + i;
        st[i % 5] = ch_bload4(rsW, voff, stage_soff(q) + (f % 3) * CH_PLANE_BYTES + (f / 3) * 1024);
    };
    auto fill_store = [&](int buf, int i) { sRing[buf][12 * w + i][lane] = st[i % 5]; };
#pragma unroll
    for (int i = 0; i < 12; ++i) { fill_load(0, i); fill_store(0, i); }
    __syncthreads();
    int lbuf = 0;
    uint4 af[2][3];                                      // A fragments of the current / next k-step; af[0] is preloaded across stage boundaries
#pragma unroll
    for (int p = 0; p < 3; ++p) af[0][p] = sRing[0][p][lane];

    // weight-gradient accumulators (MODE 2)
    float aB3 = 0.f;
    double lossFs = 0.0, lossSdf = 0.0;
    float* accw = MODE == 2 ? a.partials + (size_t)blockIdx.x * NL_DEC_PARAMS + NL_OFF_W2 + w * CH_ACC_FLOATS : nullptr;
    // ReLU-word assembly (MODE 2): lane j < 32 collects the word of unit j of the current 32-unit tile
    const int mw_r = (l31 & 3) + 4 * (l31 >> 3), mw_hi = (l31 >> 2) & 1;

    const int nsub = (P + 31) >> 5, npass = (nsub + 1) >> 1;
    // inputs of a pass: this lane's sample of either sub-tile (channels 8 lh .. 8 lh + 7) and its loss geometry; the NEXT pass's are
    // fetched under the current pass's backward phase
    float4 nx[2][2];
    float ncz[2], ncd[2];
    auto fetch_inputs = [&](int pass) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int gg = ((2 * pass + u) << 5) + l31;
            nx[u][0] = make_float4(0.f, 0.f, 0.f, 0.f); nx[u][1] = nx[u][0]; ncz[u] = 0.f; ncd[u] = 0.f;
            if (gg < P) {
                const float4* xp = reinterpret_cast<const float4*>(a.X + (size_t)gg * NL_C + 8 * lh);
                nx[u][0] = xp[0]; nx[u][1] = xp[1];
                if (MODE >= 1) {
                    const int ray = a.s_ray[gg];
                    ncz[u] = a.s_depth[gg] * a.cos_gt[ray]; ncd[u] = a.gt_dist[ray];
                }
            }
        }
    };
    fetch_inputs(blockIdx.x * 4 + w);
    int pass_no = 0;
    // the four waves walk the stages in lock step: a wave whose pass lies beyond the data computes on zeros and writes nothing
    for (int base = blockIdx.x * 4; base < npass; base += gridDim.x * 4, ++pass_no) {
        CH_STAMP(0);
        const int pass = base + w;
        float xin[2][8], cz[2], cd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            xin[u][0] = ch_from_acc(nx[u][0].x); xin[u][1] = ch_from_acc(nx[u][0].y); xin[u][2] = ch_from_acc(nx[u][0].z); xin[u][3] = ch_from_acc(nx[u][0].w);
            xin[u][4] = ch_from_acc(nx[u][1].x); xin[u][5] = ch_from_acc(nx[u][1].y); xin[u][6] = ch_from_acc(nx[u][1].z); xin[u][7] = ch_from_acc(nx[u][1].w);
            cz[u] = ch_from_acc(ncz[u]); cd[u] = ch_from_acc(ncd[u]);
        }
        if (MODE == 0) fetch_inputs(pass + gridDim.x * 4);
        unsigned m1w[2][4], m2w[2][4];
        float ds[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile = 2 * pass + u, g = (tile << 5) + l31;
            const bool live = g < P;
            // ---------------- layer 1: H1^T = relu(W1 X^T + b1), split into three bf16 operand planes (registers) ----------------
            uint4 hb[16][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) { m1w[u][q] = 0u; m2w[u][q] = 0u; }
            // software pipeline over the eight 32-unit tiles: slot k of tile ut = the k-th fp32 MFMA of tile ut + 1 (a dependent chain:
            // 64 cycles each) + the split of value pair k of tile ut (~20 VALU instructions) - neither phase waits for the other.
            // VALU diet: the packed word of a plane, shifted / masked, IS the pair of truncated values; the two residuals are one
            // packed subtraction.
            auto l1_bias = [&](int ut, f32x16& c) {
                const float4* tb = reinterpret_cast<const float4*>(sTab + ut * 32 + lh * 16);
                const float4 t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
                c[0] = t0.x; c[1] = t0.y; c[2] = t0.z; c[3] = t0.w; c[4] = t1.x; c[5] = t1.y; c[6] = t1.z; c[7] = t1.w;
                c[8] = t2.x; c[9] = t2.y; c[10] = t2.z; c[11] = t2.w; c[12] = t3.x; c[13] = t3.y; c[14] = t3.z; c[15] = t3.w;
            };
            f32x16 cc;
            float wrow[8];
            auto l1_weights = [&](int ut) {
                const float4* wr = reinterpret_cast<const float4*>(sW1 + (32 * ut + l31) * CH_W1_STRIDE + 8 * lh);
                const float4 w0 = wr[0], w1 = wr[1];
                wrow[0] = w0.x; wrow[1] = w0.y; wrow[2] = w0.z; wrow[3] = w0.w; wrow[4] = w1.x; wrow[5] = w1.y; wrow[6] = w1.z; wrow[7] = w1.w;
            };
            f32x16 cn;                                                    // bias of the tile after the next one: read one tile ahead
            float wnext[8];
            l1_bias(0, cc); l1_weights(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) cc = MFMA32(wrow[k], xin[u][k], cc);
            l1_bias(1, cn); l1_weights(1);
#pragma unroll
            for (int ut = 0; ut < 8; ++ut) {
                float cp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[r] = cc[r];
                if (ut + 1 < 8) {
                    cc = cn;
#pragma unroll
                    for (int k = 0; k < 8; ++k) wnext[k] = wrow[k];
                    if (ut + 2 < 8) { l1_bias(ut + 2, cn); l1_weights(ut + 2); }
                }
                unsigned bits = 0u, hi[8], mid[8], lo[8];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (ut + 1 < 8) cc = MFMA32(wnext[q], xin[u][q], cc);
                    f32x2 hv; hv.x = fmaxf(cp[2 * q], 0.f); hv.y = fmaxf(cp[2 * q + 1], 0.f);
                    bits |= (hv.x > 0.f ? (1u << (2 * q)) : 0u) | (hv.y > 0.f ? (2u << (2 * q)) : 0u);
                    hi[q] = ch_pack_hi16(hv.x, hv.y);
                    f32x2 t; t.x = __uint_as_float(hi[q] << 16); t.y = __uint_as_float(hi[q] & 0xFFFF0000u);
                    const f32x2 r = hv - t;
                    mid[q] = ch_pack_hi16(r.x, r.y);
                    f32x2 t2; t2.x = __uint_as_float(mid[q] << 16); t2.y = __uint_as_float(mid[q] & 0xFFFF0000u);
                    const f32x2 r2 = r - t2;
                    lo[q] = ch_pack_hi16(r2.x, r2.y);
                    asm volatile("" : "+v"(hi[q]), "+v"(mid[q]), "+v"(lo[q]));      // computed HERE (the compiler would sink the split to the first use)
                    __builtin_amdgcn_sched_barrier(0);
                }
                hb[2 * ut][0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); hb[2 * ut + 1][0] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
                hb[2 * ut][1] = make_uint4(mid[0], mid[1], mid[2], mid[3]); hb[2 * ut + 1][1] = make_uint4(mid[4], mid[5], mid[6], mid[7]);
                hb[2 * ut][2] = make_uint4(lo[0], lo[1], lo[2], lo[3]); hb[2 * ut + 1][2] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
                m1w[u][ut >> 1] |= bits << (16 * (ut & 1));
                // materialise the word HERE: left alone, the compiler sinks the 256 compares to where the dgrad first reads the
                // words and keeps every fp32 H1 value alive (and spilled) until then
                asm volatile("" : "+v"(m1w[u][ut >> 1]));
            }
            CH_STAMP(1 + 3 * u);
            // ---------------- layer 2 + output layer: s = w3 . relu(W2 H1 + b2) + b3, H2 never stored ----------------
            // A stage is 16 k-steps x NP MFMAs, one issue slot each (a scheduling barrier closes every slot: the order below IS the
            // schedule).  The stage's own memory instructions and the EPILOGUE OF THE PREVIOUS output tile (ReLU, row-sum FMAs, mask
            // bits, the ReLU words of the dW2 kernel) ride in the slots, a few VALU instructions each: a VALU instruction next to an
            // MFMA costs 1-2 cycles of matrix pipe, a block of them after the loop its full 5-6 (scripts/micro/mfma_valu.hip).
            float spart = 0.f;
            f32x16 h;
            float hp[16];                                                 // the previous tile's pre-activations (copied out of the accumulator)
            float wv[16], hv = 0.f;
            unsigned ebits = 0u, mword = 0u;
            constexpr int IPE = MODE == 2 ? 4 : 2;                         // epilogue items per element
            auto epi_item = [&](int slot, int ntp) {
                if (slot < 4) {
                    const float4 t = *reinterpret_cast<const float4*>(sTab + 512 + ntp * 32 + lh * 16 + 4 * slot);
                    wv[4 * slot] = t.x; wv[4 * slot + 1] = t.y; wv[4 * slot + 2] = t.z; wv[4 * slot + 3] = t.w;
                    if (slot == 0) { ebits = 0u; mword = 0u; }
                    return;
                }
                const int k = slot - 4;
                if (k < 16 * IPE) {
                    const int r = k / IPE, part = k % IPE;
                    if (part == 0) { hv = fmaxf(hp[r], 0.f); ebits |= hv > 0.f ? (1u << r) : 0u; }
                    else if (part == 1) spart = fmaf(hv, wv[r], spart);
                    else if (part == 2) {
                        const unsigned long long bal = __ballot(hv > 0.f);
                        const unsigned pick = mw_hi ? (unsigned)(bal >> 32) : (unsigned)bal;
                        mword = (mw_r == r) ? pick : mword;
                    }
                    return;
                }
                if (k == 16 * IPE) {
                    // 128-bit shift register: after the 8th tile, tile nt sits in bits [16 nt, 16 nt + 16)
                    m2w[u][0] = (m2w[u][0] >> 16) | (m2w[u][1] << 16); m2w[u][1] = (m2w[u][1] >> 16) | (m2w[u][2] << 16);
                    m2w[u][2] = (m2w[u][2] >> 16) | (m2w[u][3] << 16); m2w[u][3] = (m2w[u][3] >> 16) | (ebits << 16);
                    if (MODE == 2 && lh == 0 && tile < nsub) a.relu2_nat[(size_t)tile * NL_W + 32 * ntp + l31] = mword;
                }
            };
            auto fwd_stage = [&](auto prev_tag, int nt) {
                constexpr bool PREV = decltype(prev_tag)::value;
                const int qn = (8 * u + nt + 1) % NST;                      // the stage this one fills
                {
                    const float4* tb = reinterpret_cast<const float4*>(sTab + 256 + nt * 32 + lh * 16);
                    const float4 t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
                    h[0] = t0.x; h[1] = t0.y; h[2] = t0.z; h[3] = t0.w; h[4] = t1.x; h[5] = t1.y; h[6] = t1.z; h[7] = t1.w;
                    h[8] = t2.x; h[9] = t2.y; h[10] = t2.z; h[11] = t2.w; h[12] = t3.x; h[13] = t3.y; h[14] = t3.z; h[15] = t3.w;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
#pragma unroll
                    for (int m = 0; m < NP; ++m) {
                        const int pa = NP == 9 ? m / 3 : (m < 3 ? 0 : (m < 5 ? 1 : 2)), pb = NP == 9 ? m % 3 : (m < 3 ? m : (m < 5 ? m - 3 : 0));
                        h = MFMA_BF16(__builtin_bit_cast(bf16x8, af[s & 1][pa]), __builtin_bit_cast(bf16x8, hb[s][pb]), h);
                        if (m < 3 && s + 1 < 16) af[(s + 1) & 1][m] = sRing[lbuf][3 * (s + 1) + m][lane];
                        if (m == 3 && s >= 4) fill_store(lbuf ^ 1, s - 4);
                        if (m == 4 && s < 12) fill_load(qn, s);
                        if (PREV) epi_item(NP * s + m, nt - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                nl_lds_barrier();                                         // next stage complete in LDS; this one free to be refilled
                lbuf ^= 1;
#pragma unroll
                for (int p = 0; p < 3; ++p) af[0][p] = sRing[lbuf][p][lane];
#pragma unroll
                for (int r = 0; r < 16; ++r) hp[r] = h[r];
            };
            fwd_stage(std::false_type{}, 0);
#pragma unroll 1
            for (int nt = 1; nt < 8; ++nt) fwd_stage(std::true_type{}, nt);
#pragma unroll
            for (int slot = 0; slot < 4 + 16 * IPE + 1; ++slot) epi_item(slot, 7);       // the last tile's epilogue: nothing left to hide it under
            CH_STAMP(2 + 3 * u);
            const float sv = (spart + __shfl_xor(spart, 32)) + b3;
            if (MODE == 0) {
                if (live && lh == 0) a.sdf[g] = sv;
            } else if (live) {
                // ---------------- loss gradient (criterion.py): a lane constant ----------------
                bool f, m;
                nl_loss_masks(cz[u], cd[u], ls.tau, ls.max_depth, &f, &m);
                float q1, q2;
                ds[u] = nl_loss_grad(sv, cz[u], cd[u], f, m, ls, &q1, &q2);
                if (lh == 0) {
                    a.sdf[g] = sv; a.dsdf[g] = ds[u];
                    lossFs += (double)q1; lossSdf += (double)q2;
                    if (MODE == 2) aB3 += ds[u];
                }
            }
            CH_STAMP(3 + 3 * u);
        }
        if (MODE == 0) continue;
        // ---------------- dgrad over both sub-tiles: dH1^T = ((w3 W2)^T mask^T) * dsdf * [H1 > 0]; layer-1 backward per 32-unit tile ----------------
        fetch_inputs(pass + gridDim.x * 4);                     // the next pass's inputs arrive under this pass's backward
        uint4 mf[2][16];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 16; ++s) mf[u][s] = sLut[(m2w[u][s >> 2] >> (8 * (s & 3))) & 0xFFu];
        float xw[2][8];
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) { const int gi = ((2 * pass + u) << 5) + 4 * ii + lq; xw[u][ii] = gi < P ? a.X[(size_t)gi * NL_C + l15] : 0.f; }
        }
        f32x4 dxa[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) for (int sub = 0; sub < 2; ++sub) for (int r = 0; r < 4; ++r) dxa[u][sub][r] = 0.f;
        CH_STAMP(7);
        // A dgrad stage is 16 k-steps x 6 MFMAs = 96 issue slots.  The layer-1 backward of the PREVIOUS stage's tile rides in their
        // shadow (the order below IS the schedule: a scheduling barrier closes every slot):
        //   slots  0..39  the 32 raw dgrad values per lane -> * dsdf * [H1 > 0] -> the wave's transposition tiles in LDS (8 x 16 B per lane)
        //   k-step 7      operand reads of units 0, 1;  k-steps 8..15: the reads of the next two units, one per slot, and after the
        //                 sixth bf16 MFMA a BURST of the 8 (train) / 4 (frozen) 16x16x4 fp32 MFMAs of two units - alternating the two
        //                 MFMA kinds one by one costs ~10 cycles per switch (scripts/micro/mfma_shapes.hip: 84 cycles per pair, not 64).
        //                 unit j = (sub-tile u = j >> 3, i = j & 7): dX of both 16-sample halves over units 4 i .. 4 i + 3 of the tile,
        //                 dW1 of both 16-unit halves over samples 4 i .. 4 i + 3 (+ db1)
        f32x16 gacc[2], gprev[2];
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0, wl0 = w0, wl1 = w0;      // dW1 / db1 of the tile in its layer-1 backward; wl*, bl*: loaded for the next
        float b0 = 0.f, b1 = 0.f, bl0 = 0.f, bl1 = 0.f;
        float2 rta[4], raa[4];                                            // operands of two unit pairs in flight
        float rbw[4];
        float4 wv4;                                                       // the float4 being assembled for the transposition tile
        auto l1bwd_reads = [&](int j, int part, int ktp) {                 // operand reads of unit j: three LDS instructions
            const int u = j >> 3, i = j & 7;
            const float* sTu = sTw + u * (32 * CH_T_STRIDE);
            if (part == 0) {          // dX A operand: dH1 rows l15 / 16 + l15, units lq + 4 i
                rta[j & 3].x = sTu[l15 * CH_T_STRIDE + lq + 4 * i]; rta[j & 3].y = sTu[(16 + l15) * CH_T_STRIDE + lq + 4 * i];
            } else if (part == 1) {   // dX B operand: W1 rows of those units
                rbw[j & 3] = sW1[(32 * ktp + lq + 4 * i) * CH_W1_STRIDE + l15];
            } else if (MODE == 2) {   // dW1 A operand: dH1 columns l15 / 16 + l15, samples lq + 4 i
                raa[j & 3].x = sTu[(lq + 4 * i) * CH_T_STRIDE + l15]; raa[j & 3].y = sTu[(lq + 4 * i) * CH_T_STRIDE + l15 + 16];
            }
        };
        auto l1bwd_unit = [&](int j) {
            const int u = j >> 3, i = j & 7;
            dxa[u][0] = MFMA16(rta[j & 3].x, rbw[j & 3], dxa[u][0]);
            dxa[u][1] = MFMA16(rta[j & 3].y, rbw[j & 3], dxa[u][1]);
            if (MODE == 2) {
                w0 = MFMA16(raa[j & 3].x, xw[u][i], w0); w1 = MFMA16(raa[j & 3].y, xw[u][i], w1);
                b0 += raa[j & 3].x; b1 += raa[j & 3].y;
            }
        };
        auto l1bwd_item = [&](int s, int m, int ktp) {                     // ktp: the tile (stage) whose layer-1 backward this is
            const int slot = 6 * s + m;
            if (slot < 40) {
                const int j = slot / 5, u = j >> 2, q = j & 3, part = slot % 5;
                const unsigned bits = (m1w[u][0] >> (4 * q)) & 0xFu;        // m1w is shifted down 16 bits per tile (end of dgrad_stage)
                const float dsu = ds[u];
                if (part == 0) wv4.x = gprev[u][4 * q] * ((bits & 1u) ? dsu : 0.f);
                else if (part == 1) wv4.y = gprev[u][4 * q + 1] * ((bits & 2u) ? dsu : 0.f);
                else if (part == 2) wv4.z = gprev[u][4 * q + 2] * ((bits & 4u) ? dsu : 0.f);
                else if (part == 3) wv4.w = gprev[u][4 * q + 3] * ((bits & 8u) ? dsu : 0.f);
                else *reinterpret_cast<float4*>(sTw + u * (32 * CH_T_STRIDE) + l31 * CH_T_STRIDE + 8 * q + 4 * lh) = wv4;   // column = unit nl_chain_unit(r, lh)
                return;
            }
            if (s < 7) return;
            if (s + 1 < 16 || s == 7) {                                  // reads of the pair the NEXT k-step multiplies
                const int jn = 2 * (s - 7) + m / 3;
                if (jn < 16) l1bwd_reads(jn, m % 3, ktp);
            }
            if (s >= 8 && m == 5) { l1bwd_unit(2 * (s - 8)); l1bwd_unit(2 * (s - 8) + 1); }
        };
        auto acc_ptrs = [&](int kt, f32x4*& aw, float*& ab) {
            aw = reinterpret_cast<f32x4*>(accw + (2 * kt * 64 + lane) * 4);
            ab = accw + 16 * 64 * 4 + 2 * kt * 64 + lane;
        };
        auto dgrad_stage = [&](auto prev_tag, int kt) {
            constexpr bool PREV = decltype(prev_tag)::value;
            const int qn = (16 + kt + 1) % NST;
#pragma unroll
            for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) gacc[u][r] = 0.f;
            if (MODE == 2 && pass_no > 0) {                       // this tile's parked accumulators: needed one stage from now
                f32x4* aw; float* ab; acc_ptrs(kt, aw, ab);
                wl0 = aw[0]; wl1 = aw[64]; bl0 = ab[0]; bl1 = ab[64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    gacc[m & 1] = MFMA_BF16(__builtin_bit_cast(bf16x8, af[s & 1][m >> 1]), __builtin_bit_cast(bf16x8, mf[m & 1][s]), gacc[m & 1]);
                    if (m < 3 && s + 1 < 16) af[(s + 1) & 1][m] = sRing[lbuf][3 * (s + 1) + m][lane];
                    if (m == 3 && s >= 4) fill_store(lbuf ^ 1, s - 4);
                    if (m == 4 && s < 12) fill_load(qn, s);
                    if (PREV) l1bwd_item(s, m, kt - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (PREV && MODE == 2) { f32x4* aw; float* ab; acc_ptrs(kt - 1, aw, ab); aw[0] = w0; aw[64] = w1; ab[0] = b0; ab[64] = b1; }
            if (PREV) {                                          // the next tile's H1 ReLU bits move to the low half-word
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    m1w[u][0] = (m1w[u][0] >> 16) | (m1w[u][1] << 16); m1w[u][1] = (m1w[u][1] >> 16) | (m1w[u][2] << 16);
                    m1w[u][2] = (m1w[u][2] >> 16) | (m1w[u][3] << 16); m1w[u][3] >>= 16;
                }
            }
            nl_lds_barrier();
            lbuf ^= 1;
#pragma unroll
            for (int p = 0; p < 3; ++p) af[0][p] = sRing[lbuf][p][lane];
            gprev[0] = gacc[0]; gprev[1] = gacc[1];
            w0 = wl0; w1 = wl1; b0 = bl0; b1 = bl1;
        };
        dgrad_stage(std::false_type{}, 0);
        CH_STAMP(8);
#pragma unroll 1
        for (int kt = 1; kt < 8; ++kt) dgrad_stage(std::true_type{}, kt);
        CH_STAMP(9);
        // the last tile's layer-1 backward: nothing left to hide it under
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float* sTu = sTw + u * (32 * CH_T_STRIDE);
            const unsigned b1bits = m1w[u][0] & 0xFFFFu;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = gprev[u][4 * q] * (((b1bits >> (4 * q)) & 1u) ? ds[u] : 0.f);
                v.y = gprev[u][4 * q + 1] * (((b1bits >> (4 * q + 1)) & 1u) ? ds[u] : 0.f);
                v.z = gprev[u][4 * q + 2] * (((b1bits >> (4 * q + 2)) & 1u) ? ds[u] : 0.f);
                v.w = gprev[u][4 * q + 3] * (((b1bits >> (4 * q + 3)) & 1u) ? ds[u] : 0.f);
                *reinterpret_cast<float4*>(sTu + l31 * CH_T_STRIDE + 8 * q + 4 * lh) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();                         // same-wave LDS write -> read (in order in hardware; pins the compiler)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* sTu = sTw + u * (32 * CH_T_STRIDE);
            {
                const float* ta = sTu + l15 * CH_T_STRIDE + lq;
                const float* wb = sW1 + (32 * 7 + lq) * CH_W1_STRIDE + l15;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float bw = wb[4 * kk * CH_W1_STRIDE];
                    dxa[u][0] = MFMA16(ta[4 * kk], bw, dxa[u][0]);
                    dxa[u][1] = MFMA16(ta[16 * CH_T_STRIDE + 4 * kk], bw, dxa[u][1]);
                }
            }
            if (MODE == 2) {
                const float* ta = sTu + lq * CH_T_STRIDE + l15;
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
                    const float a0 = ta[4 * ii * CH_T_STRIDE], a1 = ta[4 * ii * CH_T_STRIDE + 16];
                    w0 = MFMA16(a0, xw[u][ii], w0);
                    w1 = MFMA16(a1, xw[u][ii], w1);
                    b0 += a0; b1 += a1;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (MODE == 2) { f32x4* aw; float* ab; acc_ptrs(7, aw, ab); aw[0] = w0; aw[64] = w1; ab[0] = b0; ab[64] = b1; }
        CH_STAMP(10);
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) ch_keep_in_acc(xw[u][ii]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = ((2 * pass + u) << 5) + 16 * sub + 4 * lq + r;
                    if (gi < P) a.dX[(size_t)gi * NL_C + l15] = dxa[u][sub][r];
                }
    }
    if (MODE == 0) return;

    // ---------------- loss sums ----------------
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { lossFs += __shfl_xor(lossFs, off); lossSdf += __shfl_xor(lossSdf, off); }
    if (lane == 0 && (lossFs != 0.0 || lossSdf != 0.0)) { atomicAdd(&a.dcounters[NLD_FS_SQ], lossFs); atomicAdd(&a.dcounters[NLD_SDF_SQ], lossSdf); }
    if (MODE != 2) return;

    // ---------------- weight-gradient slab of this workgroup: the four waves' parked accumulators summed in a fixed order ----------
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) aB3 += __shfl_xor(aB3, off);
    if (lane == 0) sTab[w] = aB3;
    __threadfence();
    __syncthreads();                                             // every wave's last accumulator stores are visible
    float* base = a.partials + (size_t)blockIdx.x * NL_DEC_PARAMS;
    const float* acc0 = base + NL_OFF_W2;
    const bool any = blockIdx.x * 4 < npass;                     // a workgroup without a pass never wrote its scratch
    for (int i = tid; i < NL_W * NL_C; i += CH_THREADS) {
        const int u = i >> 4, c = i & 15, e = ((u >> 4) * 64 + ((u >> 2) & 3) * 16 + c) * 4 + (u & 3);
        float s = 0.f;
        if (any) s = ((acc0[e] + acc0[CH_ACC_FLOATS + e]) + acc0[2 * CH_ACC_FLOATS + e]) + acc0[3 * CH_ACC_FLOATS + e];
        base[NL_OFF_W1 + i] = s;
    }
    {
        const int e = 16 * 64 * 4 + (tid >> 4) * 64 + (tid & 15);
        float s = 0.f;
        if (any) {
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const float* ap = acc0 + ww * CH_ACC_FLOATS + e;
                s += ((ap[0] + ap[16]) + ap[32]) + ap[48];
            }
        }
        base[NL_OFF_B1 + tid] = s;
    }
    if (tid == 0) base[NL_OFF_B3] = ((sTab[0] + sTab[1]) + sTab[2]) + sTab[3];
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
    const int need = nl_div_up(nl_div_up(P, 64), 4);
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
