// nl_common.h -- shared declarations of the HIP implementation (not part of the public C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nl_device_math.h"

// status codes returned by every nl_* entry point (include/nerfloam_hip.h)
#define NL_OK 0
#define NL_ERR_INVALID_ARG 1
#define NL_ERR_LAUNCH 2
#define NL_ERR_NO_DEVICE 3
#define NL_ERR_CAPACITY 4

// ---------------------------------------------------------------------------------------------
// Launch shapes by ray count: THE table.  Every entry is a measured crossover (the profile that fixed it in brackets); the code that
// selects a launch shape (nl_geometry.hip intersect_launch / scan_launch / nl_sample_rays*, nl_iteration.cpp, pipeline.py through
// nl_isect_lanes_for) reads these names and nothing else.
// ---------------------------------------------------------------------------------------------
#define NL_RAYS_ONE_WORKGROUP_SCAN 4096     // <=: hit-ray scan (+ the DFS fallback) and sample-offset scan (+ the loss scalars) by ONE workgroup (r01_m timeline)
#define NL_RAYS_SINGLE_LAUNCH_SCAN 32768    // <=: those scans as one launch of <= 8 workgroups that sum what is in front of them; beyond: two launches (r05_s)
#define NL_RAYS_FUSED_SAMPLER      8192     // <=: count pass + scan + normalisers + emit pass as ONE launch, step-parallel count pass (r01_l, r03_e)
#define NL_RAYS_ISECT_32_LANES     4096     // <=: 32 lanes per ray in the work-list intersect on every map (r04_n)
#define NL_RAYS_ISECT_16_LANES     16384    // <=: 16 lanes per ray - 32 on a map of >= NL_BLOCKS_WIDE_MAP children blocks -, 8 beyond (r01_k, r04_n)
#define NL_RAYS_DECODER_SPLIT      16384    // >: the fp16-pair fused decoder kernel as two 4-wave workgroups per CU (k_decoder2), <=: one 8-wave workgroup (r06_j: 2048 rays +3 %, 8192 +4 %, 16 384 0 %, 32 768 -1.3 %, 131 072 -4 % per iteration)
#define NL_BLOCKS_WIDE_MAP         60000    // children blocks from which a ray's front is wider than 16 nodes per round (40 scans / 71 k blocks gain, 15 / 38 k lose)

// device counter block: int32[NL_CNT_INTS] followed (8-byte aligned) by double[NL_CNT_DOUBLES]
enum {
    NLC_R = 0,          // number of rays with >=1 hit
    NLC_HMAX,           // max valid hits per ray over the batch ("P" of the sampler)
    NLC_SMAX,           // max samples per ray
    NLC_P,              // total valid samples
    NLC_NFS,            // front-mask count over VALID samples
    NLC_NSDF,           // sdf-mask count over VALID samples
    NLC_INV_FS_RAYS,    // sum over hit rays of front_inv(r)            (invalid slots: z = 80*cos)
    NLC_INV_FS_CNT,     // sum over hit rays of front_inv(r) * cnt_r
    NLC_INV_SDF_RAYS,   // sum over hit rays of sdfm_inv(r)
    NLC_INV_SDF_CNT,    // sum over hit rays of sdfm_inv(r) * cnt_r
    NLC_OVERFLOW,       // set when P exceeded the sample capacity
    NLC_GUARD,          // set when a ray's summed interval length exceeds 10*MAX_DEPTH (reference returns None)
    NLC_R_OFFSET,       // multi-GPU: number of hit rays on lower ranks (global rank of local hit-ray 0)
    NLC_R_GLOBAL,       // multi-GPU: global number of hit rays (== NLC_R on one GPU)
    NLC_ISECT_OVF,      // rays handed from the queue intersect kernel to the sequential DFS fallback
    NLC_TICKET,         // workgroups of the fused sampler that have finished (the last one computes the loss normalisers)
    NL_CNT_INTS = 16
};
enum {
    NLD_FS_SQ = 0,      // sum of fs residual^2 over valid samples
    NLD_SDF_SQ,         // sum of sdf residual^2 over valid samples
    NLD_INV_D2,         // sum over hit rays of sdfm_inv(r) * d_r^2
    NLD_INV_D2CNT,      // sum over hit rays of sdfm_inv(r) * cnt_r * d_r^2
    NL_CNT_DOUBLES = 4
};
#define NL_CNT_BYTES (NL_CNT_INTS * 4 + NL_CNT_DOUBLES * 8)

// decoder parameter block (floats), nn.Linear layouts W[out][in]
#define NL_C 16                 // embedding channels = decoder input width
#define NL_W 256                // hidden width
#define NL_OFF_W1 0
#define NL_OFF_B1 (NL_OFF_W1 + NL_W * NL_C)
#define NL_OFF_W2 (NL_OFF_B1 + NL_W)
#define NL_OFF_B2 (NL_OFF_W2 + NL_W * NL_W)
#define NL_OFF_W3 (NL_OFF_B2 + NL_W)
#define NL_OFF_B3 (NL_OFF_W3 + NL_W)
#define NL_DEC_PARAMS 70401
// decoder weight workspace (floats; include/nerfloam_hip.h nl_decoder_transpose_w2): W2^T fp32 | W2X, W2TX (3 bf16 planes each) | W2H, W2TH (2 fp16 planes each)
//   | W1F, W1X (2 fp16 planes each: W1 * 2^8 in its two operand forms, round 6)
#define NL_DEC_WS_W1F_OFF (NL_W * NL_W + 2 * 3 * NL_W * NL_W / 2 + 2 * 2 * NL_W * NL_W / 2)      // floats
#define NL_DEC_WS_W1X_OFF (NL_DEC_WS_W1F_OFF + 2 * NL_W * NL_C / 2)
//   | range block: 16 words (round 6) - what the fp16-pair arithmetic needs to know to notice that an operand left its range
#define NL_DEC_WS_RANGE_OFF (NL_DEC_WS_W1X_OFF + 2 * NL_W * NL_C / 2)
#define NL_DEC_WS_TOTAL (NL_DEC_WS_RANGE_OFF + 16)
static_assert(NL_DEC_WS_W1F_OFF == 393216 && NL_DEC_WS_TOTAL == 401424, "NL_DEC_WS_FLOATS of include/nerfloam_hip.h");
// Range block.  Word 0 is written where the operand planes are (k_prepare_w2x from scratch - it clears the word first -, k_optim_step after every step),
// word 4 is the STICKY status the decoder kernels raise and the optimiser latches into the call status; the rest is reserved.
enum {
    NLR_PLANE_SAT = 0,      // uint: NL_SAT_PLANES when a weight plane clipped at +-65504 (|W1| or |W2| >= 255.9, |w3_j W2[j][k]| >= 63.97)
    NLR_STATUS = 4,         // uint: NL_SAT_* bits, raised (atomicOr) by k_decoder / k_decoder2 / k_decoder_fwd under gemm modes 4 / 5
    NLR_WORDS = 16
};
#define NL_SAT_X 1u         // an input left the range of X * 2^6 (|X| > 1023.5; with a trainable decoder |X| >= 255.9: U = sigma dsdf 16 X), or is NaN / Inf  (every sample)
#define NL_SAT_H1 2u        // H1 * 2^4 reached 65504 (H1 >= 4094) and was clipped           (checked on the FIRST tile of every workgroup: all samples of a launch of
#define NL_SAT_Q 4u         // a dgrad accumulator (2^10 dH1 / dsdf) reached +-65504           <= 512 tiles - the live 2048 / 4096-ray shapes -, a 3 % sample of the full scan)
#define NL_SAT_PLANES 8u    // a weight operand plane clipped (exact)
// element (k = hidden unit, c = channel, plane) of the two forms, in 16-bit elements from the start of the form:
//   W1F [column tile k >> 5][plane][lane = 32 (c >> 3) + (k & 31)][c & 7]   - layer 1's B fragments (32x32x16: lane n holds 8 consecutive c)
//   W1X [k-step k >> 5][plane][lane = 16 ((k >> 3) & 3) + c][k & 7]        - dX's B fragments (16x16x32: lane (c, q) holds k = 32 s + 8 q + e)
#define NL_W1F_INDEX(k, c, plane) (((((k) >> 5) * 2 + (plane)) * 64 + 32 * ((c) >> 3) + ((k) & 31)) * 8 + ((c) & 7))
#define NL_W1X_INDEX(k, c, plane) (((((k) >> 5) * 2 + (plane)) * 64 + 16 * (((k) >> 3) & 3) + (c)) * 8 + ((k) & 7))
#define NL_DEC_WS_W2H_OFF16 (2 * 3 * NL_W * NL_W)          // 16-bit elements from the start of W2X to the start of W2H
static_assert(NL_DEC_PARAMS == NL_OFF_B3 + 1, "decoder parameter block");

#define NL_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return NL_ERR_LAUNCH;        \
    } while (0)

static inline int nl_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// Embedding rows touched since the optimiser was created (NlTouchedRows of include/nerfloam_hip.h): one bit per row + the list of
// the rows whose bit is set.  Whoever first adds to a row's gradient accumulators appends the row - the optimiser then sweeps the
// list instead of the whole table (a row that was never touched has zero gradient and zero moments: Adam leaves it alone, so
// skipping it is bit-identical to the dense sweep the reference's torch.optim.Adam performs).
struct NlTouchedDev { int* list; int* count; unsigned* flags; int copies; long long copy_stride; };
__device__ __forceinline__ void nl_touch_row(const NlTouchedDev& t, int row)
{
    if (!t.flags) return;
    unsigned* wp = t.flags + (row >> 5);
    const unsigned bit = 1u << (row & 31);
    if (*reinterpret_cast<volatile unsigned*>(wp) & bit) return;               // the common case after the first iteration of a call
    if (!(atomicOr(wp, bit) & bit)) t.list[atomicAdd(t.count, 1)] = row;
}

// max of a non-negative value over the wave (DPP inside the 16-lane rows, the four rows through scalar registers: no LDS round trips)
__device__ __forceinline__ float nl_wave_max_nonneg(float v)
{
    int x = __float_as_int(v);                           // non-negative floats order like their bit patterns
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true));        // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true));       // row_half_mirror
    x = max(x, __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true));       // row_mirror
    const int m = max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
    return __int_as_float(m);
}
// Range watch of a decoder kernel under the fp16-pair arithmetic: ONE register per lane.  `xw` is the running maximum of the bit patterns of |X| over everything the
// lane staged (non-negative floats order like their bit patterns; NaN / Inf are the largest), and a lane that finds a clipped H1 / dgrad accumulator on its workgroup's
// first tile parks 0xFFFFFFF0 | NL_SAT_* there (above every |X| pattern, so the maximum keeps it).  The tile loop pays ~1.5 instructions per staged value and, on the
// first tile only, ~100 over the accumulators.
#define NL_XW_SEEN 0xFFFFFFF0u
__device__ __forceinline__ unsigned nl_xw_update(unsigned xw, float a, float b)
{
    return max(max(__float_as_uint(a) & 0x7FFFFFFFu, __float_as_uint(b) & 0x7FFFFFFFu), xw);
}
__device__ __forceinline__ unsigned nl_xw_mark(unsigned xw, unsigned sat_bit) { return (xw >= NL_XW_SEEN ? xw : NL_XW_SEEN) | sat_bit; }
// End of the kernel (every lane calls; one lane per wave acts): the sticky status word of the weight workspace gets the bits of whatever left its range
__device__ __forceinline__ void nl_range_check(const float* __restrict__ ws, unsigned xw, bool train)
{
    const bool marked = xw >= NL_XW_SEEN;
    unsigned bits = (__ballot(marked && (xw & NL_SAT_H1)) ? NL_SAT_H1 : 0u) | (__ballot(marked && (xw & NL_SAT_Q)) ? NL_SAT_Q : 0u);
    const float lim = train ? NL_F16_MAX / 256.0f : NL_F16_MAX / NL_F16_SX;      // |X| 2^6 (and, trainable, sigma dsdf 16 |X| < 256 |X|) inside the fp16 range
    if (__ballot(!marked && !(__uint_as_float(xw) <= lim))) bits |= NL_SAT_X;      // (a NaN / Inf pattern fails the comparison too)
    if ((threadIdx.x & 63) != 0) return;
    unsigned* r = reinterpret_cast<unsigned*>(const_cast<float*>(ws) + NL_DEC_WS_RANGE_OFF);
    if (r[NLR_PLANE_SAT]) bits |= NL_SAT_PLANES;
    if (bits && (bits & ~*reinterpret_cast<volatile unsigned*>(r + NLR_STATUS))) atomicOr(r + NLR_STATUS, bits);
}

// Workgroup barrier that orders the workgroup's LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() also drains
// vmcnt: every barrier then waits for the global loads in flight (the next tile's input prefetch: HBM latency) and for the
// acknowledgement of earlier global stores.  Use where the waves exchange data through LDS and nothing through global memory.
__device__ __forceinline__ void nl_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif
