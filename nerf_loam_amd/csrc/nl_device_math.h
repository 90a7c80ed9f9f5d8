// nl_device_math.h -- per-ray / per-sample arithmetic of the NeRF-LOAM SDF iteration, written once
// as inline functions that are __device__ in the HIP kernels (nl_*.hip) and plain C++ in the
// host-side logic tests (tests/host_harness.cpp).  IEEE fp32, compiled with -ffp-contract=off: the
// operation ORDER below is part of the parity contract with the oracle (oracle/nl_oracle.c,
// oracle/oracle.py) and, through it, with the reference.
//
// Reference behaviour implemented (paths under /root/reference):
//   nl_slab             third_party/sparse_voxels/src/intersect_gpu.cu:77-142
//   nl_octree_walk      third_party/sparse_voxels/src/intersect_gpu.cu:225-270
//   nl_sort_cull_hits   src/variations/voxel_helpers.py:543-560
//   nl_sample_walk      third_party/sparse_voxels/src/sample_gpu.cu:165-238 (+ wrapper layout of
//                       src/variations/voxel_helpers.py:274-316 for the tail-loop quirk)
//   nl_trilinear_*      src/variations/render_helpers.py:39-70
//   nl_loss_*           src/criterion.py:59-100
//   nl_adam_*           torch/optim/adam.py::_single_tensor_adam (torch 2.10)
//   nl_rodrigues*       src/se3pose.py:24-32,64-83
//   nl_unit_dir         src/lidarFrame.py:47-52
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define NL_HD __host__ __device__ __forceinline__
#else
#define NL_HD inline
#endif

#define NL_MAX_HITS 20          // voxel_helpers.py:533 (hard-coded in the reference)
#define NL_MAX_LEVELS 24        // octree depth 18 for the 262144^3 lattice (mapping.py:82)
#define NL_FILL_DEPTH 80.0f     // voxel_helpers.py:24 MAX_DEPTH
#define NL_SAMPLER_G 200        // voxel_helpers.py:274
#define NL_SAMPLER_CHUNK 800    // voxel_helpers.py:304

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32, round-to-nearest-even (torch's c10::BFloat16 conversion)
// ---------------------------------------------------------------------------------------------
NL_HD float nl_bf16_to_f32(uint16_t b) {
    union { uint32_t u; float f; } v; v.u = ((uint32_t)b) << 16; return v.f;
}
NL_HD uint16_t nl_f32_to_bf16(float f) {
    union { uint32_t u; float f; } v; v.f = f;
    if (f != f) return (uint16_t)0x7FC0;
    uint32_t r = v.u + 0x7FFFu + ((v.u >> 16) & 1u);
    return (uint16_t)(r >> 16);
}
NL_HD float nl_round_bf16(float f) { return nl_bf16_to_f32(nl_f32_to_bf16(f)); }

// ---------------------------------------------------------------------------------------------
// counter-based sampler noise, identical to oracle.hash_noise
// ---------------------------------------------------------------------------------------------
NL_HD float nl_noise(uint32_t seed, uint32_t ray, uint32_t step) {
    uint32_t x = seed ^ (ray * 0x9E3779B1u) ^ (step * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u < 0.001f ? 0.001f : (u > 0.999f ? 0.999f : u);
}

// Exact three-term bf16 split of an fp32 value (nl_decoder.hip / nl_optim.hip: operands of the bf16 matrix cores):
// hi = v truncated to its top 8 significand bits, mid = (v - hi) truncated likewise, lo = the rest (<= 8 significant bits).
// Each term is exactly representable in bf16 (returned as its 16-bit pattern) and hi + mid + lo == v exactly.
NL_HD void nl_split3_bf16(float v, uint16_t* hi, uint16_t* mid, uint16_t* lo) {
    union { float f; uint32_t u; } a, b, r;
    a.f = v; a.u &= 0xFFFF0000u;
    b.f = v - a.f; b.u &= 0xFFFF0000u;
    r.f = (v - a.f) - b.f;
    *hi = (uint16_t)(a.u >> 16); *mid = (uint16_t)(b.u >> 16); *lo = (uint16_t)(r.u >> 16);
}

// ---------------------------------------------------------------------------------------------
// Two-term fp16 split of an fp32 value ("f16 pair": gemm modes 4 / 5 of the decoder kernels).
//   x = v * scale (a power of two: exact), saturated at +-65504;  hi = f16(x), lo = f16(x - hi), both round-to-nearest-even.
// x - hi is exact in fp32, |x - hi| <= 2^-11 |x|, and lo carries its next 11 bits: |x - (hi + lo)| <= 2^-23 |x| (one fp32 rounding of the
// operand at worst, exact for most values) as long as lo is a normal fp16 number, i.e. |x| >= 2^-2; below that the error is absolute,
// <= 2^-25 (half the fp16 subnormal spacing) - the matrix cores keep subnormal fp16 inputs (scripts/micro/f16_pair.hip).  A product of two
// fp16 values is exact in fp32 (22 significand bits), so hi*hi' + hi*lo' + lo*hi' (+ lo*lo') accumulated in fp32 is the fp32 product of
// the operands up to 2^-22 (2^-23 with the fourth product) - below the rounding of the 256-deep fp32 accumulation it enters.
// The scales put the operands of the shipped decoder high in fp16's range and bound what saturates:
//   X * 2^6 (|X| < 1023.5; with a trainable decoder |X| < 255.9: the dW1 operand U = sigma dsdf 2^4 X with sigma dsdf in [8, 16)), W1 * 2^8 (|W1| < 255.9),
//   H1 * 2^4 (H1 < 4094), W2 * 2^8 (|W2| < 255.9), w3_j W2[j][k] * 2^10 = NL_F16_SG (|w3 W2| < 63.97; the dgrad accumulators, sums of up to 256 of them, < 63.97 too).
// A value beyond its range is CLIPPED (v_med3, which also turns a NaN into a finite value where the reference's fp32 decoder would propagate it) - and reported:
// the decoder kernels raise NL_SAT_* bits in the weight workspace's status word (nl_common.h nl_range_check, round 6), the optimiser latches them into the call
// status, the API raises.  Gemm mode 3 (exact three-term bf16 splits, no scaling, no range) is the fallback.
// These software conversions are what the weight-plane kernels (nl_optim.hip) and the host tests use; the decoder kernels convert with
// v_cvt_pk_f16_f32 / v_cvt_f32_f16 (same IEEE results).
// ---------------------------------------------------------------------------------------------
#define NL_F16_MAX 65504.0f
#define NL_F16_SX 64.0f
#define NL_F16_SW1 256.0f
#define NL_F16_SH 16.0f
#define NL_F16_SW2 256.0f
#define NL_F16_SG 1024.0f
NL_HD uint16_t nl_f32_to_f16(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    const uint32_t sign = (v.u >> 16) & 0x8000u, a = v.u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (a > 0x7F800000u ? 0x200u : 0u));      // inf / NaN
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                         // >= 65520 rounds to inf
    if (a < 0x33000001u) return (uint16_t)sign;                                                      // <= 2^-25 rounds to zero (the tie goes to even)
    const int e = (int)(a >> 23) - 127;
    const uint32_t m = (a & 0x7FFFFFu) | 0x800000u;                                                  // 24-bit significand
    const int shift = e < -14 ? 13 + (-14 - e) : 13;                                                 // bits below the fp16 result (subnormal: more)
    uint32_t kept = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (kept & 1u))) ++kept;
    return (uint16_t)(sign | (e < -14 ? kept : ((uint32_t)(e + 15) << 10) + (kept - 0x400u)));       // a rounding carry moves into the exponent
}
NL_HD float nl_f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    union { uint32_t u; float f; } v;
    if (e == 31u) v.u = sign | 0x7F800000u | (m << 13);
    else if (e) v.u = sign | ((e + 112u) << 23) | (m << 13);
    else { v.f = (float)m * 5.9604644775390625e-08f; v.u |= sign; }                                  // m * 2^-24, exact
    return v.f;
}
NL_HD void nl_split2_f16(float v, float scale, uint16_t* hi, uint16_t* lo) {
    float x = v * scale;
    x = x > NL_F16_MAX ? NL_F16_MAX : (x < -NL_F16_MAX ? -NL_F16_MAX : x);
    const uint16_t h = nl_f32_to_f16(x);
    *hi = h; *lo = nl_f32_to_f16(x - nl_f16_to_f32(h));
}

// ray-selection key (nl_select.hip): lowbias32 is a bijection on 32-bit integers, so keys of distinct rays never tie
NL_HD uint32_t nl_select_key(uint32_t seed, uint32_t i) {
    uint32_t x = i ^ (seed * 0x9E3779B9u + 0x7F4A7C15u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// unit direction of a LiDAR return (src/lidarFrame.py:47-52: rays_norm = ||p||_2 + 1e-8, rays_d = p / rays_norm).  The reference
// evaluates it with torch on the host, whose vector-norm kernel accumulates the squares with fused multiply-adds
// (s = x x; s = fma(y, y, s); s = fma(z, z, s) - measured: 400 000 random points, 0 mismatches; the unfused and the fp64 forms differ on
// 10-15 % of them); square root and division are IEEE.  tests/test_device_math_host.py pins this against torch, bit for bit.
NL_HD float nl_unit_dir(float x, float y, float z, float* dx, float* dy, float* dz) {
    float s = x * x;
    s = fmaf(y, y, s);
    s = fmaf(z, z, s);
    const float n = sqrtf(s) + 1e-8f;
    *dx = x / n; *dy = y / n; *dz = z / n;
    return n;
}

// ---------------------------------------------------------------------------------------------
// ray / cube slab test.  Returns true on hit.
// ---------------------------------------------------------------------------------------------
NL_HD bool nl_slab(float ox, float oy, float oz, float dx, float dy, float dz,
                   float cx, float cy, float cz, float half, float* tn, float* tf) {
    float lo = 0.0f, hi = 100000.0f;
    const float o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz}, c[3] = {cx, cy, cz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float inv = 1.0f / d[a];
        float t0 = (c[a] - half - o[a]) * inv;
        float t1 = (c[a] + half - o[a]) * inv;
        if (t1 < t0) { float t = t0; t0 = t1; t1 = t; }
        if (t1 < lo) return false;
        if (t0 > hi) return false;
        lo = (t0 > lo) ? t0 : lo;
        hi = (t1 < hi) ? t1 : hi;
        if (lo > hi) return false;
    }
    *tn = lo; *tf = hi;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Octree DFS.  The reference pushes the existing children 0..7 of a hit interior node on a stack
// and pops the last one first; that is a depth-first walk visiting children in DESCENDING octant
// order.  It is restated with one (node, next-octant) cursor per level - at most tree-depth+1
// entries instead of 256 - which visits exactly the same nodes in the same order, so the <=n_max
// recorded leaves (and their order) are identical.
//   centres[n,3], structure[n,9] = reference layouts (voxel_center_xyz / voxel_structure)
//   out_idx/out_t0/out_t1: n_max entries owned by the calling thread; returns the hit count.
// StackT lets the kernel keep the cursors in LDS and the host harness in a plain array.
// ---------------------------------------------------------------------------------------------
template <typename StackT>
NL_HD int nl_octree_walk(const float* __restrict__ centres, const int* __restrict__ structure,
                         float ox, float oy, float oz, float dx, float dy, float dz,
                         float half_voxel, int n_max, StackT& stk,
                         int* out_idx, float* out_t0, float* out_t1) {
    int cnt = 0, lvl = 0;
    // visit(root)
    {
        float tn, tf;
        const int side = structure[8];
        if (!nl_slab(ox, oy, oz, dx, dy, dz, centres[0], centres[1], centres[2], half_voxel * (float)side, &tn, &tf))
            return 0;
        if (side == 1) { out_idx[0] = 0; out_t0[0] = tn; out_t1[0] = tf; return 1; }
        stk.set(0, 0, 7);
    }
    while (lvl >= 0 && cnt < n_max) {
        const int node = stk.node(lvl);
        int u = stk.cursor(lvl);
        int child = -1;
        while (u >= 0) {                       // next existing child, descending octant
            child = structure[node * 9 + u];
            --u;
            if (child > -1) break;
            child = -1;
        }
        stk.set_cursor(lvl, u);
        if (child < 0) { --lvl; continue; }    // node exhausted
        float tn, tf;
        const int side = structure[child * 9 + 8];
        if (!nl_slab(ox, oy, oz, dx, dy, dz, centres[child * 3], centres[child * 3 + 1], centres[child * 3 + 2],
                     half_voxel * (float)side, &tn, &tf))
            continue;
        if (side == 1) {
            out_idx[cnt] = child; out_t0[cnt] = tn; out_t1[cnt] = tf; ++cnt;
            continue;
        }
        ++lvl;
        stk.set(lvl, child, 7);
    }
    return cnt;
}

struct NlLocalStack {           // host harness / fallback: cursors in a private array
    int n[NL_MAX_LEVELS]; int c[NL_MAX_LEVELS];
    NL_HD void set(int l, int node, int cur) { n[l] = node; c[l] = cur; }
    NL_HD int node(int l) const { return n[l]; }
    NL_HD int cursor(int l) const { return c[l]; }
    NL_HD void set_cursor(int l, int cur) { c[l] = cur; }
};

// ---------------------------------------------------------------------------------------------
// ray_intersect post-processing on one ray's <=20 hits (registers / private arrays):
// invalid -> max_distance, STABLE sort by t_min (ties keep DFS order; the reference's torch.sort
// leaves tie order unspecified), cull t_max > 2*max_distance or t_min > max_distance.
// Returns the number of valid hits (valid hits form a prefix).
// ---------------------------------------------------------------------------------------------
NL_HD int nl_sort_cull_hits(int cnt, int* idx, float* t0, float* t1, float max_distance) {
    for (int i = 1; i < cnt; ++i) {            // insertion sort, stable
        int ki = idx[i]; float a = t0[i], b = t1[i];
        int j = i - 1;
        while (j >= 0 && t0[j] > a) { idx[j + 1] = idx[j]; t0[j + 1] = t0[j]; t1[j + 1] = t1[j]; --j; }
        idx[j + 1] = ki; t0[j + 1] = a; t1[j + 1] = b;
    }
    int valid = 0;
    for (int i = 0; i < cnt; ++i) {
        bool keep = !(t1[i] > 2.0f * max_distance) && !(t0[i] > max_distance);
        if (keep) ++valid; else { idx[i] = -1; t0[i] = max_distance; t1[i] = max_distance; }
    }
    // culled entries always form a suffix (t_min sorted, and t_max > 2*md implies t_min > md for
    // any voxel of side <= md); keep the reference's per-element semantics regardless
    return valid;
}

// ---------------------------------------------------------------------------------------------
// Inverse-CDF sampler for one ray.
//   hits: idx/t0/t1[0..P) with P = the batch-wide max hit count (entries >= own count are
//         (-1, max_distance, max_distance) like the reference's trimmed tensors)
//   step_size_m: sampling step in metres; noise(step) in [0.001,0.999]
//   Tail quirk (sample_gpu.cu:224-237): the closing loop runs only while
//         rays_in_row > j_in_row*P + curr_bin   and stops when row_first_idx[curr_bin] == -1,
//         where row_first_idx is the hit list of the FIRST ray of this ray's batch row.
//         tail_always = true selects the "fixed" behaviour (always close the last interval).
//   emit(s, voxel, depth, dist) is called for each sample in order; returns the sample count.
// ---------------------------------------------------------------------------------------------
struct NlTailCtx {
    int rays_in_row;            // m   = rays per row in the launched chunk
    int j_in_row;               // j   = this ray's index inside its row chunk
    const int* row_first_idx;   // hit list (stride 1) of the row's first ray in the chunk
    int row_first_count;        // number of valid entries in that list (entries beyond it read as -1)
    int row_first_bias;         // the list stores idx + bias (1 for the table nl_dist_x1_merge derives from the gathered hit counts: 2 below the count, 0 beyond)
    bool tail_always;
};

// Core walk over accessor functors (idx(b), t0(b), t1(b) for 0 <= b < P) with the interval-length total `tot` supplied by
// the caller (sequential fp32 sum over b = 0..P-1 of (idx == -1 ? 0 : t1 - t0)).  The fused GPU sampler keeps a ray's hit
// list in LDS and passes accessors for it: a per-lane array indexed with a run-time bin index lives in scratch memory, and
// its ~700-cycle latency sits on the walk's critical path at every interval change of ANY lane of the wave.
template <typename GetI, typename GetF0, typename GetF1, typename NoiseF, typename EmitF>
NL_HD int nl_sample_walk_core(GetI idx, GetF0 t0, GetF1 t1, int P, float tot, float step_size_m,
                              const NlTailCtx& tc, NoiseF noise, EmitF emit) {
    const float steps = tot / step_size_m;
    int curr_bin = 0, s = 0;
    int curr_idx = idx(0);
    float curr_min_depth = t0(0), curr_max_depth = t1(0);
    float curr_min_cdf = 0.0f;
    float curr_max_cdf = ((curr_idx == -1) ? 0.0f : (curr_max_depth - curr_min_depth)) / tot;
    const float step = (float)(1.0 / (double)steps);
    float z_low = curr_min_depth;
    const int total_steps = (int)ceilf(steps);
    bool done = false;
    for (int cs = 0; cs < total_steps; ++cs) {
        float curr_cdf = ((float)cs + noise(cs)) * step;
        while (curr_cdf > curr_max_cdf) {
            emit(s, curr_idx, (curr_max_depth + z_low) * 0.5f, curr_max_depth - z_low);
            ++curr_bin; ++s;
            if (curr_bin >= P) { done = true; break; }
            curr_idx = idx(curr_bin);
            if (curr_idx == -1) { done = true; break; }
            curr_min_depth = t0(curr_bin); curr_max_depth = t1(curr_bin);
            curr_min_cdf = curr_max_cdf;
            curr_max_cdf = curr_max_cdf + (curr_max_depth - curr_min_depth) / tot;
            z_low = curr_min_depth;
        }
        if (done) break;
        float u = (curr_cdf - curr_min_cdf) / (curr_max_cdf - curr_min_cdf);
        float z = curr_min_depth + u * (curr_max_depth - curr_min_depth);
        emit(s, curr_idx, (z + z_low) * 0.5f, z - z_low);
        z_low = z;
        ++s;
    }
    while (z_low < curr_max_depth && !done &&
           (tc.tail_always || tc.rays_in_row > tc.j_in_row * P + curr_bin)) {
        emit(s, curr_idx, (curr_max_depth + z_low) * 0.5f, curr_max_depth - z_low);
        ++curr_bin; ++s;
        if (curr_bin >= P) break;
        curr_idx = idx(curr_bin);
        if ((tc.tail_always ? curr_idx : (curr_bin < tc.row_first_count ? tc.row_first_idx[curr_bin] - tc.row_first_bias : -1)) == -1) break;
        curr_min_depth = t0(curr_bin); curr_max_depth = t1(curr_bin);
        z_low = curr_min_depth;
    }
    return s;
}

template <typename NoiseF, typename EmitF>
NL_HD int nl_sample_walk(const int* idx, const float* t0, const float* t1, int P, float step_size_m,
                         const NlTailCtx& tc, NoiseF noise, EmitF emit) {
    // ray_sample(): dists, probs, steps      (voxel_helpers.py:572-577), sequential fp32 sums
    float tot = 0.0f;
    for (int h = 0; h < P; ++h) { float d = (idx[h] == -1) ? 0.0f : (t1[h] - t0[h]); tot = tot + d; }
    return nl_sample_walk_core([&](int b) { return idx[b]; }, [&](int b) { return t0[b]; }, [&](int b) { return t1[b]; },
                               P, tot, step_size_m, tc, noise, emit);
}

// ---------------------------------------------------------------------------------------------
// Step-parallel form of the same sampler (same arithmetic, same sample order): the sequential walk advances a bin cursor
// while it visits the stratified steps cs = 0..T-1; because (cs + noise) * step increases with cs and the interval CDF
// boundaries cum[b] are non-decreasing, the bin of step cs depends on cs alone - bin(cs) = first b with !(cdf > cum[b]) -
// so every step can be evaluated independently (one GPU lane per step) and the samples land at known indices:
//   own sample of step cs            -> index cs + bin(cs)
//   closing sample of interval q     -> index cs + q, emitted by the first step cs that lies beyond interval q
// (before it come cs own samples and q closing samples).  cum[] is the sequential fp32 chain of the walk
// (cum[0] = len_0 / tot, cum[b] = cum[b-1] + len_b / tot); nb = number of leading usable intervals (idx != -1, b < P).
//   nl_walk_plan    fills cum[0..nb) and returns nb
//   nl_walk_eval    bin and depth of one step
//   nl_walk_step    emits everything step cs is responsible for; returns 1 if it produced an own sample
//   nl_walk_tail    the closing loop after the last step (sequential, a few iterations) ; returns the final sample count
// ---------------------------------------------------------------------------------------------
template <typename GetI, typename GetF0, typename GetF1, typename PutC>
NL_HD int nl_walk_plan(GetI idx, GetF0 t0, GetF1 t1, int P, float tot, PutC put_cum) {
    int nb = 0;
    float c = 0.0f;
    for (int b = 0; b < P; ++b) {
        const int i = idx(b);
        if (b > 0 && i == -1) break;                       // the walk stops at the first invalid interval after the first
        const float len = (i == -1) ? 0.0f : (t1(b) - t0(b));
        c = (b == 0) ? (len / tot) : (c + len / tot);
        put_cum(b, c);
        ++nb;
        if (i == -1) break;
    }
    return nb;
}

template <typename GetC, typename GetF0, typename GetF1>
NL_HD void nl_walk_eval(int cs, float noise_cs, float step, int nb, GetC cum, GetF0 t0, GetF1 t1, int* bin, float* z) {
    const float cdf = ((float)cs + noise_cs) * step;
    int b = 0;
    while (b < nb && cdf > cum(b)) ++b;
    *bin = b;
    *z = 0.0f;
    if (b < nb) {
        const float lo = (b > 0) ? cum(b - 1) : 0.0f, hi = cum(b);
        const float u = (cdf - lo) / (hi - lo);
        const float d0 = t0(b), d1 = t1(b);
        *z = d0 + u * (d1 - d0);
    }
}

// nl_walk_step after its two evaluations: (bp_e, zp_e) = nl_walk_eval(cs - 1) (not looked at for cs == 0), (b, z) = nl_walk_eval(cs).
// The GPU sampler hands a step's evaluation to the lane of the next step instead of evaluating every step twice.
template <typename GetI, typename GetF0, typename GetF1, typename EmitF>
NL_HD int nl_walk_step_from(int cs, int bp_e, float zp_e, int b, float z, int nb, GetI idx, GetF0 t0, GetF1 t1, EmitF emit) {
    int bp = 0; float zl = t0(0);
    if (cs > 0) {
        if (bp_e >= nb) return 0;                            // the walk ended at an earlier step
        bp = bp_e; zl = zp_e;
    }
    for (int q = bp; q < b; ++q) {                           // closing samples of the intervals this step leaves behind
        const float d1 = t1(q);
        emit(cs + q, idx(q), (d1 + zl) * 0.5f, d1 - zl);
        if (q + 1 < nb) zl = t0(q + 1);
    }
    if (b >= nb) return 0;
    emit(cs + b, idx(b), (z + zl) * 0.5f, z - zl);
    return 1;
}

template <typename GetI, typename GetC, typename GetF0, typename GetF1, typename NoiseF, typename EmitF>
NL_HD int nl_walk_step(int cs, float step, int nb, GetI idx, GetC cum, GetF0 t0, GetF1 t1, NoiseF noise, EmitF emit) {
    int bp = 0; float zp = 0.0f;
    if (cs > 0) nl_walk_eval(cs - 1, noise(cs - 1), step, nb, cum, t0, t1, &bp, &zp);
    int b; float z;
    nl_walk_eval(cs, noise(cs), step, nb, cum, t0, t1, &b, &z);
    return nl_walk_step_from(cs, bp, zp, b, z, nb, idx, t0, t1, emit);
}

// nl_walk_tail after the evaluation of the last step: (bin_e, z_e) = eval(T - 1) (not looked at for T == 0); eval(cs, &bin, &z)
// evaluates a step (the search for the step at which a ray ran out of intervals)
template <typename GetI, typename GetF0, typename GetF1, typename EvalF, typename EmitF>
NL_HD int nl_walk_tail_from(int T, int bin_e, float z_e, int nb, int P, GetI idx, GetF0 t0, GetF1 t1, const NlTailCtx& tc, EvalF eval, EmitF emit) {
    int bin = 0, s = 0; float zl = t0(0);
    if (T > 0) {
        bin = bin_e; zl = z_e;
        if (bin >= nb) {                                     // ran out of intervals during the steps: own samples = steps before the end
            int lo = 0, hi = T - 1;                          // first step whose bin is nb (bins are monotone in cs)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1; int bm; float zm;
                eval(mid, &bm, &zm);
                if (bm >= nb) hi = mid; else lo = mid + 1;
            }
            return lo + nb;
        }
        s = T + bin;
    }
    float curr_max_depth = t1(bin);
    int curr_idx = idx(bin);
    while (zl < curr_max_depth && (tc.tail_always || tc.rays_in_row > tc.j_in_row * P + bin)) {
        emit(s, curr_idx, (curr_max_depth + zl) * 0.5f, curr_max_depth - zl);
        ++bin; ++s;
        if (bin >= P) break;
        curr_idx = idx(bin);
        if ((tc.tail_always ? curr_idx : (bin < tc.row_first_count ? tc.row_first_idx[bin] - tc.row_first_bias : -1)) == -1) break;
        zl = t0(bin); curr_max_depth = t1(bin);
    }
    return s;
}

template <typename GetI, typename GetC, typename GetF0, typename GetF1, typename NoiseF, typename EmitF>
NL_HD int nl_walk_tail(int T, float step, int nb, int P, GetI idx, GetC cum, GetF0 t0, GetF1 t1, const NlTailCtx& tc, NoiseF noise, EmitF emit) {
    auto eval = [&](int cs, int* bin, float* z) { nl_walk_eval(cs, noise(cs), step, nb, cum, t0, t1, bin, z); };
    int bin = 0; float zl = 0.0f;
    if (T > 0) eval(T - 1, &bin, &zl);
    return nl_walk_tail_from(T, bin, zl, nb, P, idx, t0, t1, tc, eval, emit);
}

// host / single-thread driver of the step-parallel form: same results and emit order as nl_sample_walk
template <typename NoiseF, typename EmitF>
NL_HD int nl_sample_walk_steps(const int* idx, const float* t0, const float* t1, int P, float step_size_m,
                               const NlTailCtx& tc, NoiseF noise, EmitF emit) {
    float tot = 0.0f;
    for (int h = 0; h < P; ++h) { float d = (idx[h] == -1) ? 0.0f : (t1[h] - t0[h]); tot = tot + d; }
    float cum[NL_MAX_HITS];
    auto gi = [&](int b) { return idx[b]; };
    auto g0 = [&](int b) { return t0[b]; };
    auto g1 = [&](int b) { return t1[b]; };
    auto gc = [&](int b) { return cum[b]; };
    const int nb = nl_walk_plan(gi, g0, g1, P < NL_MAX_HITS ? P : NL_MAX_HITS, tot, [&](int b, float c) { cum[b] = c; });
    const float steps = tot / step_size_m;
    const float step = (float)(1.0 / (double)steps);
    const int T = (int)ceilf(steps);
    for (int cs = 0; cs < T; ++cs) nl_walk_step(cs, step, nb, gi, gc, g0, g1, noise, emit);
    return nl_walk_tail(T, step, nb, P, gi, gc, g0, g1, tc, noise, emit);
}

// position of hit-ray r (0-based rank among the R hit rays) in the reference wrapper's padded
// [200, L, P] layout, chunked by 800 along dim 1 (voxel_helpers.py:274-316)
NL_HD void nl_sampler_layout(int r, int R, int* j_in_row, int* rays_in_row, int* row_first_rank) {
    const int L = (R + NL_SAMPLER_G - 1) / NL_SAMPLER_G;
    const int g = r / L, l = r - g * L;
    const int c0 = (l / NL_SAMPLER_CHUNK) * NL_SAMPLER_CHUNK;
    *j_in_row = l - c0;
    const int rem = L - c0;
    *rays_in_row = rem < NL_SAMPLER_CHUNK ? rem : NL_SAMPLER_CHUNK;
    int first = g * L + c0;
    *row_first_rank = first < R ? first : 0;     // padding rows replicate hit-ray 0
}

// index of a row-first hit-rank in the table of nl_dist_x1_merge: [batch row 200][chunk ceil(L / 800)]
NL_HD int nl_row_first_entry(int first_rank, int R) {
    const int L = (R + NL_SAMPLER_G - 1) / NL_SAMPLER_G;
    const int nch = (L + NL_SAMPLER_CHUNK - 1) / NL_SAMPLER_CHUNK;
    const int g = first_rank / L, c = (first_rank - g * L) / NL_SAMPLER_CHUNK;
    return g * nch + c;
}

// ---------------------------------------------------------------------------------------------
// trilinear weights: p = (x - c)/vs + 0.5 ; corner k = 4qx + 2qy + qz ; w_k = (tx*ty)*tz
// ---------------------------------------------------------------------------------------------
NL_HD void nl_trilinear_p(const float x[3], const float c[3], float vs, float p[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = (x[a] - c[a]) / vs + 0.5f;
}
NL_HD void nl_trilinear_w(const float p[3], float w[8]) {
    const float q[3] = {1.0f - p[0], 1.0f - p[1], 1.0f - p[2]};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float tx = (k & 4) ? p[0] : q[0];
        float ty = (k & 2) ? p[1] : q[1];
        float tz = (k & 1) ? p[2] : q[2];
        w[k] = (tx * ty) * tz;
    }
}
// d(sum_k w_k * dot_k)/dp, dot_k = <e_k, dfeat>
NL_HD void nl_trilinear_dp(const float p[3], const float dot[8], float dp[3]) {
    const float q[3] = {1.0f - p[0], 1.0f - p[1], 1.0f - p[2]};
    dp[0] = dp[1] = dp[2] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float tx = (k & 4) ? p[0] : q[0];
        float ty = (k & 2) ? p[1] : q[1];
        float tz = (k & 1) ? p[2] : q[2];
        dp[0] += ((k & 4) ? 1.0f : -1.0f) * (ty * tz) * dot[k];
        dp[1] += ((k & 2) ? 1.0f : -1.0f) * (tx * tz) * dot[k];
        dp[2] += ((k & 1) ? 1.0f : -1.0f) * (tx * ty) * dot[k];
    }
}

// ---------------------------------------------------------------------------------------------
// loss.  Per-sample classification needs only geometry (z = depth*cos, d = ||p_gt||*cos):
//   front: z < d - tau ; back: z > d + tau ; sdfm = !front & !back & (0 < d < max_depth)
// ---------------------------------------------------------------------------------------------
struct NlLossScalars {          // filled once per iteration from the global counters
    float w_fs, w_sdf;          // 1 - n_fs/(n_fs+n_sdf), 1 - n_sdf/(n_fs+n_sdf)
    float two_over_n;           // 2 / (R * S_max)
    float inv_n;                // 1 / (R * S_max)
    float fs_weight, sdf_weight, tau, max_depth;
    int   R, S_max, P;          // hit rays, max samples per ray, valid samples
    unsigned ds_max_bits;       // bit pattern of max |dL/dsdf| over the iteration's samples: zeroed by the finalize, raised by the decoder kernel (train),
                                // read by the fp16-pair dW2 kernel to place dsdf_i * H1[i][k] in fp16's range
};
NL_HD void nl_loss_masks(float z, float d, float tau, float max_depth, bool* front, bool* sdfm) {
    const bool f = z < (d - tau);
    const bool b = z > (d + tau);
    *front = f;
    *sdfm = (!f) && (!b) && (d > 0.0f) && (d < max_depth);
}
// dL/dsdf for a VALID sample, and its two squared residuals (for the loss value)
NL_HD float nl_loss_grad(float sdf, float z, float d, bool front, bool sdfm, const NlLossScalars& ls,
                         float* r_fs_sq, float* r_sdf_sq) {
    const float f = front ? 1.0f : 0.0f, m = sdfm ? 1.0f : 0.0f;
    const float r_fs = sdf * f - f;
    const float r_sdf = (z + sdf * ls.tau) * m - d * m;
    *r_fs_sq = r_fs * r_fs;
    *r_sdf_sq = r_sdf * r_sdf;
    return ls.fs_weight * ls.w_fs * ls.two_over_n * r_fs * f +
           ls.sdf_weight * ls.w_sdf * ls.two_over_n * r_sdf * ls.tau * m;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch 2.10 _single_tensor_adam, default branch).  bf16 variant rounds at the same seven
// points as the bf16 tensors of the reference (parameter, grad, exp_avg, exp_avg_sq are bf16).
// ---------------------------------------------------------------------------------------------
struct NlAdamHyper { float lr_over_bc1; float bc2_sqrt; float beta1; float beta2; float eps; float one_m_b1; float one_m_b2; };
NL_HD NlAdamHyper nl_adam_hyper(double lr, int step, double beta1, double beta2, double eps) {
    NlAdamHyper h;
    double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    h.lr_over_bc1 = (float)(-(lr / bc1));       // value = -step_size
    h.bc2_sqrt = (float)sqrt(bc2);
    h.beta1 = (float)beta1; h.beta2 = (float)beta2; h.eps = (float)eps;
    h.one_m_b1 = (float)(1.0 - beta1); h.one_m_b2 = (float)(1.0 - beta2);
    return h;
}
NL_HD void nl_adam_f32(float* p, float g, float* m, float* v, const NlAdamHyper& h) {
    float mm = *m + h.one_m_b1 * (g - *m);
    float vv = *v * h.beta2;
    vv = vv + (h.one_m_b2 * g) * g;
    float den = sqrtf(vv) / h.bc2_sqrt + h.eps;
    *p = *p + (h.lr_over_bc1 * mm) / den;
    *m = mm; *v = vv;
}
NL_HD void nl_adam_bf16(uint16_t* p, uint16_t g, uint16_t* m, uint16_t* v, const NlAdamHyper& h) {
    const float gf = nl_bf16_to_f32(g);
    float mm = nl_round_bf16(nl_bf16_to_f32(*m) + h.one_m_b1 * (gf - nl_bf16_to_f32(*m)));
    float vv = nl_round_bf16(nl_bf16_to_f32(*v) * h.beta2);
    vv = nl_round_bf16(vv + (h.one_m_b2 * gf) * gf);
    float den = nl_round_bf16(sqrtf(vv));
    den = nl_round_bf16(den / h.bc2_sqrt);
    den = nl_round_bf16(den + h.eps);
    float pp = nl_round_bf16(nl_bf16_to_f32(*p) + (h.lr_over_bc1 * mm) / den);
    *p = nl_f32_to_bf16(pp); *m = nl_f32_to_bf16(mm); *v = nl_f32_to_bf16(vv);
}

// ---------------------------------------------------------------------------------------------
// SE3: R = I + A(theta) [w]x + B(theta) [w]x^2, 11-term Taylor A = sin(t)/t, B = (1-cos t)/t^2
// ---------------------------------------------------------------------------------------------
NL_HD void nl_taylor_ab(float x, float* A, float* dA, float* B, float* dB) {
    float a = 0.0f, da = 0.0f, b = 0.0f, db = 0.0f;
    double denA = 1.0, denB = 1.0;
    float xp = 1.0f;            // x^(2i)
    float xpm1 = 0.0f;          // x^(2i-1)
    for (int i = 0; i <= 10; ++i) {
        if (i > 0) { denA *= (double)((2 * i) * (2 * i + 1)); xpm1 = (i == 1) ? x : xpm1 * x * x; xp = xp * x * x; }
        denB *= (double)((2 * i + 1) * (2 * i + 2));
        const float sgn = (i & 1) ? -1.0f : 1.0f;
        a = a + (sgn * xp) / (float)denA;
        b = b + (sgn * xp) / (float)denB;
        if (i > 0) {
            da = da + (sgn * (float)(2 * i) * xpm1) / (float)denA;
            db = db + (sgn * (float)(2 * i) * xpm1) / (float)denB;
        }
    }
    *A = a; *dA = da; *B = b; *dB = db;
}
NL_HD void nl_skew(const float w[3], float W[9]) {
    W[0] = 0.0f;  W[1] = -w[2]; W[2] = w[1];
    W[3] = w[2];  W[4] = 0.0f;  W[5] = -w[0];
    W[6] = -w[1]; W[7] = w[0];  W[8] = 0.0f;
}
NL_HD void nl_mat3_mul(const float a[9], const float b[9], float c[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}
NL_HD void nl_rodrigues(const float w[3], float R[9]) {
    float W[9], W2[9], A, dA, B, dB;
    nl_skew(w, W);
    nl_mat3_mul(W, W, W2);
    const float th = sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    nl_taylor_ab(th, &A, &dA, &B, &dB);
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + A * W[i] + B * W2[i];
}
// dL/dw given G = dL/dR (row-major 3x3)
NL_HD void nl_rodrigues_bwd(const float w[3], const float G[9], float gw[3]) {
    float W[9], W2[9], A, dA, B, dB;
    nl_skew(w, W);
    nl_mat3_mul(W, W, W2);
    const float th = sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    nl_taylor_ab(th, &A, &dA, &B, &dB);
    float gA = 0.0f, gB = 0.0f;
    for (int i = 0; i < 9; ++i) { gA += G[i] * W[i]; gB += G[i] * W2[i]; }
    // gW = A*G + B*(G W^T + W^T G)
    float Wt[9], t1[9], t2[9], gW[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Wt[i * 3 + j] = W[j * 3 + i];
    nl_mat3_mul(G, Wt, t1);
    nl_mat3_mul(Wt, G, t2);
    for (int i = 0; i < 9; ++i) gW[i] = A * G[i] + B * (t1[i] + t2[i]);
    const float gth = gA * dA + gB * dB;
    gw[0] = gW[7] - gW[5];
    gw[1] = gW[2] - gW[6];
    gw[2] = gW[3] - gW[1];
    if (th > 0.0f) { for (int a = 0; a < 3; ++a) gw[a] += gth * (w[a] / th); }
}
