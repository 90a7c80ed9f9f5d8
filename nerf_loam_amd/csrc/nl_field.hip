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
#include <atomic>
#include "../../include/nerfloam_hip.h"

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
    double* g_pose;                 // [F,12] (dt[3], dR[9]) fp64 accumulators (backward)
    int n_frames;
    int want_emb_grad;
    int want_pose_grad;
    long long* dbg;                 // optional [blocks][8] s_memtime stamps of thread 0 (profiling aid, k_trilinear_bwd)
    int resident_blocks;            // k_trilinear_bwd: workgroups the device holds at once (4 per compute unit)
    int probes;                     // k_trilinear_bwd: open-addressing probes before a run goes straight to memory
    int flush_steps_left;           // k_trilinear_bwd: a full table is written out mid-span if the groups have at least this many sample steps left
    NlTouchedDev touched;           // optional: rows whose accumulators receive a contribution are recorded (nl_touch_row)
};
#define FSTAMP(k) do { if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)

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

// get_scores (render_helpers.py:96-153) without the point tensor: the res^3 grid points of voxels [vox0, vox0 + n_vox) are generated here - point
// (ix, iy, iz) of voxel v is lin[i] * voxel_size + centre_v per axis, the reference's two fp32 operations (`sampled_xyz *= voxel_size`, `+ points`), with
// lin = torch.linspace(-0.5, 0.5, res) handed over by the caller (torch's own values) - and go through k_gather_points' arithmetic.  Saves the 12 + 4 bytes per
// point of the xyz / voxel-id tensors (and the torch kernels that built them): a point costs its 64-byte X row and nothing else.
__global__ __launch_bounds__(NL_FIELD_THREADS) void k_gather_grid(long long P, int vox0, int res, const float* __restrict__ lin,
                                                                  const float* __restrict__ centres, const int* __restrict__ vertex_rows,
                                                                  const uint16_t* __restrict__ emb, float voxel_size, float* __restrict__ X)
{
    const int half = threadIdx.x & 1;
    const int r3 = res * res * res;
    for (long long s = ((long long)blockIdx.x * NL_FIELD_THREADS + threadIdx.x) >> 1; s < P; s += ((long long)gridDim.x * NL_FIELD_THREADS) >> 1) {
        const int v = vox0 + (int)(s / r3), q = (int)(s % r3);
        const int ix = q / (res * res), iy = (q / res) % res, iz = q % res;
        float x[3], c[3], p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) c[i] = centres[3 * (size_t)v + i];
        const float ox = lin[ix] * voxel_size, oy = lin[iy] * voxel_size, oz = lin[iz] * voxel_size;
        x[0] = ox + c[0]; x[1] = oy + c[1]; x[2] = oz + c[2];
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
// The scatter is the hot spot (8 rows x 16 channels per sample = 140 M fp32 adds for one 64x2048 scan).  Samples are
// packed in ray order, so consecutive samples mostly stay in one voxel and a span of consecutive samples touches few
// distinct vertex rows.  Layout: EIGHT lanes per sample, one per voxel corner - every lane owns one embedding row at a
// time: its 16-channel contribution accumulates in registers over the voxel run, the row id and the bf16 row (pose
// gradient) are loaded once per run, and the lanes of a wave do uniform work.  A group of 8 lanes walks ~32 consecutive
// samples; finished runs go into an LDS hash table (open addressing on the row id) that is PRIVATE TO THE WAVE and
// updated WITHOUT float atomics: LDS executes a wave's instructions in order, the 8 corner rows of one voxel are
// distinct, so adding one 8-lane group at a time (read 16 floats, add, write back) is race-free.  ds_add_f32 retires
// about one lane per cycle per CU: the atomic version of this kernel spent 220 of its 380 us in them (measured by
// swapping them for plain read-modify-writes).  Every touched row leaves the table once per span with global
// atomics; table overflow falls back to direct global atomics.
#define TB_WAVES (NL_FIELD_THREADS / 64)
#define TB_SLOTS 128                                    // per wave
#define TB_PROBES 16
#define TB_FLUSH_MIN_STEPS_LEFT 2                       // ... and the 8-lane groups have at least this many sample steps of their span left
#define TB_FLUSH_OCCUPANCY 96                          // a run without a slot: the wave writes its table out if at least this many of the 128 slots are taken
#define TB_GROUPS (NL_FIELD_THREADS / 8)                 // 8-lane groups per workgroup
#define TB_MIN_SPAN 256                                 // samples per workgroup: at least this many (aggregation), else P / grid
#define TB_SMALL_P 524288                               // below: the kernel is a latency chain over a group's samples, not atomics-bound:
#define TB_MIN_SPAN_SMALL 64                            // shorter spans (2 samples per 8-lane group instead of 8; 1 measured worse).
                                                        // 2048 rays + embeddings: iteration 0.217 -> 0.184 ms; 4096 x 4: 0.371 -> 0.341
#define TB_LONG_RAY_SAMPLES 20                          // samples per hit ray from which a small launch takes ...
#define TB_MIN_SPAN_LONG_RAYS 192                       // ... 6 samples per 8-lane group
#define TB_ONE_ROUND_SPAN 6                             // the kernel is a latency chain per wave (~4 waves per SIMD resident): as long as ONE round of
                                                        // resident workgroups covers P with at most this many samples per 8-lane group, the workgroups of a
                                                        // second round leave at once.  scripts/scatter_sweep.py, 1024 against 2048 workgroups, single-scan map:
                                                        // 8192 rays 26.8 / 35.9 us, 16 384 (a rank's share) 39.6 / 45.1; 150-scan map (a ray crosses ~15 voxels
                                                        // with ~1.5 samples each: a wave's 128-slot table overflows early): 4096 rays 125 / 145, 8192 rays 177 / 238,
                                                        // but 16 384 rays (12 samples per group in one round) 479 / 316 - hence 6, not the 18 the single-scan map
                                                        // alone would allow (32 768 rays there: 57.9 / 62.2, 65 536: 91.7 / 94.9, full scan 179 / 150)

__device__ __forceinline__ int tb_insert(int* s_key, int key, int probes)
{
    unsigned h = ((unsigned)key * 2654435761u) >> 25;          // top 7 bits -> [0, 128)
#pragma unroll 1
    for (int i = 0; i < probes; ++i) {
        const int prev = atomicCAS(&s_key[h], -1, key);         // lanes of this wave racing for a slot (integer CAS: cheap; a plain read
        if (prev == -1 || prev == key) return (int)h;           // in front of it for rows already in the table measured no gain)
        h = (h + 1) & (TB_SLOTS - 1);
    }
    return -1;
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(NL_FIELD_THREADS, 4) void k_trilinear_bwd(FieldArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_val_all[TB_WAVES * TB_SLOTS * NL_C];
    __shared__ int s_key_all[TB_WAVES * TB_SLOTS];
    __shared__ double s_pose[NL_MAX_FRAMES * 12];
    __shared__ int s_new[TB_WAVES * TB_SLOTS], s_new_n, s_new_base;     // rows this workgroup touches first (a.touched): one list append per workgroup
    __shared__ unsigned char s_own_all[TB_WAVES * TB_SLOTS];            // per table slot: the lane whose claim of the slot stands in this round (flush_run)
    const int P = a.ls->P;
    int grid = (int)gridDim.x;
    if (a.resident_blocks > 0 && a.resident_blocks < grid && P <= TB_ONE_ROUND_SPAN * TB_GROUPS * a.resident_blocks) grid = a.resident_blocks;
    if ((int)blockIdx.x >= grid) return;
    if (threadIdx.x == 0) s_new_n = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* s_val = s_val_all + wv * TB_SLOTS * NL_C;            // this wave's table
    // this wave's copy of the accumulators (NlTouchedRows.copies: same-address atomics on the near-sensor rows of an accumulated map spread over the copies)
    float* const gacc = a.g_emb ? a.g_emb + (size_t)((blockIdx.x * TB_WAVES + wv) % (unsigned)a.touched.copies) * (size_t)a.touched.copy_stride : nullptr;
    int* s_key = s_key_all + wv * TB_SLOTS;
    unsigned char* s_own = s_own_all + wv * TB_SLOTS;
    for (int i = threadIdx.x; i < a.n_frames * 12; i += NL_FIELD_THREADS) s_pose[i] = 0.0;
    for (int i = lane; i < TB_SLOTS; i += 64) s_key[i] = -1;
    for (int i = lane; i < TB_SLOTS * NL_C; i += 64) s_val[i] = 0.f;
    FSTAMP(0);
    __syncthreads();
    FSTAMP(1);
    const int k = threadIdx.x & 7;                              // this lane's voxel corner
    const int grp = threadIdx.x >> 3;
    const int gw = lane >> 3;                                   // group index inside the wave
    const int lane0 = lane & ~7;                                // first lane of the group inside its wave
    // the P samples are split EVENLY over the workgroups (one span each, one table flush each): with fixed-size chunks
    // P = 1.05 x grid x chunk would send 5 % of the workgroups through a second chunk and double the kernel's time
    int per_group = (P + grid * TB_GROUPS - 1) / (grid * TB_GROUPS);
    // ... unless the rays are long in samples (an accumulated map: ~23 samples per hit ray over ~15 voxels against 8 over 3-4 on a one-scan map).
    // There the voxels around the sensor are crossed by every ray, their rows are the target of thousands of same-address atomics
    // (profiles/experiments/README.md), and a wave that walks 16 samples merges less than one ray's worth of them: 6 samples per group
    // (two rays per wave) - 150-scan map 2048 rays 77 -> 62 us, 4096 rays 125 -> 108; the one-scan map at 2048 rays would pay 14.5 -> 22 us for it.
    const int min_span = P < TB_SMALL_P ? (P > TB_LONG_RAY_SAMPLES * a.ls->R ? TB_MIN_SPAN_LONG_RAYS : TB_MIN_SPAN_SMALL) : TB_MIN_SPAN;
    if (per_group * TB_GROUPS < min_span) per_group = min_span / TB_GROUPS;
    const int span = per_group * TB_GROUPS;
    const int nchunks = (P + span - 1) / span;
    // pose partials (corner-0 lane).  The sums cancel heavily (rays in all directions), so they are carried in fp64: fp32 over
    // the few samples of one ray (ra: sum dx, sum depth * dx), folded at every ray change into the workgroup's fp64 accumulators
    // in LDS (ds_add_f64) and from there to memory (global_atomic_add_f64) - the result is independent of the summation order
    // to ~1e-13, i.e. reproducible run to run after the optimiser's rounding to fp32.
    // Lanes 0..2 of a sample's group carry one component each (lane i: sum dx_i, sum depth * dx_i): the division, the running sums and
    // the fold are one component's work per lane instead of three in lane 0 (a wave instruction costs the same with one lane or with three).
    int pf = -1, pr = -1;
    float ra0 = 0.f, ra1 = 0.f, rds[3] = {0.f, 0.f, 0.f};
    auto fold_ray = [&]() {
        if (pf < 0) return;
        double* sp = s_pose + 12 * pf;
        atomicAdd(sp + k, (double)ra0);
        const double t = (double)ra1;
        atomicAdd(sp + 3 + 3 * k, t * (double)rds[0]); atomicAdd(sp + 4 + 3 * k, t * (double)rds[1]); atomicAdd(sp + 5 + 3 * k, t * (double)rds[2]);
        ra0 = 0.f; ra1 = 0.f;
    };
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += grid) {
        const int s_end = min(P, (chunk + 1) * span);
        int cur_vox = -1, row = -1;
        float acc[NL_C];
        unsigned eb[NL_C / 2];                                      // this lane's embedding row, packed bf16 pairs (pose gradient)
#pragma unroll
        for (int c = 0; c < NL_C; ++c) acc[c] = 0.f;
#pragma unroll
        for (int c = 0; c < NL_C / 2; ++c) eb[c] = 0u;
        // add the finished runs (row, acc) of the lanes with a table slot (slot >= 0) into the wave's table.  All 64 lanes call it together.
        auto add_runs = [&](int slot) {
            // Lanes whose slots differ add together; two groups of the wave that finish a run on the SAME row (neighbouring rays
            // share voxel corners) must not read-modify-write the slot at once: every pending lane claims its slot, the claim that
            // stands adds its 16 floats, the others go again - 1-3 rounds instead of one per group (eight dependent LDS round trips
            // of ~300 cycles on the path of every sample step).  Which claim stands is the LDS's fixed write order: run-to-run stable.
            bool pend = slot >= 0;
#pragma unroll 1
            while (__ballot(pend) != 0ull) {
                if (pend) s_own[slot] = (unsigned char)lane;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool win = pend && s_own[slot] == (unsigned char)lane;
                if (win) {
                    float4* dst = reinterpret_cast<float4*>(s_val + slot * NL_C);
                    float4 v0 = dst[0], v1 = dst[1], v2 = dst[2], v3 = dst[3];
                    v0.x += acc[0]; v0.y += acc[1]; v0.z += acc[2]; v0.w += acc[3];
                    v1.x += acc[4]; v1.y += acc[5]; v1.z += acc[6]; v1.w += acc[7];
                    v2.x += acc[8]; v2.y += acc[9]; v2.z += acc[10]; v2.w += acc[11];
                    v3.x += acc[12]; v3.y += acc[13]; v3.z += acc[14]; v3.w += acc[15];
                    dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();                    // the slot is written before the next claim's lane reads it
                pend = pend && !win;
            }
        };
        // add this lane's finished run (row, acc) into the wave's table; `flush` may differ per 8-lane group but is uniform
        // inside a group.  All 64 lanes call it together.
        auto flush_run = [&](bool flush, int steps_left) {
            flush = flush && row >= 0 && a.want_emb_grad;
            const unsigned long long fm = __ballot(flush);
            if (fm == 0ull) return;
            int slot = -1;
            if (flush) slot = tb_insert(s_key, row, a.probes);
            // A run that finds no slot (a wave that meets more distinct rows than its table holds: the accumulated map, where a ray crosses ~15
            // voxels with ~1.5 samples each) used to go to memory from its own lane - 16 scalar atomics per lane, the slowest thing this kernel can
            // do (scripts/scatter_sweep.py --probes).  Now the wave writes the table out the way the end of a span does - 16 lanes per slot, one
            // 64-byte row per atomic instruction quarter - and the run goes into the emptied table.  Runs that did find a slot add first.
            bool ovf = flush && slot < 0;
            // (Not when the table is half empty - a probe sequence also fails on clustering - and not in the last steps of a span, where the table
            //  is about to be written out anyway and the odd lane's scalar atomics are cheaper than a second pass over 128 slots: a rank's
            //  interleaved share of the single-scan map fills its tables right at the end of its 5-sample spans, 40.9 -> 46.4 us without this test.
            //  scripts/scatter_sweep.py [--large-map] --probes, never / always / with >= 2 steps left: that share 40.9 / 46.4 / 40.9 us; 150-scan map
            //  8192 rays 178 / 168 / 178, 16 384 rays 318 / 295 / 318, 32 768 rays 579 / 366 / 392, 65 536 rays 1137 / 483 / 503; a criterion on the
            //  number of lanes without a slot did not separate the two maps.)
            if (__ballot(ovf) != 0ull && steps_left >= a.flush_steps_left &&
                __popcll(__ballot(s_key[lane] >= 0)) + __popcll(__ballot(s_key[lane + 64] >= 0)) >= TB_FLUSH_OCCUPANCY) {
                add_runs(slot);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int base = 0; base < TB_SLOTS; base += 4) {
                    const int sl = base + (lane >> 4), c = lane & 15;
                    const int key = s_key[sl];
                    if (key >= 0) {
                        const float v = s_val[sl * NL_C + c];
                        if (c == 0) nl_touch_row(a.touched, key);
                        if (v != 0.f) atomicAdd(gacc + (size_t)key * NL_C + c, v);
                        s_val[sl * NL_C + c] = 0.f;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < TB_SLOTS; i += 64) s_key[i] = -1;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                slot = ovf ? tb_insert(s_key, row, a.probes) : -1;     // at most 64 runs into 128 empty slots
                ovf = ovf && slot < 0;                                 // (a probe sequence can still fail - with the A/B aid's 1 or 2 probes it does)
            }
            if (ovf) {                                              // straight to memory from this lane
                nl_touch_row(a.touched, row);
                float* dst = gacc + (size_t)row * NL_C;
#pragma unroll
                for (int c = 0; c < NL_C; ++c) atomicAdd(dst + c, acc[c]);
            }
            add_runs(slot);
        };
        const int s_base = chunk * span + grp * per_group;
#pragma unroll 1
        for (int j = 0; j < per_group; ++j) {
            const int s = s_base + j;
            const bool live = s < s_end;                            // the 8 lanes of a group share s; dead groups keep calling flush_run
            if (__ballot(live) == 0ull) break;
            SampleGeom g;
            if (live) {
                // the sample's local coordinates, one axis per lane (lanes 0..2 of the group; the others repeat axis 2) and one shuffle
                // per axis, instead of all three axes - nine loads, three IEEE divisions - in every lane
                g.vox = a.s_vox[s]; g.ray = a.s_ray[s]; g.depth = a.s_depth[s];
                const int ax = k < 2 ? k : 2;
                const float t_ax = a.poses[12 * (a.frame_id ? a.frame_id[g.ray] : 0) + 9 + ax];
                const float x_ax = t_ax + a.rays_d_world[3 * g.ray + ax] * g.depth;          // ray(): o + d * depth
                const float p_ax = (x_ax - a.centres[3 * g.vox + ax]) / a.voxel_size + 0.5f;  // nl_trilinear_p, one axis
                g.p[0] = __shfl(p_ax, lane0); g.p[1] = __shfl(p_ax, lane0 + 1); g.p[2] = __shfl(p_ax, lane0 + 2);
            } else { g.vox = cur_vox; g.ray = 0; g.depth = 0.f; g.p[0] = g.p[1] = g.p[2] = 0.f; }
            const bool change = live && g.vox != cur_vox;
            flush_run(change, per_group - 1 - j);
            if (change) {
                cur_vox = g.vox;
                row = a.vertex_rows[8 * (size_t)g.vox + k];
#pragma unroll
                for (int c = 0; c < NL_C; ++c) acc[c] = 0.f;
                if (a.want_pose_grad) {
                    const uint4* er = reinterpret_cast<const uint4*>(a.emb + (size_t)row * NL_C);
                    const uint4 v0 = er[0], v1 = er[1];
                    eb[0] = v0.x; eb[1] = v0.y; eb[2] = v0.z; eb[3] = v0.w; eb[4] = v1.x; eb[5] = v1.y; eb[6] = v1.z; eb[7] = v1.w;
                }
            }
            if (!live) continue;
            // this lane's corner weight (nl_trilinear_w's w[k]: (tx * ty) * tz)
            const float wk = (((k & 4) ? g.p[0] : 1.0f - g.p[0]) * ((k & 2) ? g.p[1] : 1.0f - g.p[1])) * ((k & 1) ? g.p[2] : 1.0f - g.p[2]);
            float d[NL_C];
            {
                const float4* di = reinterpret_cast<const float4*>(a.dX + (size_t)s * NL_C);
                const float4 d0 = di[0], d1 = di[1], d2 = di[2], d3 = di[3];
                d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
                d[8] = d2.x; d[9] = d2.y; d[10] = d2.z; d[11] = d2.w; d[12] = d3.x; d[13] = d3.y; d[14] = d3.z; d[15] = d3.w;
            }
            if (a.want_emb_grad) {
                // bf16(w_k dX) as torch's embedding backward forms it: round-to-nearest-even by the hardware conversion (two values per
                // instruction; nl_round_bf16 is the same rounding in five integer instructions per value - 80 of the loop's ~440)
#pragma unroll
                for (int c = 0; c < NL_C; c += 2) {
                    const f32x2_t pr2 = {wk * d[c], wk * d[c + 1]};
                    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(pr2, bf16x2_t));
                    acc[c] += __uint_as_float(u << 16); acc[c + 1] += __uint_as_float(u & 0xFFFF0000u);
                }
            }
            if (!a.want_pose_grad) continue;
            float t0 = 0.f, t1 = 0.f;                               // <e_k, dX>: 8 + 8 channels, then add
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t0 += __uint_as_float(eb[c] << 16) * d[2 * c];                 t0 += __uint_as_float(eb[c] & 0xFFFF0000u) * d[2 * c + 1];
                t1 += __uint_as_float(eb[4 + c] << 16) * d[8 + 2 * c];         t1 += __uint_as_float(eb[4 + c] & 0xFFFF0000u) * d[8 + 2 * c + 1];
            }
            const float dotk = t0 + t1;
            float dot[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) dot[i] = __shfl(dotk, lane0 + i);
            if (k > 2) continue;
            float dp[3]; nl_trilinear_dp(g.p, dot, dp);
            if (g.ray != pr) {
                fold_ray();
                pr = g.ray;
                rds[0] = a.rays_d_sensor[3 * g.ray]; rds[1] = a.rays_d_sensor[3 * g.ray + 1]; rds[2] = a.rays_d_sensor[3 * g.ray + 2];
                pf = a.frame_id ? a.frame_id[g.ray] : 0;
            }
            const float dx = (k == 0 ? dp[0] : (k == 1 ? dp[1] : dp[2])) / a.voxel_size;
            ra0 += dx;
            ra1 += g.depth * dx;
        }
        FSTAMP(2);
        flush_run(true, 0);
        FSTAMP(3);
        if (a.want_emb_grad) {
            // rows whose accumulators receive their first contribution since begin_call go onto the touched-rows list (a.touched).  Their
            // flag words are loaded first, so that the round trip runs under the flush below
            int tkey[TB_SLOTS / 64]; unsigned tword[TB_SLOTS / 64];
            if (a.touched.flags) {
#pragma unroll
                for (int t = 0; t < TB_SLOTS / 64; ++t) {
                    tkey[t] = s_key[lane + 64 * t];
                    tword[t] = tkey[t] >= 0 ? *reinterpret_cast<volatile unsigned*>(a.touched.flags + (tkey[t] >> 5)) : 0xFFFFFFFFu;
                }
            }
            // flush this wave's table: 16 lanes per slot (one channel each), touched rows only; slots are reset for the next span
            for (int base = 0; base < TB_SLOTS; base += 4) {
                const int slot = base + (lane >> 4), c = lane & 15;
                const int key = s_key[slot];
                if (key >= 0) {
                    const float v = s_val[slot * NL_C + c];
                    if (v != 0.f) atomicAdd(gacc + (size_t)key * NL_C + c, v);
                    s_val[slot * NL_C + c] = 0.f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < TB_SLOTS; i += 64) s_key[i] = -1;
            FSTAMP(5);
            if (a.touched.flags) {
                // the flag bit is set with a scattered atomic; the list position comes from ONE same-address atomic per workgroup (such
                // an atomic costs ~12 ns whoever issues it: one per row was 0.6 ms per 20-iteration call at 2048 rays)
#pragma unroll
                for (int t = 0; t < TB_SLOTS / 64; ++t) {
                    const unsigned bit = 1u << (tkey[t] & 31);
                    if (tkey[t] >= 0 && !(tword[t] & bit) && !(atomicOr(a.touched.flags + (tkey[t] >> 5), bit) & bit)) s_new[atomicAdd(&s_new_n, 1)] = tkey[t];
                }
                __syncthreads();
                const int n_new = s_new_n;                          // snapshot: nobody may see a later chunk's increment of the counter
                __syncthreads();
                if (n_new > 0) {                                    // uniform over the workgroup; after the first iterations of a call: rare
                    if (threadIdx.x == 0) { s_new_base = atomicAdd(a.touched.count, n_new); s_new_n = 0; }
                    __syncthreads();
                    for (int i = threadIdx.x; i < n_new; i += NL_FIELD_THREADS) a.touched.list[s_new_base + i] = s_new[i];
                    __syncthreads();
                }
            }
        }
    }
    if (a.want_pose_grad) {
        if (k <= 2) fold_ray();
        __syncthreads();
        for (int i = threadIdx.x; i < a.n_frames * 12; i += NL_FIELD_THREADS)
            if (s_pose[i] != 0.0) atomicAdd(a.g_pose + i, s_pose[i]);
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

static std::atomic<long long*> g_field_dbg{nullptr};   // process-global A/B / profiling state (include/nerfloam_hip_debug.h), relaxed atomics
static std::atomic<int> g_field_probes{TB_PROBES};   // (A/B aid: nl_field_set_probes)
static std::atomic<int> g_field_flush_steps{TB_FLUSH_MIN_STEPS_LEFT};   // (A/B aid: nl_field_set_midspan_flush; a huge value = never)
static std::atomic<int> g_field_one_round{1};       // k_trilinear_bwd: TB_ONE_ROUND_SPAN rule on (0: every launched workgroup takes samples; A/B aid)

// workgroups of k_trilinear_bwd the current device holds at once: 4 per compute unit (its 33 KB of LDS and launch bounds)
static int field_resident_blocks()
{
    static int cached = 0;
    if (cached == 0) {
        int dev = 0; hipDeviceProp_t prop;
        cached = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                     ? 4 * prop.multiProcessorCount : -1;
    }
    return cached > 0 ? cached : 0;
}

extern "C" {

static int fill_args(FieldArgs& a, const void* ls, const int* s_vox, const float* s_depth, const int* s_ray,
                     const float* rays_d_world, const float* rays_d_sensor, const int* frame_id, const float* poses, int n_frames,
                     const float* centres, const int* vertex_rows, const void* emb, float voxel_size)
{
    a.dbg = nullptr; a.touched.list = nullptr; a.touched.count = nullptr; a.touched.flags = nullptr; a.touched.copies = 1; a.touched.copy_stride = 0;
    if (!ls || !s_vox || !s_depth || !s_ray || !rays_d_world || !poses || !centres || !vertex_rows || !emb) return NL_ERR_INVALID_ARG;
    if (n_frames <= 0 || n_frames > NL_MAX_FRAMES) return NL_ERR_INVALID_ARG;
    a.ls = (const NlLossScalars*)ls; a.s_vox = s_vox; a.s_depth = s_depth; a.s_ray = s_ray; a.rays_d_world = rays_d_world;
    a.rays_d_sensor = rays_d_sensor; a.frame_id = frame_id; a.poses = poses; a.centres = centres; a.vertex_rows = vertex_rows;
    a.emb = (const uint16_t*)emb; a.voxel_size = voxel_size; a.n_frames = n_frames;
    a.X = nullptr; a.dX = nullptr; a.g_emb = nullptr; a.g_pose = nullptr; a.want_emb_grad = 0; a.want_pose_grad = 0;
    a.resident_blocks = 0; a.probes = g_field_probes; a.flush_steps_left = g_field_flush_steps;
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

int nl_gather_grid(int n_vox, int vox0, int res, const float* lin, const float* centres, const int* vertex_rows, const void* emb, float voxel_size,
                   float* X, void* stream)
{
    if (n_vox < 0 || vox0 < 0 || res < 2 || res > 64 || !lin || !centres || !vertex_rows || !emb || !X) return NL_ERR_INVALID_ARG;
    if (n_vox == 0) return NL_OK;
    const long long P = (long long)n_vox * res * res * res;
    const long long nb = (P * 2 + NL_FIELD_THREADS - 1) / NL_FIELD_THREADS;
    hipLaunchKernelGGL(k_gather_grid, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(NL_FIELD_THREADS), 0, (hipStream_t)stream, P, vox0, res, lin, centres,
                       vertex_rows, (const uint16_t*)emb, voxel_size, X);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* A/B aid: 0 = every launched workgroup of k_trilinear_bwd takes samples (the pre-round-4 behaviour), 1 = the one-round rule (default) */
int nl_field_set_one_round(int on) { g_field_one_round = on != 0; return NL_OK; }
int nl_field_set_midspan_flush(int min_steps_left) { g_field_flush_steps = min_steps_left < 0 ? (1 << 30) : min_steps_left; return NL_OK; }
int nl_field_set_probes(int n) { if (n < 1 || n > TB_SLOTS) return NL_ERR_INVALID_ARG; g_field_probes = n; return NL_OK; }
/* profiling aid: device buffer [nblocks][8] int64 receiving s_memtime stamps of k_trilinear_bwd (NULL disables) */
int nl_field_set_debug_buffer(void* dbg) { g_field_dbg = (long long*)dbg; return NL_OK; }

int nl_trilinear_bwd_t(const void* loss_scalars, const int* s_vox, const float* s_depth, const int* s_ray,
                       const float* rays_d_world, const float* rays_d_sensor, const int* frame_id, const float* poses, int n_frames,
                       const float* centres, const int* vertex_rows, const void* emb, float voxel_size,
                       const float* dX, float* g_emb, double* g_pose, int nblocks, const NlTouchedRows* touched, void* stream)
{
    FieldArgs a;
    int rc = fill_args(a, loss_scalars, s_vox, s_depth, s_ray, rays_d_world, rays_d_sensor, frame_id, poses, n_frames, centres, vertex_rows, emb, voxel_size);
    if (rc != NL_OK || !dX || nblocks <= 0 || (g_pose && !rays_d_sensor)) return NL_ERR_INVALID_ARG;
    if (touched && (touched->struct_size != (int)sizeof(NlTouchedRows) || (touched->flags && (!touched->list || !touched->count)))) return NL_ERR_INVALID_ARG;
    a.dX = dX; a.g_emb = g_emb; a.g_pose = g_pose; a.want_emb_grad = g_emb != nullptr; a.want_pose_grad = g_pose != nullptr;
    a.dbg = g_field_dbg;
    a.resident_blocks = g_field_one_round ? field_resident_blocks() : 0;
    if (touched && g_emb) {
        a.touched.list = touched->list; a.touched.count = touched->count; a.touched.flags = touched->flags;
        if (touched->copies > 1) {
            if (!touched->flags || touched->copy_stride <= 0) return NL_ERR_INVALID_ARG;     // copies are folded by the sweep over the touched rows only
            a.touched.copies = touched->copies; a.touched.copy_stride = touched->copy_stride;
        }
    }
    hipLaunchKernelGGL(k_trilinear_bwd, dim3(nblocks), dim3(NL_FIELD_THREADS), 0, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_trilinear_bwd(const void* loss_scalars, const int* s_vox, const float* s_depth, const int* s_ray,
                     const float* rays_d_world, const float* rays_d_sensor, const int* frame_id, const float* poses, int n_frames,
                     const float* centres, const int* vertex_rows, const void* emb, float voxel_size,
                     const float* dX, float* g_emb, double* g_pose, int nblocks, void* stream)
{
    return nl_trilinear_bwd_t(loss_scalars, s_vox, s_depth, s_ray, rays_d_world, rays_d_sensor, frame_id, poses, n_frames, centres, vertex_rows, emb,
                              voxel_size, dX, g_emb, g_pose, nblocks, nullptr, stream);
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
