// nl_geometry.hip -- ray/octree intersection, hit compaction (scan) and inverse-CDF ray sampling
// for gfx950.  HBM/L2-bound integer+fp32 kernels: one ray per lane, octree read in place (the
// reference replicates it G<=256 times, voxel_helpers.py:106-108), DFS cursors in LDS, samples
// written compacted through an exclusive scan instead of a padded [R, max_steps] tensor.
//
// Reference behaviour: third_party/sparse_voxels/src/intersect_gpu.cu:193-272,
// src/variations/voxel_helpers.py:531-598, third_party/sparse_voxels/src/sample_gpu.cu:133-239.
#include "nl_common.h"
#include <atomic>

#define NL_GEO_THREADS 256

// ---------------------------------------------------------------------------------------------
// DFS cursor stack in LDS: entry = (node << 4) | (cursor + 1), laid out [level][thread] so that a
// wave's accesses to one level hit 64 consecutive banks.
// ---------------------------------------------------------------------------------------------
struct LdsStack {
    unsigned* base;     // &lds[threadIdx.x]
    __device__ __forceinline__ void set(int l, int node, int cur) { base[l * NL_GEO_THREADS] = ((unsigned)node << 4) | (unsigned)(cur + 1); }
    __device__ __forceinline__ int node(int l) const { return (int)(base[l * NL_GEO_THREADS] >> 4); }
    __device__ __forceinline__ int cursor(int l) const { return (int)(base[l * NL_GEO_THREADS] & 15u) - 1; }
    __device__ __forceinline__ void set_cursor(int l, int cur) {
        unsigned v = base[l * NL_GEO_THREADS];
        base[l * NL_GEO_THREADS] = (v & ~15u) | (unsigned)(cur + 1);
    }
};

// ---------------------------------------------------------------------------------------------
// (b1) drop-in kernel behind grid.svo_intersect: raw DFS-order hits, reference tensor layouts
// [B,m,3] rays, [B,n,3]/[B,n,9] octree per batch row, [B,m,n_max] outputs (idx -1 padded, depths
// zero where unused like intersect.cpp:98-106).
// ---------------------------------------------------------------------------------------------
template <bool DIRECT>                                  // DIRECT: n_max beyond the 20 register slots - the walk records straight into the ray's output row
__global__ __launch_bounds__(NL_GEO_THREADS) void k_svo_intersect_raw(
    int b, int n, int m, float voxelsize, int n_max,
    const float* __restrict__ ray_start, const float* __restrict__ ray_dir,
    const float* __restrict__ points, const int* __restrict__ children,
    int* __restrict__ idx, float* __restrict__ min_depth, float* __restrict__ max_depth)
{
    __shared__ unsigned s_stack[NL_MAX_LEVELS * NL_GEO_THREADS];
    const long long gid = (long long)blockIdx.x * NL_GEO_THREADS + threadIdx.x;
    if (gid >= (long long)b * m) return;
    const int bi = (int)(gid / m);
    const float* pts = points + (size_t)bi * n * 3;
    const int* ch = children + (size_t)bi * n * 9;
    const float* o = ray_start + gid * 3;
    const float* d = ray_dir + gid * 3;
    LdsStack stk{&s_stack[threadIdx.x]};
    int* oi = idx + gid * n_max; float* o0 = min_depth + gid * n_max; float* o1 = max_depth + gid * n_max;
    if (DIRECT) {
        const int cnt = nl_octree_walk(pts, ch, o[0], o[1], o[2], d[0], d[1], d[2], voxelsize * 0.5f, n_max, stk, oi, o0, o1);
        for (int l = cnt; l < n_max; ++l) { oi[l] = -1; o0[l] = 0.0f; o1[l] = 0.0f; }
    } else {
        int hi[NL_MAX_HITS]; float h0[NL_MAX_HITS], h1[NL_MAX_HITS];
        const int cnt = nl_octree_walk(pts, ch, o[0], o[1], o[2], d[0], d[1], d[2], voxelsize * 0.5f, n_max, stk, hi, h0, h1);
        for (int l = 0; l < n_max; ++l) {
            const bool v = l < cnt;
            oi[l] = v ? hi[l] : -1;
            o0[l] = v ? h0[l] : 0.0f;
            o1[l] = v ? h1[l] : 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused ray set-up + intersect + sort/cull (render_helpers.py:366-388, voxel_helpers.py:531-567).
//   inputs per ray: unit direction in the sensor frame, gt return (sensor frame), cos, frame id
//   poses[F][12]: rotation row-major (9) then translation (3)
//   octree in the CHILDREN-BLOCK layout built once per map update (pipeline.pack_children_blocks):
//     blk_hdr [B]    int2    (index of the first child block, exist mask | own-a-block mask << 8); the
//                            children blocks of a block are numbered consecutively in octant order (BFS)
//     blk_ids [B][8] int32   child node ids (read on the bottom level only, to name the hit voxels)
//     block 0 is a pseudo block for the root; block b >= 1 belongs to one interior node
//   Same hits in the same order as the reference DFS (children visited in descending octant order,
//   leaves recorded until 20).  Expanding a node costs ONE 8-byte load: child centres are recomputed from
//   the integer lattice path instead of being fetched (the per-lane divergent 16-byte loads of node records
//   saturated the CU's address path: ~55 expansions x 11 loads per ray), and the 8 children are slab-tested
//   together; the reference pays two dependent memory round trips per visited child.
//   Per-level DFS state (block, surviving-children bitmask) lives in LDS; hits go to the ray's own
//   output row in DFS order and are then sorted in place (L2-resident).
//   outputs: world direction, gt distance*cos, hit_idx/t0/t1[N,20] (first hit_count[r] entries are
//   the sorted, culled hits; the rest of a row is NOT written), hit count; counters[NLC_HMAX].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool slab_inv(const float o[3], const float inv[3], float cx, float cy, float cz, float half,
                                         float* tn, float* tf)
{
    // nl_slab with the per-ray reciprocals hoisted: identical arithmetic (inv = 1.0f / d is the same value)
    float lo = 0.0f, hi = 100000.0f;
    const float c[3] = {cx, cy, cz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t0 = (c[a] - half - o[a]) * inv[a];
        float t1 = (c[a] + half - o[a]) * inv[a];
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

// The sequential DFS over the work items [first_block, ...) of `n_blocks` workgroups: body of k_ray_intersect_dfs, also called by the
// one-workgroup hit-ray scan (k_scan_hits_fused) for the rays the work-list kernel handed over (normally none).
struct DfsArgs {
    int N; const float* rays_d_sensor; const float* points_gt; const float* cos_gt; const int* frame_id; const float* poses;
    const int2* blk_hdr; const int4* blk_ids; int root_side; float voxel_size, max_distance;
    float* rays_d_world; float* gt_dist; int* hit_idx; float* hit_t0; float* hit_t1; int* hit_count; int* counters; const int* ray_list;
};
__device__ __forceinline__ void dfs_rays(const DfsArgs& a, int first_block, int n_blocks)
{
    const int N = a.N;
    const float* __restrict__ rays_d_sensor = a.rays_d_sensor; const float* __restrict__ points_gt = a.points_gt;
    const float* __restrict__ cos_gt = a.cos_gt; const int* __restrict__ frame_id = a.frame_id; const float* __restrict__ poses = a.poses;
    const int2* __restrict__ blk_hdr = a.blk_hdr; const int4* __restrict__ blk_ids = a.blk_ids;
    const int root_side = a.root_side; const float voxel_size = a.voxel_size, max_distance = a.max_distance;
    float* __restrict__ rays_d_world = a.rays_d_world; float* __restrict__ gt_dist = a.gt_dist;
    int* __restrict__ hit_idx = a.hit_idx; float* __restrict__ hit_t0 = a.hit_t0; float* __restrict__ hit_t1 = a.hit_t1;
    int* __restrict__ hit_count = a.hit_count; int* __restrict__ counters = a.counters; const int* __restrict__ ray_list = a.ray_list;
    __shared__ int s_base[NL_MAX_LEVELS * NL_GEO_THREADS];
    __shared__ unsigned char s_mask[NL_MAX_LEVELS * NL_GEO_THREADS];
    __shared__ unsigned char s_has[NL_MAX_LEVELS * NL_GEO_THREADS];
    __shared__ int s_hmax;
    if (threadIdx.x == 0) s_hmax = 0;
    __syncthreads();
    // ray_list != null: fallback pass over the rays the queue kernel could not finish (count in counters[NLC_ISECT_OVF])
    const int n_work = ray_list ? counters[NLC_ISECT_OVF] : N;
    for (int w0 = first_block * NL_GEO_THREADS; w0 < n_work; w0 += n_blocks * NL_GEO_THREADS) {
    const int wi = w0 + threadIdx.x;
    const int r = wi < n_work ? (ray_list ? ray_list[wi] : wi) : N;
    int valid = 0;
    if (r < N) {
        const float* P = poses + 12 * (frame_id ? frame_id[r] : 0);
        const float s0 = rays_d_sensor[3 * r], s1 = rays_d_sensor[3 * r + 1], s2 = rays_d_sensor[3 * r + 2];
        float d[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[i] = (s0 * P[3 * i] + s1 * P[3 * i + 1]) + s2 * P[3 * i + 2];
        rays_d_world[3 * r] = d[0]; rays_d_world[3 * r + 1] = d[1]; rays_d_world[3 * r + 2] = d[2];
        const float gx = points_gt[3 * r], gy = points_gt[3 * r + 1], gz = points_gt[3 * r + 2];
        gt_dist[r] = sqrtf((gx * gx + gy * gy) + gz * gz) * cos_gt[r];
        const float o[3] = {P[9], P[10], P[11]};
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        const float half_voxel = voxel_size * 0.5f;
        int* oi = hit_idx + (size_t)r * NL_MAX_HITS; float* o0 = hit_t0 + (size_t)r * NL_MAX_HITS; float* o1 = hit_t1 + (size_t)r * NL_MAX_HITS;
        int* st_base = s_base + threadIdx.x; unsigned char* st_mask = s_mask + threadIdx.x; unsigned char* st_has = s_has + threadIdx.x;
        int cnt = 0, lvl = -1;
        int px = 0, py = 0, pz = 0;                                 // min-corner voxel coordinates of the node being expanded

        // Expand the block of the node at (px,py,pz) whose children have side `cs` (a power of two).
        // Node centres are NOT loaded: the reference's centre = (xyz + side/2) * voxel_size (mapping.py:322) is
        // recomputed from the integer lattice path, bit-for-bit (xyz and side/2 are exact in fp32), so an expansion
        // costs ONE 8-byte load (+ the 8 leaf ids on the bottom level) instead of 8 x 16-byte records.
        auto expand = [&](int b, int cs) {
            const int2 hdr = blk_hdr[b];
            const unsigned exist = (unsigned)hdr.y & 255u, has = ((unsigned)hdr.y >> 8) & 255u;
            const float fs = (float)cs, hs = fs * 0.5f, half = half_voxel * fs;
            unsigned m = 0;
            if (cs == 1) {                                          // bottom level: children are voxels -> record hits, descending octant
                const int4 ia = blk_ids[2 * (size_t)b], ib = blk_ids[2 * (size_t)b + 1];
                const int ids[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
                for (int u = 7; u >= 0; --u) {
                    if (!((exist >> u) & 1u) || ids[u] < 0) continue;
                    const float cx = ((float)(px + (u & 1)) + hs) * voxel_size, cy = ((float)(py + ((u >> 1) & 1)) + hs) * voxel_size,
                                cz = ((float)(pz + ((u >> 2) & 1)) + hs) * voxel_size;
                    float tn, tf;
                    if (slab_inv(o, inv, cx, cy, cz, half, &tn, &tf) && cnt < NL_MAX_HITS) { oi[cnt] = ids[u]; o0[cnt] = tn; o1[cnt] = tf; ++cnt; }
                }
                return;
            }
#pragma unroll
            for (int u = 7; u >= 0; --u) {
                if (!((has >> u) & 1u)) continue;                   // a hit child without a block has nothing below it
                const float cx = ((float)(px + ((u & 1) ? cs : 0)) + hs) * voxel_size, cy = ((float)(py + ((u & 2) ? cs : 0)) + hs) * voxel_size,
                            cz = ((float)(pz + ((u & 4) ? cs : 0)) + hs) * voxel_size;
                float tn, tf;
                if (slab_inv(o, inv, cx, cy, cz, half, &tn, &tf)) m |= 1u << u;
            }
            if (m) { ++lvl; st_base[lvl * NL_GEO_THREADS] = hdr.x; st_mask[lvl * NL_GEO_THREADS] = (unsigned char)m; st_has[lvl * NL_GEO_THREADS] = (unsigned char)has; }
        };
        {   // root (block 0 is the pseudo block holding it): slab test, then its children block
            const float fs = (float)root_side, hs = fs * 0.5f;
            float tn, tf;
            const int2 h0 = blk_hdr[0];
            if (slab_inv(o, inv, hs * voxel_size, hs * voxel_size, hs * voxel_size, half_voxel * fs, &tn, &tf) && h0.x >= 0)
                expand(h0.x, root_side >> 1);
        }
        while (lvl >= 0 && cnt < NL_MAX_HITS) {
            const unsigned m = st_mask[lvl * NL_GEO_THREADS];
            if (m == 0) { --lvl; continue; }
            const int u = 31 - __clz((int)m);                       // highest octant first (reference pop order)
            st_mask[lvl * NL_GEO_THREADS] = (unsigned char)(m & ~(1u << u));
            // stack level l holds children of side root_side >> (l + 1); the chosen child becomes the expanded node
            const int cs = root_side >> (lvl + 1);
            const int keep = ~(2 * cs - 1);
            px = (px & keep) | ((u & 1) ? cs : 0); py = (py & keep) | ((u & 2) ? cs : 0); pz = (pz & keep) | ((u & 4) ? cs : 0);
            // children blocks of one block are numbered consecutively in octant order: no load needed to find it
            const int cb = st_base[lvl * NL_GEO_THREADS] + __popc((unsigned)st_has[lvl * NL_GEO_THREADS] & ((1u << u) - 1u));
            expand(cb, cs >> 1);
        }
        // in-place stable insertion sort by t_min on the ray's own row, then cull (voxel_helpers.py:543-552)
        for (int i = 1; i < cnt; ++i) {
            const int ki = oi[i]; const float a = o0[i], b = o1[i];
            int j = i - 1;
            while (j >= 0) {
                const float tj = o0[j];
                if (!(tj > a)) break;
                oi[j + 1] = oi[j]; o0[j + 1] = tj; o1[j + 1] = o1[j];
                --j;
            }
            if (j + 1 != i) { oi[j + 1] = ki; o0[j + 1] = a; o1[j + 1] = b; }
        }
        for (int i = 0; i < cnt; ++i) {
            const bool keep = !(o1[i] > 2.0f * max_distance) && !(o0[i] > max_distance);
            if (keep) ++valid; else { oi[i] = -1; o0[i] = max_distance; o1[i] = max_distance; }
        }
        hit_count[r] = valid;
    }
    // block max of valid hits -> one atomic per block
    int wmax = valid;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&s_hmax, wmax);
    }   // work loop
    __syncthreads();
    if (threadIdx.x == 0 && s_hmax > 0) atomicMax(&counters[NLC_HMAX], s_hmax);
}

__global__ __launch_bounds__(NL_GEO_THREADS) void k_ray_intersect_dfs(DfsArgs a) { dfs_rays(a, blockIdx.x, gridDim.x); }

// ---------------------------------------------------------------------------------------------
// Work-list version of the same traversal: FOUR lanes per ray work through a per-ray LDS stack of pending node
// expansions, so up to four expansions of one ray are in flight at once.  The sequential DFS is bound by the
// ray's dependent chain of ~55-110 expansions (one memory round trip each, ~300 us whatever the ray count);
// with the queue the chain shrinks to about the tree depth.  Expansion order is free because the DFS order of
// the leaves is recoverable afterwards: the reference visits children in descending octant order, so its hit
// order is DESCENDING Morton order of the voxel coordinates (z most significant at each level); the 20-hit cap
// keeps the 20 largest codes.  Hits are then stably sorted by t_min and culled exactly as before.
// Rays whose queue (32) or hit list (24) overflows are appended to a list and redone by k_ray_intersect_dfs.
// ---------------------------------------------------------------------------------------------
// lanes per ray: template parameter of k_ray_intersect_q.  Measured (profiles/r01_k_intersect_lanes_per_ray.txt): more lanes pop
// more pending nodes per round - 16 lanes reach the minimum of one round per tree level (18) - and the traversal is a latency
// chain per round, so small ray counts want 16 lanes (2048 rays: 77 us with 4 lanes, 52 with 8, 39 with 16); at 131 072 rays the
// extra workgroups cost throughput (198 / 162 / 198 us) and 8 is best.  0 = choose by ray count (default: 16 lanes up to 16 384 rays).
// A ray that meets more than 20 voxels (an accumulated map: the ground plane of every scan, walls behind walls) keeps the FIRST 20 in
// DFS order like the reference's capped traversal: whenever its list holds more than 20 hits, the lanes of the ray rank them (DFS
// order), keep the first 20 and remember the 20th as a threshold - later hits behind it and pending subtrees that lie entirely behind
// it are dropped on the spot (the threshold only ever moves forward, so nothing dropped could have made the final 20).  A leaf-parent
// expansion reserves room for all its hits at once or puts itself back on the stack until the list has been compacted (>= 8 free
// slots afterwards: progress); a ray whose pending stack overflows is started again inside the kernel with cautious pops - the
// sequential fallback is left with rays whose stack is too small even one node at a time.
template <int LPR> struct IqCaps { static constexpr int Q = 32, H = 28; };      // 8 / 4 lanes per ray (large ray counts): 1.2 KB of LDS per ray
template <> struct IqCaps<16> { static constexpr int Q = 64, H = 40; };         // 16 lanes per ray (up to 16 384 rays): 2 KB per ray
template <> struct IqCaps<32> { static constexpr int Q = 128, H = 40; };        // 32 lanes per ray (the default up to 4096 rays, and on accumulated maps up to 16 384): 3 KB per ray

__device__ __forceinline__ bool less_msb(unsigned a, unsigned b) { return a < b && a < (a ^ b); }
// true if voxel 1 precedes voxel 2 in the reference's DFS order (= larger z-major Morton code)
__device__ __forceinline__ bool dfs_before(int x1, int y1, int z1, int x2, int y2, int z2)
{
    const unsigned dx = (unsigned)(x1 ^ x2), dy = (unsigned)(y1 ^ y2), dz = (unsigned)(z1 ^ z2);
    unsigned m = dz; int a = z1, b = z2;
    if (less_msb(m, dy)) { m = dy; a = y1; b = y2; }
    if (less_msb(m, dx)) { m = dx; a = x1; b = x2; }
    return a > b;
}

// The 8 children of a node have only two candidate centre coordinates per axis (offset 0 or `cs`): the six (entry, exit)
// parameter pairs are computed once per expansion and every child combines three of them with the early-out sequence
// of slab_inv - the same operations on the same values, so the results are bit-identical to testing each child separately.
struct ChildSlabs {
    float t0[3][2], t1[3][2];
    __device__ __forceinline__ void init(const float o[3], const float inv[3], int px, int py, int pz, int cs, float hs, float voxel_size, float half)
    {
        const int p[3] = {px, py, pz};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float c = ((float)(p[a] + (k ? cs : 0)) + hs) * voxel_size;
                float u0 = (c - half - o[a]) * inv[a];
                float u1 = (c + half - o[a]) * inv[a];
                if (u1 < u0) { const float t = u0; u0 = u1; u1 = t; }
                t0[a][k] = u0; t1[a][k] = u1;
            }
    }
    __device__ __forceinline__ bool hit(int u, float* tn, float* tf) const
    {
        float lo = 0.0f, hi = 100000.0f;
        const int k[3] = {u & 1, (u >> 1) & 1, (u >> 2) & 1};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float a0 = k[a] ? t0[a][1] : t0[a][0], a1 = k[a] ? t1[a][1] : t1[a][0];
            if (a1 < lo) return false;
            if (a0 > hi) return false;
            lo = (a0 > lo) ? a0 : lo;
            hi = (a1 < hi) ? a1 : hi;
            if (lo > hi) return false;
        }
        *tn = lo; *tf = hi;
        return true;
    }
};

static std::atomic<int> g_isect_lpr{0};                // process-global A/B / test state (include/nerfloam_hip_debug.h), relaxed atomics
static std::atomic<int> g_isect_prune{1};          // 0 (tests): no first-20 pruning - a ray with more hits than its list holds goes to the sequential fallback
static std::atomic<int> g_sampler_mode{2};         // 0: one lane per ray, sequential walk (k_sample); 1: step-parallel (k_sample_par); 2: by ray count
__device__ long long* g_isect_dbg = nullptr;            // optional [blocks][8] stamps of thread 0 (profiling aid, nl_geometry_set_debug_buffer)
#define ISTAMP(k, v) do { if (g_isect_dbg && threadIdx.x == 0) g_isect_dbg[(size_t)blockIdx.x * 8 + (k)] = (long long)(v); } while (0)
#define SSTAMP(k) do { if (g_isect_dbg && threadIdx.x == 0) g_isect_dbg[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)

// First entry of a ray's work-list: the root's box, then the single-child chain under it (pack_children_blocks: the pseudo block's id
// slots hold its length, octants and end) tested in registers with the arithmetic of an ordinary expansion - what the traversal would
// do level by level, without the memory round trip per level.  false: the ray misses the tree.
__device__ __forceinline__ bool iq_first_entry(const float o[3], const float inv[3], const int2* __restrict__ blk_hdr, const int4* __restrict__ blk_ids,
                                               int root_side, float voxel_size, float half_voxel, int4* entry)
{
    const int4 c0 = blk_ids[0], c1 = blk_ids[1];                    // ids of the pseudo block: (root, n_chain, octants lo, octants hi | end block, x, y, z)
    const int2 h0 = blk_hdr[0];
    const float fs = (float)root_side, hs = fs * 0.5f;
    float tn, tf;
    if (!(slab_inv(o, inv, hs * voxel_size, hs * voxel_size, hs * voxel_size, half_voxel * fs, &tn, &tf) && h0.x >= 0)) return false;
    int px = 0, py = 0, pz = 0, cs = root_side >> 1;
    const int n_chain = c0.y;
    for (int l = 0; l < n_chain; ++l) {
        const int u = (int)((((unsigned long long)(unsigned)c0.w << 30) | (unsigned long long)(unsigned)c0.z) >> (3 * l)) & 7;
        const float fcs = (float)cs;
        ChildSlabs slabs;
        slabs.init(o, inv, px, py, pz, cs, fcs * 0.5f, voxel_size, half_voxel * fcs);
        if (!slabs.hit(u, &tn, &tf)) return false;
        px += (u & 1) ? cs : 0; py += (u & 2) ? cs : 0; pz += (u & 4) ? cs : 0;
        cs >>= 1;
    }
    *entry = make_int4(n_chain ? c1.x : h0.x, px, py, pz | ((31 - __clz(cs)) << 20));
    return true;
}

// More than 20 hits on a ray's list (k_ray_intersect_q): its lanes rank them in DFS order (every lane its own entries against all), keep
// the first 20 in that order, the 20th becomes the ray's pruning threshold.  Out of line: it is the rare path of a latency-bound loop.
template <int IQ_LPR, int IQ_HCAP>
__device__ __noinline__ void iq_compact(int* s_hid, int* s_hx, int* s_hy, int* s_hz, float* s_ht0, float* s_ht1, int* thr, int4* st, int a0, int nh_now, int j)
{
    constexpr int IQ_EPL = (IQ_HCAP + IQ_LPR - 1) / IQ_LPR;         // list entries per lane of a ray
    int eid[IQ_EPL], ex[IQ_EPL], ey[IQ_EPL], ez[IQ_EPL], erk[IQ_EPL]; float e0[IQ_EPL], e1[IQ_EPL];
#pragma unroll
    for (int t = 0; t < IQ_EPL; ++t) {
        const int i = j + t * IQ_LPR;
        erk[t] = IQ_HCAP;
        if (i < nh_now) {
            eid[t] = s_hid[a0 + i]; ex[t] = s_hx[a0 + i]; ey[t] = s_hy[a0 + i]; ez[t] = s_hz[a0 + i]; e0[t] = s_ht0[a0 + i]; e1[t] = s_ht1[a0 + i];
            int rk = 0;
            for (int q = 0; q < nh_now; ++q) rk += dfs_before(s_hx[a0 + q], s_hy[a0 + q], s_hz[a0 + q], ex[t], ey[t], ez[t]) ? 1 : 0;
            erk[t] = rk;                                            // voxels are distinct: the ranks are a permutation of 0 .. nh - 1
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                               // every lane of the ray holds its entries in registers
#pragma unroll
    for (int t = 0; t < IQ_EPL; ++t) {
        if (erk[t] < NL_MAX_HITS) {
            const int a = a0 + erk[t];
            s_hid[a] = eid[t]; s_hx[a] = ex[t]; s_hy[a] = ey[t]; s_hz[a] = ez[t]; s_ht0[a] = e0[t]; s_ht1[a] = e1[t];
            if (erk[t] == NL_MAX_HITS - 1) { thr[1] = ex[t]; thr[2] = ey[t]; thr[3] = ez[t]; st->w = 1; }
        }
    }
    if (j == 0) st->y = NL_MAX_HITS;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int IQ_LPR>
__global__ __launch_bounds__(NL_GEO_THREADS) void k_ray_intersect_q(
    int N, const float* __restrict__ rays_d_sensor, const float* __restrict__ points_gt,
    const float* __restrict__ cos_gt, const int* __restrict__ frame_id, const float* __restrict__ poses,
    const int2* __restrict__ blk_hdr, const int4* __restrict__ blk_ids, int root_side,
    float voxel_size, float max_distance,
    float* __restrict__ rays_d_world, float* __restrict__ gt_dist,
    int* __restrict__ hit_idx, float* __restrict__ hit_t0, float* __restrict__ hit_t1,
    int* __restrict__ hit_count, int* __restrict__ counters, int* __restrict__ ovf_list, int prune_flags)
{
    const int prune = prune_flags & 1;
    int n_compact = 0;
    constexpr int IQ_RAYS = NL_GEO_THREADS / IQ_LPR, IQ_QCAP = IqCaps<IQ_LPR>::Q, IQ_HCAP = IqCaps<IQ_LPR>::H;
    __shared__ int4 s_q[IQ_RAYS * IQ_QCAP];
    __shared__ int s_hid[IQ_RAYS * IQ_HCAP], s_hx[IQ_RAYS * IQ_HCAP], s_hy[IQ_RAYS * IQ_HCAP], s_hz[IQ_RAYS * IQ_HCAP];
    __shared__ float s_ht0[IQ_RAYS * IQ_HCAP], s_ht1[IQ_RAYS * IQ_HCAP];
    // per-ray state in ONE 16-byte word, read once per round (the traversal is a latency chain: every dependent LDS access costs a
    // round trip): x = stack height, y = hits on the list, z = overflow flag, w = pruning threshold set
    __shared__ int4 s_st[IQ_RAYS];
    __shared__ int s_thr[IQ_RAYS * 4];                              // (-, x, y, z): the ray's 20th hit in DFS order so far
#define s_tail(r_) s_st[r_].x
#define s_nh(r_) s_st[r_].y
#define s_ovf(r_) s_st[r_].z
#define s_thron(r_) s_st[r_].w
    __shared__ int s_hmax;
    if (threadIdx.x == 0) s_hmax = 0;
    ISTAMP(0, __builtin_readcyclecounter());
    int rounds = 0;
    const int rl = threadIdx.x / IQ_LPR, j = threadIdx.x % IQ_LPR;
    const int r = blockIdx.x * IQ_RAYS + rl;
    const bool live = r < N;
    float o[3] = {0.f, 0.f, 0.f}, inv[3] = {1.f, 1.f, 1.f};
    const float half_voxel = voxel_size * 0.5f;
    if (j == 0) s_st[rl] = make_int4(0, 0, 0, 0);
    if (live) {
        const float* P = poses + 12 * (frame_id ? frame_id[r] : 0);
        const float s0 = rays_d_sensor[3 * r], s1 = rays_d_sensor[3 * r + 1], s2 = rays_d_sensor[3 * r + 2];
        float d[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[i] = (s0 * P[3 * i] + s1 * P[3 * i + 1]) + s2 * P[3 * i + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) { o[i] = P[9 + i]; inv[i] = 1.0f / d[i]; }
        if (j == 0) {
            rays_d_world[3 * r] = d[0]; rays_d_world[3 * r + 1] = d[1]; rays_d_world[3 * r + 2] = d[2];
            const float gx = points_gt[3 * r], gy = points_gt[3 * r + 1], gz = points_gt[3 * r + 2];
            gt_dist[r] = sqrtf((gx * gx + gy * gy) + gz * gz) * cos_gt[r];
        }
    }
    {
        // first work-list entry (iq_first_entry's tests, spread over the ray's lanes: lane j takes chain levels j, j + LPR, ... - in one
        // lane the nine levels of a one-scan map are a 13 k-cycle dependent chain of slab arithmetic, a third of the kernel at 2048 rays)
        const int4 c0 = blk_ids[0], c1 = blk_ids[1];
        const int2 h0 = blk_hdr[0];
        const int n_chain = c0.y;
        const unsigned long long octs = ((unsigned long long)(unsigned)c0.w << 30) | (unsigned long long)(unsigned)c0.z;
        bool ok = live;
        if (live) {
            const float fs = (float)root_side, hs = fs * 0.5f;
            float tn, tf;
            ok = slab_inv(o, inv, hs * voxel_size, hs * voxel_size, hs * voxel_size, half_voxel * fs, &tn, &tf) && h0.x >= 0;
            for (int l = j; l < n_chain && ok; l += IQ_LPR) {
                int px = 0, py = 0, pz = 0, cs = root_side >> 1;
                for (int i = 0; i < l; ++i) {                       // lattice position of level l's node: the octants above it
                    const int u = (int)(octs >> (3 * i)) & 7;
                    px += (u & 1) ? cs : 0; py += (u & 2) ? cs : 0; pz += (u & 4) ? cs : 0;
                    cs >>= 1;
                }
                const float fcs = (float)cs;
                ChildSlabs slabs;
                slabs.init(o, inv, px, py, pz, cs, fcs * 0.5f, voxel_size, half_voxel * fcs);
                ok = slabs.hit((int)(octs >> (3 * l)) & 7, &tn, &tf);
            }
        }
#pragma unroll
        for (int off = 1; off < IQ_LPR; off <<= 1) ok = (__shfl_xor((int)ok, off) != 0) && ok;        // every lane of the ray: all levels hit
        if (live && j == 0 && ok) {
            const int cs_end = root_side >> (1 + n_chain);
            s_q[rl * IQ_QCAP] = make_int4(n_chain ? c1.x : h0.x, n_chain ? c1.y : 0, n_chain ? c1.z : 0, (n_chain ? c1.w : 0) | ((31 - __clz(cs_end)) << 20));
            s_tail(rl) = 1;
        }
    }
    __syncthreads();
    // The pending expansions form a STACK (s_tail = height): the four lanes always take the deepest pending nodes, which
    // keeps the pending set as small as a DFS with all siblings pushed (a line meets at most 4 of a node's 8 children,
    // so <= 3 per level + 4), while still giving four independent memory round trips per ray per round.
    ISTAMP(1, __builtin_readcyclecounter());
    bool strict = false;
    // The rounds of the traversal are a latency chain, so the loop that runs them holds nothing else: a ray that needs one of the two
    // rare services leaves it (wave-uniform exit), is served below, and the rounds resume.
    for (;;) {
    bool finished = false;
    int4 st;
    int ovf_now, nh_now, height;
    bool thr_on;
    for (;;) {
        st = s_st[rl];
        ovf_now = live ? st.z : 1;
        nh_now = !ovf_now ? st.y : 0;
        thr_on = live && st.w != 0;
        height = !ovf_now ? st.x : 0;
        if (__any(prune && live && ((ovf_now == 1 && !strict) || nh_now > NL_MAX_HITS))) break;       // a ray of this wave needs service
        if (!__any(height > 0)) { finished = true; break; }        // wave-uniform: all rays of this wave are done
        ++rounds;
        // nodes popped this round: all the lanes can take, but never more than can push four children each (a line meets at most four of a
        // node's eight octants: 4 k <= QCAP - height + k) - a stack that runs over costs the ray a second traversal
        int k = height < IQ_LPR ? height : IQ_LPR;
        if (strict) { const int room = (IQ_QCAP - height) / 3; k = k < room ? k : (room > 1 ? room : 1); }
        else if (prune) { const int room = (IQ_QCAP - height) / 2; k = k < room ? k : (room > 1 ? room : 1); }   // (first attempt: three children per node on average)
        // ... and on the first attempt a LEAF PARENT is expanded only by the lanes the hit list has room for at four hits each; the others put
        // their node back (it keeps its place in the DFS order) - a list that runs over costs the ray a second traversal as well
        const int hit_lanes = (IQ_HCAP - nh_now) / 3;
        const bool mine = j < k;
        int4 e = make_int4(0, 0, 0, 0);
        if (mine) e = s_q[rl * IQ_QCAP + height - 1 - j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                           // every lane of the group has fetched its entry
        // pushes of this round (filled below, issued after the block): the children a lane's node expands into, or - careful mode - the
        // leaf parent itself when its hits found no room
        unsigned pm = 0u; int pbase = 0, ppx = 0, ppy = 0, ppz = 0, pcs = 0, pcsl = 0; unsigned phas = 0u; bool repush = false;
        if (mine) {
            const int b = e.x, px = e.y, py = e.z, pz = e.w & 0xFFFFF, csl = e.w >> 20, cs = 1 << csl;
            // leaf-parent blocks also need their node ids: both loads depend on b only, so issue them together (inside the
            // `cs == 1` branch below the ids would wait for the header's round trip first)
            int4 ia = make_int4(-1, -1, -1, -1), ib = ia;
            if (cs == 1) { ia = blk_ids[2 * (size_t)b]; ib = blk_ids[2 * (size_t)b + 1]; }
            const int2 hdr = blk_hdr[b];
            const unsigned exist = (unsigned)hdr.y & 255u, has = ((unsigned)hdr.y >> 8) & 255u;
            const float fs = (float)cs, hs = fs * 0.5f, half = half_voxel * fs;
            ChildSlabs slabs;
            slabs.init(o, inv, px, py, pz, cs, hs, voxel_size, half);
            // children that can still matter: with a threshold (the ray's 20th hit in DFS order so far) a voxel behind it can never be
            // among the first 20, and neither can anything below a node whose DFS-first voxel - its max corner - lies behind it
            unsigned keep = cs == 1 ? exist : has;
            if (thr_on) {
                const int tx = s_thr[4 * rl + 1], ty = s_thr[4 * rl + 2], tz = s_thr[4 * rl + 3];
                const int top = cs - 1;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int cx = px + ((u & 1) ? cs : 0) + top, cy = py + ((u & 2) ? cs : 0) + top, cz = pz + ((u & 4) ? cs : 0) + top;
                    if (!dfs_before(cx, cy, cz, tx, ty, tz)) keep &= ~(1u << u);
                }
            }
            if (cs == 1) {
                const int ids[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
                if (!strict && prune && !thr_on && j >= hit_lanes && keep != 0u) repush = true;   // (with a threshold few voxels pass: no deferral)
                else if (!strict) {
                    // first attempt: one slot per hit; a list that runs over all the same (a degenerate ray touching more than four
                    // octants of a node) starts the ray again in the careful mode below
#pragma unroll
                    for (int u = 7; u >= 0; --u) {
                        if (!((keep >> u) & 1u) || ids[u] < 0) continue;
                        float tn, tf;
                        if (slabs.hit(u, &tn, &tf)) {
                            const int slot = atomicAdd(&s_nh(rl), 1);
                            if (slot < IQ_HCAP) {
                                const int a = rl * IQ_HCAP + slot;
                                s_hid[a] = ids[u]; s_ht0[a] = tn; s_ht1[a] = tf;
                                s_hx[a] = px + (u & 1); s_hy[a] = py + ((u >> 1) & 1); s_hz[a] = pz + ((u >> 2) & 1);
                            } else s_ovf(rl) = 1;
                        }
                    }
                } else {
                    unsigned hm = 0u;
#pragma unroll
                    for (int u = 7; u >= 0; --u) {
                        if (!((keep >> u) & 1u) || ids[u] < 0) continue;
                        float tn, tf;
                        if (slabs.hit(u, &tn, &tf)) hm |= 1u << u;
                    }
                    const int c = __popc(hm);
                    if (c > 0) {
                        // room for all of them, or none (no holes in the list): reserve with a compare-and-swap
                        int base = -1, seen = nh_now > NL_MAX_HITS ? NL_MAX_HITS : nh_now;     // (a guess: the compare-and-swap corrects it)
                        while (seen + c <= IQ_HCAP) {
                            const int prev = atomicCAS(&s_nh(rl), seen, seen + c);
                            if (prev == seen) { base = seen; break; }
                            seen = prev;
                        }
                        if (base >= 0) {
#pragma unroll
                            for (int u = 7; u >= 0; --u) {
                                if (!((hm >> u) & 1u)) continue;
                                const int a = rl * IQ_HCAP + base++;
                                float tn, tf;
                                slabs.hit(u, &tn, &tf);              // (same operations on the same values as in the counting pass)
                                s_hid[a] = ids[u]; s_ht0[a] = tn; s_ht1[a] = tf;
                                s_hx[a] = px + (u & 1); s_hy[a] = py + ((u >> 1) & 1); s_hz[a] = pz + ((u >> 2) & 1);
                            }
                        } else repush = true;      // the list is full until the next round's compaction (>= 8 free slots then): back on the stack
                    }
                }
            } else {
                // (push order: ascending octants would make the stack pop the highest octant first, i.e. run ahead in the reference's DFS
                //  order and find the final first 20 early; measured on the 150-scan map with a run-time switch, since removed -
                //  profiles/r03_c_intersect_probe_large.txt - it makes no systematic difference with 16 lanes popping at once)
#pragma unroll
                for (int u = 7; u >= 0; --u) {
                    if (!((keep >> u) & 1u)) continue;
                    float tn, tf;
                    if (slabs.hit(u, &tn, &tf)) pm |= 1u << u;
                }
                pbase = hdr.x; phas = has; ppx = px; ppy = py; ppz = pz; pcs = cs; pcsl = csl;
            }
        }
        // The pushes keep the stack in the reference's DFS ORDER (top = next): the lane that popped the top entry (j = 0) puts its
        // children on top of those of lane 1, and so on; inside a node the highest octant ends on top (the reference pops 7 .. 0).
        // Slots come from a suffix sum over the ray's lanes instead of one returning LDS atomic per child (each a round trip on the
        // traversal's latency chain, and their arrival order scrambled the stack: the first-20 pruning then explored subtrees far behind
        // the final threshold before the near ones had tightened it - the tail of the accumulated-map regime).
        {
            const int c = __popc(pm) + (repush ? 1 : 0);
            int suf = c;                                            // inclusive suffix sum over the lanes j .. LPR - 1 of this ray: DPP row shifts
            // (a ray's lanes lie inside one 16-lane row; lane i reads lane i + off, zero beyond the row - four VALU instructions instead of
            //  four cross-lane LDS permutes on the traversal's latency chain)
            { const int t = __builtin_amdgcn_mov_dpp(suf, 0x101 /* row_shl:1 */, 0xF, 0xF, true); if (j + 1 < IQ_LPR) suf += t; }
            if (IQ_LPR > 2) { const int t = __builtin_amdgcn_mov_dpp(suf, 0x102 /* row_shl:2 */, 0xF, 0xF, true); if (j + 2 < IQ_LPR) suf += t; }
            if (IQ_LPR > 4) { const int t = __builtin_amdgcn_mov_dpp(suf, 0x104 /* row_shl:4 */, 0xF, 0xF, true); if (j + 4 < IQ_LPR) suf += t; }
            if (IQ_LPR > 8) { const int t = __builtin_amdgcn_mov_dpp(suf, 0x108 /* row_shl:8 */, 0xF, 0xF, true); if (j + 8 < IQ_LPR) suf += t; }
            if (IQ_LPR > 16) {                                      // two rows per ray: the lower row adds the upper row's total (a cross-lane read)
                const int upper = __shfl(suf, (int)(threadIdx.x & 63u & ~31u) + 16);
                if (j < 16) suf += upper;
            }
            const int total = suf;                                  // (lane 0 of the ray: the ray's total; other lanes: what lies at or below them)
            const int base = height - k;
            if (base + suf > IQ_QCAP) {                             // this lane's slots (or some below them) lie beyond the stack: the ray starts again
                if (c > 0) s_ovf(rl) = repush ? 2 : 1;
            } else {
                int slot = rl * IQ_QCAP + base + (suf - c);
                if (repush) s_q[slot++] = e;
#pragma unroll 1
                for (unsigned rest = pm; rest != 0u; rest &= rest - 1u) {        // ascending octants: at most four for a line
                    const int u = __ffs((int)rest) - 1;
                    const int cx = ppx + ((u & 1) ? pcs : 0), cy = ppy + ((u & 2) ? pcs : 0), cz = ppz + ((u & 4) ? pcs : 0);
                    s_q[slot++] = make_int4(pbase + __popc(phas & ((1u << u) - 1u)), cx, cy, cz | ((pcsl - 1) << 20));
                }
            }
            if (j == 0 && height > 0) s_tail(rl) = base + total;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (finished) break;
    // (1) The pending stack (or, on the first attempt, the hit list) of a ray overflowed - all its lanes popping, each node pushing
    // up to four children: start the ray again, this time popping only as many nodes per round as can push four children each and
    // reserving list slots before writing them.  That never overflows unless the stack is too small for the ray one node at a time
    // (then: the sequential fallback).  Uniform over the lanes of a ray.
    if (ovf_now == 1 && prune && live && !strict) {
        strict = true;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                           // every lane of the ray has seen the flags
        if (j == 0) {
            s_st[rl] = make_int4(0, 0, 0, 0);
            int4 e0;
            if (iq_first_entry(o, inv, blk_hdr, blk_ids, root_side, voxel_size, half_voxel, &e0)) { s_q[rl * IQ_QCAP] = e0; s_tail(rl) = 1; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else if (nh_now > NL_MAX_HITS && prune) {
        // (2) more than 20 hits on the list: rank them in DFS order, keep the first 20, the 20th becomes the pruning threshold
        ++n_compact;
        iq_compact<IQ_LPR, IQ_HCAP>(s_hid, s_hx, s_hy, s_hz, s_ht0, s_ht1, s_thr + 4 * rl, &s_st[rl], rl * IQ_HCAP, nh_now, j);
    }
    }
    ISTAMP(2, __builtin_readcyclecounter()); ISTAMP(4, rounds); ISTAMP(5, strict ? 1 : 0); ISTAMP(6, n_compact);
    // finalise, all lanes of the ray (one lane sorting a ray's list by insertion was up to 22 k cycles, a quarter of the kernel at 2048 rays):
    // every lane ranks its entries - (1) DFS order = descending z-major Morton order of the voxel, the first 20 stay (the reference's cap);
    // (2) among those, stable by t_min (voxel_helpers.py:546: equal t_min keep their DFS order) - and writes them straight to their place in
    // the ray's row, (3) culled (voxel_helpers.py:549-552).
    int valid = 0;
    {
        constexpr int IQ_EPL = (IQ_HCAP + IQ_LPR - 1) / IQ_LPR;
        const int nh = live ? s_nh(rl) : 0;
        const bool ovf = live && (s_ovf(rl) != 0 || nh > IQ_HCAP);
        if (ovf && j == 0) {
            ovf_list[atomicAdd(&counters[NLC_ISECT_OVF], 1)] = r;
            hit_count[r] = 0;                                       // rewritten by the DFS fallback pass
        }
        const int n = ovf ? 0 : nh;
        const int a0 = rl * IQ_HCAP;
        int eid[IQ_EPL], r1[IQ_EPL]; float e0[IQ_EPL], e1[IQ_EPL];
#pragma unroll
        for (int t = 0; t < IQ_EPL; ++t) {
            const int i = j + t * IQ_LPR;
            r1[t] = IQ_HCAP;
            if (i < n) {
                const int x = s_hx[a0 + i], y = s_hy[a0 + i], z = s_hz[a0 + i];
                eid[t] = s_hid[a0 + i]; e0[t] = s_ht0[a0 + i]; e1[t] = s_ht1[a0 + i];
                int rk = 0;
                for (int q = 0; q < n; ++q) rk += dfs_before(s_hx[a0 + q], s_hy[a0 + q], s_hz[a0 + q], x, y, z) ? 1 : 0;
                r1[t] = rk;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                           // every lane of the ray is done with the coordinates
#pragma unroll
        for (int t = 0; t < IQ_EPL; ++t) { const int i = j + t * IQ_LPR; if (i < n) s_hx[a0 + i] = r1[t]; }      // DFS ranks, for the others' tie-breaks
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int* oi = hit_idx + (size_t)r * NL_MAX_HITS; float* o0 = hit_t0 + (size_t)r * NL_MAX_HITS; float* o1 = hit_t1 + (size_t)r * NL_MAX_HITS;
#pragma unroll
        for (int t = 0; t < IQ_EPL; ++t) {
            if (r1[t] >= NL_MAX_HITS) continue;                     // (also: no entry)
            int pos = 0;
            for (int q = 0; q < n; ++q) {
                const int rq = s_hx[a0 + q];
                const float tq = s_ht0[a0 + q];
                pos += (rq < NL_MAX_HITS && (tq < e0[t] || (tq == e0[t] && rq < r1[t]))) ? 1 : 0;
            }
            const bool keep = !(e1[t] > 2.0f * max_distance) && !(e0[t] > max_distance);
            if (keep) { ++valid; oi[pos] = eid[t]; o0[pos] = e0[t]; o1[pos] = e1[t]; }
            else { oi[pos] = -1; o0[pos] = max_distance; o1[pos] = max_distance; }
        }
#pragma unroll
        for (int off = 1; off < IQ_LPR; off <<= 1) valid += __shfl_xor(valid, off);     // every lane of the ray: the ray's count
        if (live && !ovf && j == 0) hit_count[r] = valid;
        if (j != 0) valid = 0;
    }
    ISTAMP(3, __builtin_readcyclecounter());
    int wmax = valid;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&s_hmax, wmax);
    __syncthreads();
    if (threadIdx.x == 0 && s_hmax > 0) atomicMax(&counters[NLC_HMAX], s_hmax);
}
#undef s_tail
#undef s_nh
#undef s_ovf
#undef s_thron

// ---------------------------------------------------------------------------------------------
// Exclusive scan (int32), n <= 1024*1024: per-block (1024 items) scan + block sums, one block
// scans the sums, third pass adds.  `flag_mode` scans (in[i] > 0) instead of in[i].
// ---------------------------------------------------------------------------------------------
#define NL_SCAN_ITEMS 4
#define NL_SCAN_BLOCK (NL_GEO_THREADS * NL_SCAN_ITEMS)

__device__ __forceinline__ int block_exclusive_scan(int v, int* s_wave, int* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < NL_GEO_THREADS / 64; ++w) { int t = s_wave[w]; if (w < wid) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(NL_GEO_THREADS) void k_scan_blocks(const int* __restrict__ in, int* __restrict__ out,
                                                                 int* __restrict__ block_sums, int n, int flag_mode)
{
    __shared__ int s_wave[NL_GEO_THREADS / 64];
    const int base = blockIdx.x * NL_SCAN_BLOCK + threadIdx.x * NL_SCAN_ITEMS;
    int v[NL_SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < NL_SCAN_ITEMS; ++i) {
        int x = (base + i < n) ? in[base + i] : 0;
        if (flag_mode) x = x > 0 ? 1 : 0;
        v[i] = x; sum += x;
    }
    int tot;
    int ex = block_exclusive_scan(sum, s_wave, &tot);
#pragma unroll
    for (int i = 0; i < NL_SCAN_ITEMS; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// Second (last) pass of the large scan: block b adds the sum of the block sums before it (at most a few hundred values,
// reduced by the block itself: no separate "scan of the sums" launch), optionally writes the hit-ray compaction
// ray_of_rank[rank] = ray (render_helpers.py:219-227) and the grand total to up to two places.
__global__ __launch_bounds__(NL_GEO_THREADS) void k_scan_finish(const int* __restrict__ in, int* __restrict__ out,
                                                                 const int* __restrict__ block_sums, int n, int* __restrict__ ray_of_rank,
                                                                 int* __restrict__ total_out, int* __restrict__ total_out2)
{
    __shared__ int s_wave[NL_GEO_THREADS / 64];
    int part = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += NL_GEO_THREADS) part += block_sums[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = part;
    __syncthreads();
    int add = 0;
#pragma unroll
    for (int w = 0; w < NL_GEO_THREADS / 64; ++w) add += s_wave[w];
    const int base = blockIdx.x * NL_SCAN_BLOCK + threadIdx.x * NL_SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < NL_SCAN_ITEMS; ++i) {
        if (base + i < n) {
            const int r = out[base + i] + add;
            out[base + i] = r;
            if (ray_of_rank && in[base + i] > 0) ray_of_rank[r] = base + i;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int tot = add + block_sums[blockIdx.x];
        *total_out = tot;
        if (total_out2) *total_out2 = tot;
    }
}

// Whole scan in ONE launch for small inputs (tracking: 2048 rays): one block walks the input in chunks of 1024 with a carry,
// ~1 us per chunk behind a ~4.8 us launch; the two-launch version costs 2 x 4.8 us at any of these sizes, so it wins from
// 5 chunks on (16 384 rays: 20.8 us in one block, profiles/r01_m_timeline_latency_bound_steps.txt).
#define NL_SCAN_ONE_BLOCK_MAX NL_RAYS_ONE_WORKGROUP_SCAN
__device__ __forceinline__ int scan_one_block(const int* __restrict__ in, int* __restrict__ out, int n, int flag_mode,
                                              int* __restrict__ ray_of_rank, int* __restrict__ total_out, int* __restrict__ total_out2)
{
    __shared__ int s_wave[NL_GEO_THREADS / 64];
    int carry = 0;
    for (int c0 = 0; c0 < n; c0 += NL_SCAN_BLOCK) {
        const int base = c0 + threadIdx.x * NL_SCAN_ITEMS;
        int v[NL_SCAN_ITEMS], sum = 0;
#pragma unroll
        for (int i = 0; i < NL_SCAN_ITEMS; ++i) {
            int x = (base + i < n) ? in[base + i] : 0;
            if (flag_mode) x = x > 0 ? 1 : 0;
            v[i] = x; sum += x;
        }
        int tot;
        int ex = block_exclusive_scan(sum, s_wave, &tot) + carry;
#pragma unroll
        for (int i = 0; i < NL_SCAN_ITEMS; ++i) {
            if (base + i < n) {
                out[base + i] = ex;
                if (ray_of_rank && v[i] > 0) ray_of_rank[ex] = base + i;
            }
            ex += v[i];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) { *total_out = carry; if (total_out2) *total_out2 = carry; }
    return carry;
}
__global__ __launch_bounds__(NL_GEO_THREADS) void k_scan_one_block(const int* __restrict__ in, int* __restrict__ out, int n, int flag_mode,
                                                                    int* __restrict__ ray_of_rank, int* __restrict__ total_out,
                                                                    int* __restrict__ total_out2)
{
    scan_one_block(in, out, n, flag_mode, ray_of_rank, total_out, total_out2);
}
// Launch-bound regime (<= NL_SCAN_ONE_BLOCK_MAX rays: the reference's live shapes): the hit-ray scan also runs the DFS fallback of
// the rays the work-list intersect kernel handed over - a launch of its own otherwise, for a list that is empty on ordinary scans
__global__ __launch_bounds__(NL_GEO_THREADS) void k_scan_hits_fused(DfsArgs a, int* __restrict__ hit_rank, int* __restrict__ ray_of_rank,
                                                                     int* __restrict__ total_out, int* __restrict__ total_out2)
{
    if (a.counters[NLC_ISECT_OVF] > 0) {                 // uniform over the workgroup
        dfs_rays(a, 0, 1);
        __threadfence_block();
        __syncthreads();
    }
    scan_one_block(a.hit_count, hit_rank, a.N, 1, ray_of_rank, total_out, total_out2);
}

// Mid-size inputs (NL_SCAN_ONE_BLOCK_MAX < n <= NL_SCAN_SINGLE_MAX: a rank's share of a scan, 4096-ray x 4-frame bundle adjustment) as ONE
// launch of up to 8 workgroups of 1024 threads: a workgroup first sums the (flagged) items of all workgroups in front of it itself - at most
// 7 x 16 KB out of L2, seven 16-byte loads per thread - then scans its own 4096.  The two-launch version pays a second ~4.9 us launch for
// the same result; the redundant reads cost ~1 us.  The launch's LAST workgroup knows the grand total and can do what a launch of its own
// did before: the loss normalisers (nl_scan_samples_finalize) or the send block of exchange 1 (nl_dist_x1_pack: one byte per ray = its hit
// count, clamped to 255, + the counter block), whose bytes every workgroup packs from the items it holds anyway.
#define NL_SCAN1_THREADS 1024
#define NL_SCAN1_BLOCK (NL_SCAN1_THREADS * NL_SCAN_ITEMS)
#define NL_SCAN_SINGLE_MAX NL_RAYS_SINGLE_LAUNCH_SCAN
__device__ __forceinline__ void loss_finalize_one(int* __restrict__ counters, NlLossScalars* __restrict__ ls,
                                                  float fs_weight, float sdf_weight, float tau, float max_depth, int capacity);
struct ScanTail {
    int* counters; NlLossScalars* ls; float fs_weight, sdf_weight, tau, max_depth; int capacity; int finalize;   // finalize: loss_finalize_one by the last workgroup
    int* pack_send; int pack_cap;                                                                               // pack_send: [counter block | pack_cap bytes]
};
__global__ __launch_bounds__(NL_SCAN1_THREADS) void k_scan_single(const int* __restrict__ in, int* __restrict__ out, int n, int flag_mode,
                                                                   int* __restrict__ ray_of_rank, int* total_out, int* total_out2,      // (may point into t.counters)
                                                                   ScanTail t)
{
    __shared__ int s_wave[NL_SCAN1_THREADS / 64], s_part[NL_SCAN1_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int base = b * NL_SCAN1_BLOCK + tid * NL_SCAN_ITEMS;
    int raw[NL_SCAN_ITEMS];
    if (base + NL_SCAN_ITEMS <= n) {
        const int4 q = *reinterpret_cast<const int4*>(in + base);
        raw[0] = q.x; raw[1] = q.y; raw[2] = q.z; raw[3] = q.w;
    } else {
#pragma unroll
        for (int i = 0; i < NL_SCAN_ITEMS; ++i) raw[i] = base + i < n ? in[base + i] : 0;
    }
    int part = 0;                                                   // items of the workgroups in front (all of them whole: b * 4096 < n)
    for (int j = tid * NL_SCAN_ITEMS; j < b * NL_SCAN1_BLOCK; j += NL_SCAN1_BLOCK) {
        const int4 q = *reinterpret_cast<const int4*>(in + j);
        part += flag_mode ? (q.x > 0) + (q.y > 0) + (q.z > 0) + (q.w > 0) : q.x + q.y + q.z + q.w;
    }
    int v[NL_SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < NL_SCAN_ITEMS; ++i) { v[i] = flag_mode ? (raw[i] > 0 ? 1 : 0) : raw[i]; sum += v[i]; }
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off); if (lane >= off) inc += u; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 63) s_wave[wid] = inc;
    if (lane == 0) s_part[wid] = part;
    __syncthreads();
    int before = 0, tot = 0, add = 0;
#pragma unroll
    for (int w = 0; w < NL_SCAN1_THREADS / 64; ++w) { const int u = s_wave[w]; if (w < wid) before += u; tot += u; add += s_part[w]; }
    int ex = add + before + inc - sum;
#pragma unroll
    for (int i = 0; i < NL_SCAN_ITEMS; ++i) {
        if (base + i < n) {
            out[base + i] = ex;
            if (ray_of_rank && v[i] > 0) ray_of_rank[ex] = base + i;
        }
        ex += v[i];
    }
    unsigned* const send_bytes = t.pack_send ? reinterpret_cast<unsigned*>(t.pack_send + NL_CNT_INTS + 2 * NL_CNT_DOUBLES) : nullptr;
    if (send_bytes && base < t.pack_cap) {                          // (pack_cap is a multiple of 16: a thread's four bytes are inside or outside together)
        unsigned w = 0u;
#pragma unroll
        for (int i = 0; i < NL_SCAN_ITEMS; ++i) w |= (unsigned)(raw[i] < 0 ? 0 : (raw[i] > 255 ? 255 : raw[i])) << (8 * i);
        send_bytes[base >> 2] = w;
    }
    if (b != (int)gridDim.x - 1) return;
    if (send_bytes)                                                 // rays of the send block beyond this launch's items (another rank's shard is longer)
        for (int j = (int)gridDim.x * NL_SCAN1_BLOCK + tid * NL_SCAN_ITEMS; j < t.pack_cap; j += NL_SCAN1_BLOCK) send_bytes[j >> 2] = 0u;
    if (tid == 0) {
        const int total = add + tot;
        *total_out = total;
        if (total_out2) *total_out2 = total;
        if (t.finalize) loss_finalize_one(t.counters, t.ls, t.fs_weight, t.sdf_weight, t.tau, t.max_depth, t.capacity);
    }
    if (t.pack_send) {
        __threadfence_block();
        __syncthreads();                                            // the totals are in the counter block
        if (tid < NL_CNT_INTS + 2 * NL_CNT_DOUBLES) t.pack_send[tid] = reinterpret_cast<volatile const int*>(t.counters)[tid];
    }
}

// ray_of_rank[rank] = ray  for rays with hit_count > 0   (the reference's boolean-mask compaction
// of hit rays, render_helpers.py:219-227)
__global__ void k_compact_hit_rays(int N, const int* __restrict__ hit_count, const int* __restrict__ hit_rank,
                                   int* __restrict__ ray_of_rank)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < N && hit_count[r] > 0) ray_of_rank[hit_rank[r]] = r;
}

// ---------------------------------------------------------------------------------------------
// Sampler, pass 1 (count) and pass 2 (emit).  One hit ray per lane.  EMIT=false: per-ray sample
// count, S_max, and the loss normalisers that depend only on geometry (criterion.py:67-88):
// front/sdf mask counts over valid samples plus the per-ray constants of the padded slots
// (depth 80 fill, voxel_helpers.py:590).  EMIT=true: compacted (voxel, depth, dist, ray) records
// at samp_off[ray].
// ---------------------------------------------------------------------------------------------
struct SampleArgs {
    int N;
    const int* hit_idx; const float* hit_t0; const float* hit_t1;
    const int* hit_count; const int* hit_rank; const int* ray_of_rank;
    const float* cos_gt; const float* gt_dist;
    float step_size; float tau; float max_depth;
    unsigned seed; int use_hash_noise; int tail_always; int ray_id_base;
    const int* row_first;       // multi-GPU: [entries][1 + NL_MAX_HITS] = (count, 2 for bins below the count, 0 beyond) of every batch row's first ray (nl_dist_x1_merge), or NULL
    const unsigned* seed_mix;   // optional device word (the optimiser's step counter) folded into the seed: fresh jitter per iteration
    int* counters; double* dcounters;
    int* samp_count;            // [N]  (0 for rays without hits)
    const int* samp_off;        // [N]  exclusive scan of samp_count (emit pass)
    int capacity;
    int* s_vox; float* s_depth; float* s_dist; int* s_ray;
};

// ---------------------------------------------------------------------------------------------
// Step-parallel sampler: SP_LPR lanes per hit ray, one stratified step per lane per iteration (nl_walk_* in
// nl_device_math.h: the bin of a step depends on the step alone, samples land at known indices, so the steps are
// independent).  The sequential walk of k_sample costs ~700 cycles per sample on the critical path of a whole wave; here a
// ray's ~16-34 steps take 2-5 iterations.  Same arithmetic, same results bit for bit (host: tests/test_device_math_host.py).
// ---------------------------------------------------------------------------------------------
#define SP_LPR 8
#define SP_RAYS (NL_GEO_THREADS / SP_LPR)
static_assert(NL_MAX_HITS == 20 && SP_LPR == 8, "a ray's 20 hit-list entries: three per lane, as five 16-byte LDS reads");

// A ray's walk is a chain of dependent latencies, and the regime this kernel serves (a pose-refinement step: 2048 rays, 64
// workgroups) has nothing to hide them with, so the chain is kept short:
//   * every global load whose address does not depend on data is issued at the top (the ray's whole 20-entry row, its hit rank);
//     the two dependent levels of the closing loop's context (first ray of the batch row -> its hit list) are issued on the way
//     and land in LDS just before the closing loop, which then reads LDS instead of one global word per iteration;
//   * interval lengths and their quotients len / tot are computed by the ray's eight lanes (three divisions each instead of
//     twenty in one lane); the two sequential fp32 chains (tot, the CDF boundaries) run on registers in every lane;
//   * the bin search of a step compares against the 20 boundaries in registers (first b with !(cdf > cum[b]), +inf beyond the
//     usable intervals - the same bin as the linear search, which paid one LDS round trip per interval);
//   * a step needs the evaluation of the step before it: that is the neighbouring lane's result (one shuffle), not a second
//     evaluation.
// Same arithmetic in the same order as nl_walk_plan / nl_walk_eval / nl_walk_step / nl_walk_tail (nl_device_math.h), bit for bit.
struct SpRay {
    float cumr[NL_MAX_HITS];    // CDF boundary of interval b for b < nb, +inf beyond
    float tot, step;
    int nb, T, P;
    int guard;
    // context of the closing loop (the reference's tail quirk)
    int j_in_row, rays_in_row;
    int rf_val[3], rf_cnt, rf_bias;   // this lane's three entries of the row-first ray's hit list (raw), its length, the list's bias
};

struct SpLds { int* i; float* t0; float* t1; float* c; float* q; int* rf; };

__device__ __forceinline__ void sp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                               // a ray's lanes sit in one wave
}

// returns whether the ray has hits.  All eight lanes of the ray call it together.
__device__ __forceinline__ bool sp_setup(const SampleArgs& a, int r, int j, const SpLds& m, SpRay& s)
{
    const bool in_range = r < a.N;
    const int rr = in_range ? r : 0;
    // level 0: nothing here depends on loaded data
    const int nh = in_range ? a.hit_count[rr] : 0;
    const int rank_local = a.hit_rank[rr];
    int li[3]; float l0[3], l1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int l = j + SP_LPR * t;
        const size_t o = (size_t)rr * NL_MAX_HITS + (l < NL_MAX_HITS ? l : 0);
        li[t] = a.hit_idx[o]; l0[t] = a.hit_t0[o]; l1[t] = a.hit_t1[o];
    }
    const int P = a.counters[NLC_HMAX], Rg = a.counters[NLC_R_GLOBAL], Roff = a.counters[NLC_R_OFFSET], Rloc = a.counters[NLC_R];
    s.P = P;
    const bool live = nh > 0;
    float len[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int l = j + SP_LPR * t;
        const bool v = l < nh;                                     // row tails beyond the ray's own hits are padding
        const int i_ = v ? li[t] : -1;
        const float t0_ = v ? l0[t] : a.max_depth, t1_ = v ? l1[t] : a.max_depth;
        len[t] = (i_ == -1) ? 0.0f : (t1_ - t0_);
        if (l < NL_MAX_HITS) { m.i[l] = i_; m.t0[l] = t0_; m.t1[l] = t1_; m.q[l] = len[t]; }
    }
    // level 1 of the closing loop's context: which ray is the first of this ray's batch row
    int first_rank = 0;
    nl_sampler_layout(rank_local + Roff, Rg, &s.j_in_row, &s.rays_in_row, &first_rank);
    const int first_local = first_rank - Roff;
    const bool is_local = first_local >= 0 && first_local < Rloc;
    const bool want_rf = live && !a.tail_always;
    const int first_ray = (want_rf && is_local) ? a.ray_of_rank[first_local] : rr;
    sp_wave_sync();
    // tot: the sequential fp32 sum over the row (entries beyond the ray's hits add +0, as they do in the reference's padded rows)
    float L[NL_MAX_HITS]; int I[NL_MAX_HITS];
#pragma unroll
    for (int g = 0; g < NL_MAX_HITS / 4; ++g) {
        const float4 v = reinterpret_cast<const float4*>(m.q)[g];
        const int4 w = reinterpret_cast<const int4*>(m.i)[g];
        L[4 * g] = v.x; L[4 * g + 1] = v.y; L[4 * g + 2] = v.z; L[4 * g + 3] = v.w;
        I[4 * g] = w.x; I[4 * g + 1] = w.y; I[4 * g + 2] = w.z; I[4 * g + 3] = w.w;
    }
    float tot = 0.0f;
#pragma unroll
    for (int l = 0; l < NL_MAX_HITS; ++l) tot = tot + L[l];
    s.tot = tot;
    // nb (nl_walk_plan): the leading usable intervals among the first P; an invalid first interval still counts as one
    int nb = 0; bool run = true;
#pragma unroll
    for (int b = 0; b < NL_MAX_HITS; ++b) { run = run && b < P && I[b] != -1; nb += run ? 1 : 0; }
    if (I[0] == -1 && P > 0) nb = 1;
    s.nb = nb;
    // the quotients len / tot, three per lane; the chain c[b] = c[b-1] + q[b] again in every lane
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int l = j + SP_LPR * t; if (l < NL_MAX_HITS) m.c[l] = len[t] / tot; }
    sp_wave_sync();
    float c = 0.0f;
    float cb[NL_MAX_HITS];
#pragma unroll
    for (int g = 0; g < NL_MAX_HITS / 4; ++g) {
        const float4 v = reinterpret_cast<const float4*>(m.c)[g];
        cb[4 * g] = v.x; cb[4 * g + 1] = v.y; cb[4 * g + 2] = v.z; cb[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int b = 0; b < NL_MAX_HITS; ++b) { c = (b == 0) ? cb[0] : (c + cb[b]); cb[b] = c; s.cumr[b] = b < nb ? c : __builtin_inff(); }
    if (j == 0) {
#pragma unroll
        for (int g = 0; g < NL_MAX_HITS / 4; ++g) reinterpret_cast<float4*>(m.c)[g] = make_float4(cb[4 * g], cb[4 * g + 1], cb[4 * g + 2], cb[4 * g + 3]);
    }
    s.guard = (live && tot > 10.0f * NL_FILL_DEPTH) ? 1 : 0;
    const float steps = tot / a.step_size;
    s.step = (float)(1.0 / (double)steps);
    s.T = (int)ceilf(steps);
    // level 2 of the closing loop's context: the row-first ray's hit list (from the exchanged table if that ray lives on another rank)
    s.rf_cnt = 0; s.rf_bias = 0; s.rf_val[0] = s.rf_val[1] = s.rf_val[2] = -1;
    if (want_rf) {
        const int* src = a.hit_idx + (size_t)first_ray * NL_MAX_HITS;
        if (!is_local && a.row_first) {
            const int* e = a.row_first + (size_t)nl_row_first_entry(first_rank, Rg) * (1 + NL_MAX_HITS);
            src = e + 1; s.rf_cnt = e[0]; s.rf_bias = 1;
        } else {
            s.rf_cnt = a.hit_count[first_ray];
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) { const int l = j + SP_LPR * t; s.rf_val[t] = src[l < NL_MAX_HITS ? l : 0]; }
    }
    sp_wave_sync();
    return live;
}

// the walk of a ray whose set-up is in `s` / LDS; returns the ray's sample count on lane 0.  emit(sample index, voxel, depth, dist)
template <typename NoiseF, typename EmitF>
__device__ __forceinline__ int sp_walk(const SampleArgs& a, int j, const SpLds& m, const SpRay& s, NoiseF noise, EmitF emit)
{
    auto get_i = [&](int b) { return m.i[b]; };
    auto get_0 = [&](int b) { return m.t0[b]; };
    auto get_1 = [&](int b) { return m.t1[b]; };
    const int nb = s.nb, T = s.T;
    auto eval = [&](int cs, int* bin, float* z) {
        const float cdf = ((float)cs + noise(cs)) * s.step;
        unsigned gt = 0u;
#pragma unroll
        for (int q = 0; q < NL_MAX_HITS; ++q) gt |= (cdf > s.cumr[q]) ? (1u << q) : 0u;
        const int b = __builtin_ctz(~gt);                          // first interval with !(cdf > cum): at most nb (+inf beyond)
        *bin = b; *z = 0.0f;
        if (b < nb) {
            const float lo = (b > 0) ? m.c[b - 1] : 0.0f, hi = m.c[b];
            const float u = (cdf - lo) / (hi - lo);
            const float d0 = m.t0[b], d1 = m.t1[b];
            *z = d0 + u * (d1 - d0);
        }
    };
    int b_cur = 0; float z_cur = 0.0f;                             // this lane's latest evaluation
    int b_seg = 0; float z_seg = 0.0f;                             // lane 7's evaluation of the previous round (lane 0's predecessor)
    for (int base = 0; base < T; base += SP_LPR) {
        const int cs = base + j;
        if (cs < T) eval(cs, &b_cur, &z_cur);
        int bp = __shfl_up(b_cur, 1, SP_LPR); float zp = __shfl_up(z_cur, 1, SP_LPR);
        if (j == 0) { bp = b_seg; zp = z_seg; }
        b_seg = __shfl(b_cur, SP_LPR - 1, SP_LPR); z_seg = __shfl(z_cur, SP_LPR - 1, SP_LPR);
        if (cs < T) nl_walk_step_from(cs, bp, zp, b_cur, z_cur, nb, get_i, get_0, get_1, emit);
    }
    // the closing loop (lane 0), after the last step's evaluation; the row-first list goes to LDS first
    const int last = (T - 1) & (SP_LPR - 1);
    const int bin_e = __shfl(b_cur, last, SP_LPR); const float z_e = __shfl(z_cur, last, SP_LPR);
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int l = j + SP_LPR * t; if (l < NL_MAX_HITS) m.rf[l] = l < s.rf_cnt ? s.rf_val[t] - s.rf_bias : -1; }
    sp_wave_sync();
    int n = 0;
    if (j == 0) {
        NlTailCtx tc;
        tc.j_in_row = s.j_in_row; tc.rays_in_row = s.rays_in_row;
        tc.row_first_idx = m.rf; tc.row_first_count = NL_MAX_HITS; tc.row_first_bias = 0;
        tc.tail_always = a.tail_always != 0;
        n = nl_walk_tail_from(T, bin_e, z_e, nb, s.P, get_i, get_0, get_1, tc, eval, emit);
    }
    return n;
}

template <bool EMIT>
__global__ __launch_bounds__(NL_GEO_THREADS) void k_sample_par(SampleArgs a)
{
    __shared__ int s_red[8];
    __shared__ double s_dred[2];
    __shared__ __attribute__((aligned(16))) int s_i[SP_RAYS * NL_MAX_HITS], s_rf[SP_RAYS * NL_MAX_HITS];
    __shared__ __attribute__((aligned(16))) float s_0[SP_RAYS * NL_MAX_HITS], s_1[SP_RAYS * NL_MAX_HITS], s_c[SP_RAYS * NL_MAX_HITS], s_q[SP_RAYS * NL_MAX_HITS];
    if (threadIdx.x < 8) s_red[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_dred[threadIdx.x] = 0.0;
    __syncthreads();
    const int rl = threadIdx.x / SP_LPR, j = threadIdx.x % SP_LPR;
    const SpLds m = {s_i + rl * NL_MAX_HITS, s_0 + rl * NL_MAX_HITS, s_1 + rl * NL_MAX_HITS, s_c + rl * NL_MAX_HITS, s_q + rl * NL_MAX_HITS,
                     s_rf + rl * NL_MAX_HITS};
    // persistent over batches of SP_RAYS rays: the per-block reduction and its ~10 global atomics happen once per workgroup
    // (one workgroup per 32 rays would mean 4096 x 10 atomics on the same counters for a full scan)
    int vmax = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0;
    double d1 = 0.0, d2 = 0.0;
    const unsigned seed = a.seed_mix ? a.seed + 0x9E3779B9u * (*a.seed_mix) : a.seed;
    const bool hash = a.use_hash_noise != 0;
    for (int batch = blockIdx.x; batch * SP_RAYS < a.N; batch += gridDim.x) {
        const int r = batch * SP_RAYS + rl;
        int cnt = 0, nfs = 0, nsdf = 0, inv_fs = 0, inv_sdf = 0, guard = 0;
        double inv_d2 = 0.0;
        const float c = r < a.N ? a.cos_gt[r] : 0.0f, d = r < a.N ? a.gt_dist[r] : 0.0f;
        const int off = (EMIT && r < a.N) ? a.samp_off[r] : 0;
        SpRay s;
        const bool live = sp_setup(a, r, j, m, s);
        if (live) {
            guard = s.guard;
            const unsigned rid = (unsigned)(r + a.ray_id_base);
            auto noise = [&](int step) -> float { return hash ? nl_noise(seed, rid, (unsigned)step) : 0.5f; };
            const int cap = a.capacity;
            auto emit = [&](int s_, int vox, float depth, float dist) {
                if (EMIT) {
                    const int p = off + s_;
                    if (p < cap) { a.s_vox[p] = vox; a.s_depth[p] = depth; a.s_dist[p] = dist < 0.0f ? 0.0f : dist; a.s_ray[p] = r; }
                } else {
                    bool f, mk;
                    nl_loss_masks(depth * c, d, a.tau, a.max_depth, &f, &mk);
                    nfs += f ? 1 : 0; nsdf += mk ? 1 : 0;
                }
            };
            if (!guard) cnt = sp_walk(a, j, m, s, noise, emit);
            if (!EMIT && j == 0) {
                bool f, mk;
                nl_loss_masks(NL_FILL_DEPTH * c, d, a.tau, a.max_depth, &f, &mk);
                if (!guard) { inv_fs = f ? 1 : 0; inv_sdf = mk ? 1 : 0; inv_d2 = mk ? (double)d * (double)d : 0.0; }
            }
        }
        sp_wave_sync();                                                // the ray's LDS rows are reused by the next batch
        if (EMIT) continue;
        if (r < a.N && j == 0) a.samp_count[r] = cnt;
        // per-lane running sums (a ray's mask counts are spread over its lanes; cnt and the per-ray constants sit in lane 0)
        vmax = max(vmax, cnt);
        v1 += nfs; v2 += nsdf; v3 += inv_fs; v4 += inv_fs * cnt; v5 += inv_sdf; v6 += inv_sdf * cnt; v7 += guard;
        d1 += inv_d2; d2 += inv_d2 * (double)cnt;
    }
    if (EMIT) return;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        vmax = max(vmax, __shfl_xor(vmax, o2));
        v1 += __shfl_xor(v1, o2); v2 += __shfl_xor(v2, o2); v3 += __shfl_xor(v3, o2); v4 += __shfl_xor(v4, o2);
        v5 += __shfl_xor(v5, o2); v6 += __shfl_xor(v6, o2); v7 += __shfl_xor(v7, o2);
        d1 += __shfl_xor(d1, o2); d2 += __shfl_xor(d2, o2);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&s_red[0], vmax);
        atomicAdd(&s_red[1], v1); atomicAdd(&s_red[2], v2); atomicAdd(&s_red[3], v3); atomicAdd(&s_red[4], v4);
        atomicAdd(&s_red[5], v5); atomicAdd(&s_red[6], v6); atomicAdd(&s_red[7], v7);
        atomicAdd(&s_dred[0], d1); atomicAdd(&s_dred[1], d2);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_red[0] > 0) atomicMax(&a.counters[NLC_SMAX], s_red[0]);
        if (s_red[1]) atomicAdd(&a.counters[NLC_NFS], s_red[1]);
        if (s_red[2]) atomicAdd(&a.counters[NLC_NSDF], s_red[2]);
        if (s_red[3]) atomicAdd(&a.counters[NLC_INV_FS_RAYS], s_red[3]);
        if (s_red[4]) atomicAdd(&a.counters[NLC_INV_FS_CNT], s_red[4]);
        if (s_red[5]) atomicAdd(&a.counters[NLC_INV_SDF_RAYS], s_red[5]);
        if (s_red[6]) atomicAdd(&a.counters[NLC_INV_SDF_CNT], s_red[6]);
        if (s_red[7]) atomicMax(&a.counters[NLC_GUARD], 1);
        if (s_dred[0] != 0.0) atomicAdd(&a.dcounters[NLD_INV_D2], s_dred[0]);
        if (s_dred[1] != 0.0) atomicAdd(&a.dcounters[NLD_INV_D2CNT], s_dred[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// The sampler of the launch-bound regime (<= 8192 rays) as ONE launch instead of four (count pass, offset scan, loss normalisers,
// emit pass): every workgroup walks its 32 rays ONCE - the emit callback counts the loss masks and parks the samples in LDS -,
// obtains its global sample offset by decoupled look-back over the workgroups before it (status words tagged with the call's
// epoch: nothing to clear between iterations), writes the samples out, and the last workgroup to finish computes the loss
// normalisers.  Offsets are the exclusive prefix in ray order: the same samples at the same places as the four-launch sequence.
// A ray with more than SF_CAP samples makes its workgroup walk a second time, straight to memory (correct, just slower).
// ---------------------------------------------------------------------------------------------
#define SF_CAP 96
__device__ __forceinline__ void loss_finalize_one(int* __restrict__ counters, NlLossScalars* __restrict__ ls,
                                                  float fs_weight, float sdf_weight, float tau, float max_depth, int capacity);
#define SF_STATUS_AGG 1ull
#define SF_STATUS_PREFIX 2ull
struct SampleFusedArgs {
    SampleArgs s;
    int* samp_off_out;
    unsigned long long* wg_state;   // [0]: the call counter (epoch of the next launch - 1); [1 + b]: epoch << 34 | status << 32 | value
    NlLossScalars* ls; float fs_weight, sdf_weight;
};
__global__ __launch_bounds__(NL_GEO_THREADS) void k_sample_fused(SampleFusedArgs fa)
{
    const SampleArgs& a = fa.s;
    __shared__ int s_red[8];
    __shared__ double s_dred[2];
    __shared__ __attribute__((aligned(16))) int s_i[SP_RAYS * NL_MAX_HITS], s_rf[SP_RAYS * NL_MAX_HITS];
    __shared__ __attribute__((aligned(16))) float s_0[SP_RAYS * NL_MAX_HITS], s_1[SP_RAYS * NL_MAX_HITS], s_c[SP_RAYS * NL_MAX_HITS], s_q[SP_RAYS * NL_MAX_HITS];
    __shared__ int s_cnt[SP_RAYS], s_excl[SP_RAYS];
    __shared__ int b_vox[SP_RAYS * SF_CAP];
    __shared__ float b_depth[SP_RAYS * SF_CAP], b_dist[SP_RAYS * SF_CAP];
    __shared__ int s_base, s_ovf;
    // the launch's epoch lives in device memory and is advanced by the last workgroup to finish (after every workgroup has read
    // it): nothing to pass or clear from the host, and a captured launch can be replayed
    const unsigned epoch = ((unsigned)*reinterpret_cast<volatile unsigned long long*>(fa.wg_state) + 1u) & 0x3FFFFFFFu;
    unsigned long long* const wg_words = fa.wg_state + 1;
    if (threadIdx.x < 8) s_red[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_dred[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) s_ovf = 0;
    __syncthreads();
    const int rl = threadIdx.x / SP_LPR, j = threadIdx.x % SP_LPR;
    const int r = blockIdx.x * SP_RAYS + rl;
    int cnt = 0, nfs = 0, nsdf = 0, inv_fs = 0, inv_sdf = 0, guard = 0;
    double inv_d2 = 0.0;
    const SpLds m = {s_i + rl * NL_MAX_HITS, s_0 + rl * NL_MAX_HITS, s_1 + rl * NL_MAX_HITS, s_c + rl * NL_MAX_HITS, s_q + rl * NL_MAX_HITS,
                     s_rf + rl * NL_MAX_HITS};
    const float c = r < a.N ? a.cos_gt[r] : 0.0f, d = r < a.N ? a.gt_dist[r] : 0.0f;
    const unsigned seed = a.seed_mix ? a.seed + 0x9E3779B9u * (*a.seed_mix) : a.seed;
    SpRay sr;
    const bool live = sp_setup(a, r, j, m, sr);
    guard = live ? sr.guard : 0;
    const unsigned rid = (unsigned)(r + a.ray_id_base);
    const bool hash = a.use_hash_noise != 0;
    auto noise = [&](int step) -> float { return hash ? nl_noise(seed, rid, (unsigned)step) : 0.5f; };
    if (live && !guard) {
        auto emit = [&](int s_, int vox, float depth, float dist) {
            bool f, mk;
            nl_loss_masks(depth * c, d, a.tau, a.max_depth, &f, &mk);
            nfs += f ? 1 : 0; nsdf += mk ? 1 : 0;
            if (s_ < SF_CAP) { const int q = rl * SF_CAP + s_; b_vox[q] = vox; b_depth[q] = depth; b_dist[q] = dist < 0.0f ? 0.0f : dist; }
            else s_ovf = 1;
        };
        cnt = sp_walk(a, j, m, sr, noise, emit);
    }
    if (live && j == 0) {
        bool f, mk;
        nl_loss_masks(NL_FILL_DEPTH * c, d, a.tau, a.max_depth, &f, &mk);
        if (!guard) { inv_fs = f ? 1 : 0; inv_sdf = mk ? 1 : 0; inv_d2 = mk ? (double)d * (double)d : 0.0; }
    }
    if (j == 0) s_cnt[rl] = cnt;
    // the loss normalisers' sums (as in the count pass)
    {
        int vmax = cnt, v1 = nfs, v2 = nsdf, v3 = inv_fs, v4 = inv_fs * cnt, v5 = inv_sdf, v6 = inv_sdf * cnt, v7 = guard;
        double d1 = inv_d2, d2 = inv_d2 * (double)cnt;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
            vmax = max(vmax, __shfl_xor(vmax, o2));
            v1 += __shfl_xor(v1, o2); v2 += __shfl_xor(v2, o2); v3 += __shfl_xor(v3, o2); v4 += __shfl_xor(v4, o2);
            v5 += __shfl_xor(v5, o2); v6 += __shfl_xor(v6, o2); v7 += __shfl_xor(v7, o2);
            d1 += __shfl_xor(d1, o2); d2 += __shfl_xor(d2, o2);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&s_red[0], vmax);
            atomicAdd(&s_red[1], v1); atomicAdd(&s_red[2], v2); atomicAdd(&s_red[3], v3); atomicAdd(&s_red[4], v4);
            atomicAdd(&s_red[5], v5); atomicAdd(&s_red[6], v6); atomicAdd(&s_red[7], v7);
            atomicAdd(&s_dred[0], d1); atomicAdd(&s_dred[1], d2);
        }
    }
    __syncthreads();
    // the workgroup's sums go out first, from a wave that is not in the look-back (nothing waits for them here: their round trips run
    // under the look-back; the same thread draws the ticket at the end, behind its own fence)
    if (threadIdx.x == 64) {
        if (s_red[0] > 0) atomicMax(&a.counters[NLC_SMAX], s_red[0]);
        if (s_red[1]) atomicAdd(&a.counters[NLC_NFS], s_red[1]);
        if (s_red[2]) atomicAdd(&a.counters[NLC_NSDF], s_red[2]);
        if (s_red[3]) atomicAdd(&a.counters[NLC_INV_FS_RAYS], s_red[3]);
        if (s_red[4]) atomicAdd(&a.counters[NLC_INV_FS_CNT], s_red[4]);
        if (s_red[5]) atomicAdd(&a.counters[NLC_INV_SDF_RAYS], s_red[5]);
        if (s_red[6]) atomicAdd(&a.counters[NLC_INV_SDF_CNT], s_red[6]);
        if (s_red[7]) atomicMax(&a.counters[NLC_GUARD], 1);
        if (s_dred[0] != 0.0) atomicAdd(&a.dcounters[NLD_INV_D2], s_dred[0]);
        if (s_dred[1] != 0.0) atomicAdd(&a.dcounters[NLD_INV_D2CNT], s_dred[1]);
    }
    // this workgroup's sample offset: exclusive prefix over the rays inside it + look-back over the workgroups before it
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        int v = t < SP_RAYS ? s_cnt[t] : 0, inc = v;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) { const int u = __shfl_up(inc, o2); if (t >= o2) inc += u; }
        if (t < SP_RAYS) s_excl[t] = inc - v;
        const int agg = __shfl(inc, 63);
        const unsigned long long tag = (unsigned long long)epoch << 34;
        const int b = blockIdx.x;
        int base = 0;
        if (b > 0) {
            // look-back, 64 predecessors per step (one per lane): a lane spins until its word carries this call's epoch; the window is
            // summed up to and including the nearest published PREFIX, or entirely (all aggregates) and the next window follows
            if (t == 0) __hip_atomic_store(&wg_words[b], tag | (SF_STATUS_AGG << 32) | (unsigned)agg, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            for (int hi = b - 1; hi >= 0; hi -= 64) {
                const int i = hi - t;
                unsigned long long w = tag | (SF_STATUS_AGG << 32);                     // lanes before workgroup 0: empty aggregates
                if (i >= 0) {
                    do { w = __hip_atomic_load(&wg_words[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> 34) != epoch || ((w >> 32) & 3ull) == 0ull);
                }
                const unsigned long long pm = __ballot(((w >> 32) & 3ull) == SF_STATUS_PREFIX);
                const int first = pm ? __ffsll((long long)pm) - 1 : 64;                   // nearest predecessor with a prefix (lane index)
                int v = t <= first ? (int)(unsigned)(w & 0xFFFFFFFFull) : 0;
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2);
                base += v;
                if (pm) break;
            }
        }
        if (t == 0) {
            __hip_atomic_store(&wg_words[b], tag | (SF_STATUS_PREFIX << 32) | (unsigned)(base + agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_base = base;
            if (b == (int)gridDim.x - 1) a.counters[NLC_P] = base + agg;
        }
    }
    __syncthreads();
    const int off = s_base + s_excl[rl];
    if (r < a.N && j == 0) { a.samp_count[r] = cnt; fa.samp_off_out[r] = off; }
    const int cap = a.capacity;
    if (!s_ovf) {
        const int n = s_cnt[rl];
        for (int q = j; q < n; q += SP_LPR) {
            const int p = off + q, bq = rl * SF_CAP + q;
            if (p < cap) { a.s_vox[p] = b_vox[bq]; a.s_depth[p] = b_depth[bq]; a.s_dist[p] = b_dist[bq]; a.s_ray[p] = r; }
        }
    } else if (live && !guard) {                                  // a ray outgrew the LDS buffer: walk again, straight to memory
        auto emit = [&](int s_, int vox, float depth, float dist) {
            const int p = off + s_;
            if (p < cap) { a.s_vox[p] = vox; a.s_depth[p] = depth; a.s_dist[p] = dist < 0.0f ? 0.0f : dist; a.s_ray[p] = r; }
        };
        (void)sp_walk(a, j, m, sr, noise, emit);
    }
    // the ticket comes last, off the path of the workgroup's own samples: whoever draws the last one finds every workgroup's sums and
    // the total (stored before the barrier above, which drains the storing wave's memory operations) in place and computes the
    // loss normalisers
    if (threadIdx.x == 64) {
        __threadfence();
        if (atomicAdd(&a.counters[NLC_TICKET], 1) == (int)gridDim.x - 1) {
            __threadfence();
            *reinterpret_cast<volatile unsigned long long*>(fa.wg_state) = (unsigned long long)epoch;   // the next launch's epoch differs
            loss_finalize_one(a.counters, fa.ls, fa.fs_weight, fa.sdf_weight, a.tau, a.max_depth, a.capacity);
        }
    }
}

template <bool EMIT>
__global__ __launch_bounds__(NL_GEO_THREADS) void k_sample(SampleArgs a)
{
    __shared__ int s_red[8];
    __shared__ double s_dred[2];
    // the lane's hit list, entry-major ([entry][lane]: a lane's run-time bin index is conflict-free): 60 KB per workgroup
    __shared__ int s_hi[NL_MAX_HITS * NL_GEO_THREADS];
    __shared__ float s_h0[NL_MAX_HITS * NL_GEO_THREADS], s_h1[NL_MAX_HITS * NL_GEO_THREADS];
    if (threadIdx.x < 8) s_red[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_dred[threadIdx.x] = 0.0;
    __syncthreads();
    const int r = blockIdx.x * NL_GEO_THREADS + threadIdx.x;
    SSTAMP(0);
    int cnt = 0, nfs = 0, nsdf = 0, inv_fs = 0, inv_sdf = 0, guard = 0;
    double inv_d2 = 0.0;
    if (r < a.N && a.hit_count[r] > 0) {
        const int P = a.counters[NLC_HMAX];
        const int Rg = a.counters[NLC_R_GLOBAL];
        const int rank = a.hit_rank[r] + a.counters[NLC_R_OFFSET];
        float tot = 0.0f;
        const int nh = a.hit_count[r];
        int* my_i = s_hi + threadIdx.x; float* my_0 = s_h0 + threadIdx.x; float* my_1 = s_h1 + threadIdx.x;
#pragma unroll
        for (int l = 0; l < NL_MAX_HITS; ++l) {               // row tails beyond the ray's own hits are padding
            const bool v = l < nh;
            const int i_ = v ? a.hit_idx[(size_t)r * NL_MAX_HITS + l] : -1;
            const float t0_ = v ? a.hit_t0[(size_t)r * NL_MAX_HITS + l] : a.max_depth;
            const float t1_ = v ? a.hit_t1[(size_t)r * NL_MAX_HITS + l] : a.max_depth;
            my_i[l * NL_GEO_THREADS] = i_; my_0[l * NL_GEO_THREADS] = t0_; my_1[l * NL_GEO_THREADS] = t1_;
            if (l < P) tot = tot + ((i_ == -1) ? 0.0f : (t1_ - t0_));                             // same order as l = 0..P-1
        }
        auto get_i = [&](int b) { return my_i[b * NL_GEO_THREADS]; };
        auto get_0 = [&](int b) { return my_0[b * NL_GEO_THREADS]; };
        auto get_1 = [&](int b) { return my_1[b * NL_GEO_THREADS]; };
        SSTAMP(1);
        if (tot > 10.0f * NL_FILL_DEPTH) guard = 1;
        NlTailCtx tc;
        int first_rank;
        nl_sampler_layout(rank, Rg, &tc.j_in_row, &tc.rays_in_row, &first_rank);
        // single-GPU: the row's first ray is local.  multi-GPU: see k_sample_par above
        const int first_local = first_rank - a.counters[NLC_R_OFFSET];
        const bool is_local = first_local >= 0 && first_local < a.counters[NLC_R];
        const int first_ray = is_local ? a.ray_of_rank[first_local] : r;
        tc.row_first_idx = a.hit_idx + (size_t)first_ray * NL_MAX_HITS;
        tc.row_first_count = a.hit_count[first_ray];
        tc.row_first_bias = 0;
        if (!is_local && a.row_first) {
            const int* e = a.row_first + (size_t)nl_row_first_entry(first_rank, Rg) * (1 + NL_MAX_HITS);
            tc.row_first_idx = e + 1; tc.row_first_count = e[0]; tc.row_first_bias = 1;
        }
        tc.tail_always = a.tail_always != 0;
        SSTAMP(2);
        const float c = a.cos_gt[r], d = a.gt_dist[r];
        const unsigned rid = (unsigned)(r + a.ray_id_base);
        const unsigned seed = a.seed_mix ? a.seed + 0x9E3779B9u * (*a.seed_mix) : a.seed;
        const bool hash = a.use_hash_noise != 0;
        auto noise = [&](int step) -> float { return hash ? nl_noise(seed, rid, (unsigned)step) : 0.5f; };
        if (!guard) {
            if (EMIT) {
                const int off = a.samp_off[r];
                const int cap = a.capacity;
                auto emit = [&](int s, int vox, float depth, float dist) {
                    const int p = off + s;
                    if (p < cap) { a.s_vox[p] = vox; a.s_depth[p] = depth; a.s_dist[p] = dist < 0.0f ? 0.0f : dist; a.s_ray[p] = r; }
                };
                cnt = nl_sample_walk_core(get_i, get_0, get_1, P, tot, a.step_size, tc, noise, emit);
            } else {
                auto emit = [&](int s, int vox, float depth, float dist) {
                    (void)s; (void)vox; (void)dist;
                    bool f, m;
                    nl_loss_masks(depth * c, d, a.tau, a.max_depth, &f, &m);
                    nfs += f ? 1 : 0; nsdf += m ? 1 : 0;
                };
                cnt = nl_sample_walk_core(get_i, get_0, get_1, P, tot, a.step_size, tc, noise, emit);
                bool f, m;
                nl_loss_masks(NL_FILL_DEPTH * c, d, a.tau, a.max_depth, &f, &m);
                inv_fs = f ? 1 : 0; inv_sdf = m ? 1 : 0;
                inv_d2 = m ? (double)d * (double)d : 0.0;
            }
        }
    }
    SSTAMP(3);
    if (EMIT) return;
    if (r < a.N) a.samp_count[r] = cnt;
    // block reductions -> a handful of atomics per block
    int vmax = cnt;
    int v1 = nfs, v2 = nsdf, v3 = inv_fs, v4 = inv_fs * cnt, v5 = inv_sdf, v6 = inv_sdf * cnt, v7 = guard;
    double d1 = inv_d2, d2 = inv_d2 * (double)cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vmax = max(vmax, __shfl_xor(vmax, off));
        v1 += __shfl_xor(v1, off); v2 += __shfl_xor(v2, off); v3 += __shfl_xor(v3, off); v4 += __shfl_xor(v4, off);
        v5 += __shfl_xor(v5, off); v6 += __shfl_xor(v6, off); v7 += __shfl_xor(v7, off);
        d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&s_red[0], vmax);
        atomicAdd(&s_red[1], v1); atomicAdd(&s_red[2], v2); atomicAdd(&s_red[3], v3); atomicAdd(&s_red[4], v4);
        atomicAdd(&s_red[5], v5); atomicAdd(&s_red[6], v6); atomicAdd(&s_red[7], v7);
        atomicAdd(&s_dred[0], d1); atomicAdd(&s_dred[1], d2);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_red[0] > 0) atomicMax(&a.counters[NLC_SMAX], s_red[0]);
        if (s_red[1]) atomicAdd(&a.counters[NLC_NFS], s_red[1]);
        if (s_red[2]) atomicAdd(&a.counters[NLC_NSDF], s_red[2]);
        if (s_red[3]) atomicAdd(&a.counters[NLC_INV_FS_RAYS], s_red[3]);
        if (s_red[4]) atomicAdd(&a.counters[NLC_INV_FS_CNT], s_red[4]);
        if (s_red[5]) atomicAdd(&a.counters[NLC_INV_SDF_RAYS], s_red[5]);
        if (s_red[6]) atomicAdd(&a.counters[NLC_INV_SDF_CNT], s_red[6]);
        if (s_red[7]) atomicMax(&a.counters[NLC_GUARD], 1);
        if (s_dred[0] != 0.0) atomicAdd(&a.dcounters[NLD_INV_D2], s_dred[0]);
        if (s_dred[1] != 0.0) atomicAdd(&a.dcounters[NLD_INV_D2CNT], s_dred[1]);
    }
}

// one thread: global loss normalisers (criterion.py:84-88, :65 mean divisor) from the counters
__device__ __forceinline__ void loss_finalize_one(int* __restrict__ counters, NlLossScalars* __restrict__ ls,
                                                  float fs_weight, float sdf_weight, float tau, float max_depth, int capacity)
{
    const int R = counters[NLC_R_GLOBAL], S = counters[NLC_SMAX];
    const int n_fs = counters[NLC_NFS] + (S * counters[NLC_INV_FS_RAYS] - counters[NLC_INV_FS_CNT]);
    const int n_sdf = counters[NLC_NSDF] + (S * counters[NLC_INV_SDF_RAYS] - counters[NLC_INV_SDF_CNT]);
    const float nf = (float)n_fs, ns = (float)n_sdf;
    const float tot = ns + nf;
    NlLossScalars o;
    o.w_fs = 1.0f - nf / tot;
    o.w_sdf = 1.0f - ns / tot;
    const float n = (float)((long long)R * (long long)S);
    o.two_over_n = 2.0f / n;
    o.inv_n = 1.0f / n;
    o.fs_weight = fs_weight; o.sdf_weight = sdf_weight; o.tau = tau; o.max_depth = max_depth;
    o.R = R; o.S_max = S; o.P = counters[NLC_P]; o.ds_max_bits = 0u;
    if (counters[NLC_P] > capacity) { counters[NLC_OVERFLOW] = 1; o.P = capacity; }
    *ls = o;
}
__global__ void k_loss_finalize(int* __restrict__ counters, NlLossScalars* __restrict__ ls,
                                float fs_weight, float sdf_weight, float tau, float max_depth, int capacity)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_finalize_one(counters, ls, fs_weight, sdf_weight, tau, max_depth, capacity);
}
// ray-sharded iteration, exchange 2: the gathered counter blocks folded into the local one (summed loss normalisers / flags, max samples per
// ray, summed padded-slot constants - stage 2 of k_dist_merge, nl_optim.hip) and the loss scalars from the merged block, in ONE launch
__global__ void k_dist_merge_finalize(const int* __restrict__ gathered, int STRIDE, int world, int* __restrict__ counters, NlLossScalars* __restrict__ ls,
                                      float fs_weight, float sdf_weight, float tau, float max_depth, int capacity)
{
    const int t = threadIdx.x;
    if (t >= NLC_NFS && t <= NLC_GUARD) {                        // NFS, NSDF, INV_* (4), OVERFLOW, GUARD: contiguous
        int sum = 0;
        for (int r = 0; r < world; ++r) sum += gathered[r * STRIDE + t];
        counters[t] = sum;
    } else if (t == NLC_SMAX) {
        int m = 0;
        for (int r = 0; r < world; ++r) m = max(m, gathered[r * STRIDE + NLC_SMAX]);
        counters[NLC_SMAX] = m;
    } else if (t == 32 || t == 33) {
        const int d = t == 32 ? NLD_INV_D2 : NLD_INV_D2CNT;
        double sum = 0.0;
        for (int r = 0; r < world; ++r) sum += reinterpret_cast<const double*>(gathered + r * STRIDE + NL_CNT_INTS)[d];
        reinterpret_cast<double*>(counters + NL_CNT_INTS)[d] = sum;
    }
    __threadfence_block();
    __syncthreads();
    if (t == 0) loss_finalize_one(counters, ls, fs_weight, sdf_weight, tau, max_depth, capacity);
}
// the sample-offset scan of the launch-bound regime with the loss normalisers behind it (thread 0 wrote counters[NLC_P] itself)
__global__ __launch_bounds__(NL_GEO_THREADS) void k_scan_samples_finalize(const int* __restrict__ samp_count, int* __restrict__ samp_off, int n,
                                                                           int* __restrict__ counters, NlLossScalars* __restrict__ ls,
                                                                           float fs_weight, float sdf_weight, float tau, float max_depth, int capacity)
{
    scan_one_block(samp_count, samp_off, n, 0, nullptr, counters + NLC_P, nullptr);
    if (threadIdx.x == 0) loss_finalize_one(counters, ls, fs_weight, sdf_weight, tau, max_depth, capacity);
}

// ---------------------------------------------------------------------------------------------
// (b1) drop-in kernel behind grid.inverse_cdf_sampling: the reference kernel's semantics on the
// reference tensor layouts ([b, num_rays, *]); one ray per lane instead of one block per row.
// Outputs must be pre-filled (-1 / 0 / 0) by the caller like sample.cpp:78-87 does.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NL_GEO_THREADS) void k_inverse_cdf_raw(
    int b, int num_rays, int max_hits, int max_steps, float fixed_step_size,
    const int* __restrict__ pts_idx, const float* __restrict__ min_depth, const float* __restrict__ max_depth,
    const float* __restrict__ noise, const float* __restrict__ probs, const float* __restrict__ steps,
    int* __restrict__ sampled_idx, float* __restrict__ sampled_depth, float* __restrict__ sampled_dists)
{
    const long long gid = (long long)blockIdx.x * NL_GEO_THREADS + threadIdx.x;
    if (gid >= (long long)b * num_rays) return;
    const int bi = (int)(gid / num_rays), j = (int)(gid - (long long)bi * num_rays);
    const size_t rowH = (size_t)bi * num_rays * max_hits, rowK = (size_t)bi * num_rays * max_steps;
    const int* idx_row = pts_idx + rowH;
    const int H = j * max_hits;
    const size_t K = rowK + (size_t)j * max_steps;
    const int* idx = idx_row + H; const float* t0 = min_depth + rowH + H; const float* t1 = max_depth + rowH + H;
    const float* pr = probs + rowH + H; const float* nz = noise + K;
    int curr_bin = 0, s = 0;
    float curr_min_depth = t0[0], curr_max_depth = t1[0], curr_min_cdf = 0.0f, curr_max_cdf = pr[0];
    const float st = steps[(size_t)bi * num_rays + j];
    float step_size = (float)(1.0 / (double)st);
    float z_low = curr_min_depth;
    const int total_steps = (int)ceilf(st);
    bool done = false;
    if (fixed_step_size > 0.0f) step_size = fixed_step_size;
    for (int cs = 0; cs < total_steps; ++cs) {
        const float curr_cdf = ((float)cs + nz[cs]) * step_size;
        while (curr_cdf > curr_max_cdf) {
            sampled_idx[K + s] = idx[curr_bin];
            sampled_dists[K + s] = curr_max_depth - z_low;
            sampled_depth[K + s] = (curr_max_depth + z_low) * 0.5f;
            ++curr_bin; ++s;
            if (curr_bin >= max_hits || idx[curr_bin] == -1) { done = true; break; }
            curr_min_depth = t0[curr_bin]; curr_max_depth = t1[curr_bin];
            curr_min_cdf = curr_max_cdf; curr_max_cdf = curr_max_cdf + pr[curr_bin];
            z_low = curr_min_depth;
        }
        if (done) break;
        const float u = (curr_cdf - curr_min_cdf) / (curr_max_cdf - curr_min_cdf);
        const float z = curr_min_depth + u * (curr_max_depth - curr_min_depth);
        sampled_idx[K + s] = idx[curr_bin];
        sampled_dists[K + s] = z - z_low;
        sampled_depth[K + s] = (z + z_low) * 0.5f;
        z_low = z; ++s;
    }
    while (z_low < curr_max_depth && !done && num_rays > H + curr_bin) {
        sampled_idx[K + s] = idx[curr_bin];
        sampled_dists[K + s] = curr_max_depth - z_low;
        sampled_depth[K + s] = (curr_max_depth + z_low) * 0.5f;
        ++curr_bin; ++s;
        if (curr_bin >= max_hits || idx_row[curr_bin] == -1) break;      // the row's first ray
        curr_min_depth = t0[curr_bin]; curr_max_depth = t1[curr_bin];
        z_low = curr_min_depth;
    }
}

// =============================================================================================
// C ABI launchers (declared in include/nerfloam_hip.h)
// =============================================================================================
extern "C" {

int nl_svo_intersect(const float* ray_start, const float* ray_dir, const float* points, const int* children,
                     int b, int m, int n, float voxelsize, int n_max,
                     int* idx, float* min_depth, float* max_depth, void* stream)
{
    if (!ray_start || !ray_dir || !points || !children || !idx || !min_depth || !max_depth) return NL_ERR_INVALID_ARG;
    if (b <= 0 || m <= 0 || n <= 0 || n_max <= 0) return NL_ERR_INVALID_ARG;
    const long long total = (long long)b * m;
    // any n_max, like the reference (intersect.cpp:83-112; its only caller passes 20, voxel_helpers.py:533): up to 20 the hits are kept in
    // registers and written once, beyond the walk records into the output row
    if (n_max <= NL_MAX_HITS)
        hipLaunchKernelGGL(k_svo_intersect_raw<false>, dim3(nl_div_up(total, NL_GEO_THREADS)), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream,
                           b, n, m, voxelsize, n_max, ray_start, ray_dir, points, children, idx, min_depth, max_depth);
    else
        hipLaunchKernelGGL(k_svo_intersect_raw<true>, dim3(nl_div_up(total, NL_GEO_THREADS)), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream,
                           b, n, m, voxelsize, n_max, ray_start, ray_dir, points, children, idx, min_depth, max_depth);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_inverse_cdf_sampling(const int* pts_idx, const float* min_depth, const float* max_depth, const float* noise,
                            const float* probs, const float* steps, int b, int num_rays, int max_hits, int max_steps,
                            float fixed_step_size, int* sampled_idx, float* sampled_depth, float* sampled_dists, void* stream)
{
    if (!pts_idx || !min_depth || !max_depth || !noise || !probs || !steps || !sampled_idx || !sampled_depth || !sampled_dists)
        return NL_ERR_INVALID_ARG;
    if (b <= 0 || num_rays <= 0 || max_hits <= 0 || max_steps <= 0) return NL_ERR_INVALID_ARG;
    const long long total = (long long)b * num_rays;
    hipLaunchKernelGGL(k_inverse_cdf_raw, dim3(nl_div_up(total, NL_GEO_THREADS)), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream,
                       b, num_rays, max_hits, max_steps, fixed_step_size, pts_idx, min_depth, max_depth, noise, probs, steps,
                       sampled_idx, sampled_depth, sampled_dists);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

static int scan_launch(const int* in, int* out, int n, int flag_mode, int* ray_of_rank, int* total_out, int* total_out2, int* workspace,
                       void* stream, const ScanTail* tail = nullptr, bool* tail_done = nullptr);

/* lanes per ray of the work-list intersect by ray count and map size (nl_common.h: the launch-shape table); n_children_blocks 0 = unknown */
int nl_isect_lanes_for(int n_rays, int n_children_blocks)
{
    if (n_rays <= NL_RAYS_ISECT_32_LANES) return 32;
    if (n_rays <= NL_RAYS_ISECT_16_LANES) return n_children_blocks >= NL_BLOCKS_WIDE_MAP ? 32 : 16;
    return 8;
}

// (nl_dist.hip; the product header is not included by the kernel files)
int nl_dist_x1_pack(const int* counters, const int* hit_count, int N, int n_rays_cap, int* send, void* stream);

static int intersect_launch(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                            const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                            float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                            int* counters, int* scratch_rays, int* hit_rank, int* total_out, int* total_out2, int* scan_ws, int lanes, void* stream,
                            int* x1_send = nullptr, int x1_rays = 0)
{
    if (lanes != 0 && lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32) return NL_ERR_INVALID_ARG;
    if (N <= 0 || !rays_d_sensor || !points_gt || !cos_gt || !poses || !blk_ids || !blk_hdr || root_side < 2 || !rays_d_world || !gt_dist ||
        !hit_idx || !hit_t0 || !hit_t1 || !hit_count || !counters || !scratch_rays) return NL_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    // A/B override, else the caller's choice (the map's: see the header), else by ray count: up to 4096 rays the launch leaves most of the device
    // idle and 32 lanes cost nothing (one-scan map 2048 rays ~35 -> 30 us, 5 / 15 / 40 / 150 scans 39 -> 31, 61 -> 45, 94 -> 54, 122 -> 71); beyond, 32 lanes
    // mean twice the workgroups and only maps whose rays have wide fronts gain (profiles/r04_n_intersect_lanes_ab.txt)
    const int forced = g_isect_lpr.load(std::memory_order_relaxed);
    const int lpr = forced ? forced : lanes ? lanes : nl_isect_lanes_for(N, 0);
    auto kq = lpr == 32 ? k_ray_intersect_q<32> : lpr == 16 ? k_ray_intersect_q<16> : lpr == 8 ? k_ray_intersect_q<8> : k_ray_intersect_q<4>;
    const int IQ_RAYS = NL_GEO_THREADS / lpr;
    hipLaunchKernelGGL(kq, dim3(nl_div_up(N, IQ_RAYS)), dim3(NL_GEO_THREADS), 0, st,
                       N, rays_d_sensor, points_gt, cos_gt, frame_id, poses, (const int2*)blk_hdr, (const int4*)blk_ids, root_side, voxel_size, max_distance,
                       rays_d_world, gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, scratch_rays, g_isect_prune.load(std::memory_order_relaxed));
    // rays whose LDS queue / hit list overflowed (none on ordinary scans): sequential DFS, device-side count
    const DfsArgs da = {N, rays_d_sensor, points_gt, cos_gt, frame_id, poses, (const int2*)blk_hdr, (const int4*)blk_ids, root_side, voxel_size,
                        max_distance, rays_d_world, gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, (const int*)scratch_rays};
    if (hit_rank && N <= NL_SCAN_ONE_BLOCK_MAX) {        // launch-bound regime: fallback + hit-ray scan in one launch
        hipLaunchKernelGGL(k_scan_hits_fused, dim3(1), dim3(NL_GEO_THREADS), 0, st, da, hit_rank, scratch_rays, total_out, total_out2);
        NL_LAUNCH_CHECK();
        return x1_send ? nl_dist_x1_pack(counters, hit_count, N, x1_rays, x1_send, stream) : NL_OK;
    }
    hipLaunchKernelGGL(k_ray_intersect_dfs, dim3(32), dim3(NL_GEO_THREADS), 0, st, da);
    NL_LAUNCH_CHECK();
    if (!hit_rank) return NL_OK;
    // ray-sharded iteration: the send block of exchange 1 (a byte per ray + the counter block) is packed by the scan's own launch where that is
    // the single-launch kernel (a rank's share of a scan), by nl_dist_x1_pack behind it otherwise
    ScanTail tail = {};
    tail.counters = counters; tail.pack_send = x1_send; tail.pack_cap = x1_rays;
    bool packed = false;
    const int rc = scan_launch(hit_count, hit_rank, N, 1, scratch_rays, total_out, total_out2, scan_ws, stream, x1_send ? &tail : nullptr, &packed);
    if (rc != NL_OK || !x1_send || packed) return rc;
    return nl_dist_x1_pack(counters, hit_count, N, x1_rays, x1_send, stream);
}

int nl_ray_intersect(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                     const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                     float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                     int* counters, int* scratch_rays, void* stream)
{
    return intersect_launch(N, rays_d_sensor, points_gt, cos_gt, frame_id, poses, blk_hdr, blk_ids, root_side, voxel_size, max_distance, rays_d_world,
                            gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, scratch_rays, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

int nl_ray_intersect_lanes(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                           const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                           float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                           int* counters, int* scratch_rays, int lanes, void* stream)
{
    return intersect_launch(N, rays_d_sensor, points_gt, cos_gt, frame_id, poses, blk_hdr, blk_ids, root_side, voxel_size, max_distance, rays_d_world,
                            gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, scratch_rays, nullptr, nullptr, nullptr, nullptr, lanes, stream);
}

/* nl_ray_intersect + nl_scan_hit_rays (ray_of_rank == the intersect kernel's scratch list, as in the stage-wise sequence): one
 * launch fewer up to NL_SCAN_ONE_BLOCK_MAX rays, the same launches beyond */
int nl_ray_intersect_scan(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                          const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                          float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                          int* counters, int* ray_of_rank, int* hit_rank, int* total_out, int* total_out2, int* scan_ws, void* stream)
{
    if (!hit_rank || !total_out || !scan_ws) return NL_ERR_INVALID_ARG;
    return intersect_launch(N, rays_d_sensor, points_gt, cos_gt, frame_id, poses, blk_hdr, blk_ids, root_side, voxel_size, max_distance, rays_d_world,
                            gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, ray_of_rank, hit_rank, total_out, total_out2, scan_ws, 0, stream);
}

int nl_ray_intersect_scan_lanes(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                                const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                                float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                                int* counters, int* ray_of_rank, int* hit_rank, int* total_out, int* total_out2, int* scan_ws, int lanes, void* stream)
{
    if (!hit_rank || !total_out || !scan_ws) return NL_ERR_INVALID_ARG;
    return intersect_launch(N, rays_d_sensor, points_gt, cos_gt, frame_id, poses, blk_hdr, blk_ids, root_side, voxel_size, max_distance, rays_d_world,
                            gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, ray_of_rank, hit_rank, total_out, total_out2, scan_ws, lanes, stream);
}

/* nl_ray_intersect_scan_lanes + nl_dist_x1_pack (the send block of the ray-sharded iteration's first exchange: x1_send = [counter block |
 * x1_rays bytes], x1_rays >= N a multiple of 16): between 4097 and 32 768 rays the scan's launch packs the block itself */
int nl_ray_intersect_scan_x1(int N, const float* rays_d_sensor, const float* points_gt, const float* cos_gt, const int* frame_id,
                             const float* poses, const void* blk_hdr, const void* blk_ids, int root_side, float voxel_size, float max_distance,
                             float* rays_d_world, float* gt_dist, int* hit_idx, float* hit_t0, float* hit_t1, int* hit_count,
                             int* counters, int* ray_of_rank, int* hit_rank, int* total_out, int* total_out2, int* scan_ws, int lanes,
                             int* x1_send, int x1_rays, void* stream)
{
    if (!hit_rank || !total_out || !scan_ws || !x1_send || x1_rays < N || (x1_rays & 15)) return NL_ERR_INVALID_ARG;
    if (total_out != counters + NLC_R) return NL_ERR_INVALID_ARG;      // the packed counter block must already hold the hit-ray total
    return intersect_launch(N, rays_d_sensor, points_gt, cos_gt, frame_id, poses, blk_hdr, blk_ids, root_side, voxel_size, max_distance, rays_d_world,
                            gt_dist, hit_idx, hit_t0, hit_t1, hit_count, counters, ray_of_rank, hit_rank, total_out, total_out2, scan_ws, lanes, stream,
                            x1_send, x1_rays);
}

static std::atomic<int> g_scan_single{1};           // 0: mid-size scans as two launches (the pre-round-5 path; A/B and equality tests)

// `tail`: work a launch of its own would do behind the scan (ScanTail); *tail_done says whether this scan did it (only the single-launch
// kernel can: the caller issues the separate launch otherwise)
static int scan_launch(const int* in, int* out, int n, int flag_mode, int* ray_of_rank, int* total_out, int* total_out2,
                       int* workspace, void* stream, const ScanTail* tail, bool* tail_done)
{
    hipStream_t s = (hipStream_t)stream;
    if (tail_done) *tail_done = false;
    if (n <= NL_SCAN_ONE_BLOCK_MAX) {
        hipLaunchKernelGGL(k_scan_one_block, dim3(1), dim3(NL_GEO_THREADS), 0, s, in, out, n, flag_mode, ray_of_rank, total_out, total_out2);
    } else if (n <= NL_SCAN_SINGLE_MAX && g_scan_single.load(std::memory_order_relaxed) && !((uintptr_t)in & 15)) {
        ScanTail t = {};
        if (tail) { t = *tail; if (tail_done) *tail_done = true; }
        hipLaunchKernelGGL(k_scan_single, dim3(nl_div_up(n, NL_SCAN1_BLOCK)), dim3(NL_SCAN1_THREADS), 0, s, in, out, n, flag_mode, ray_of_rank, total_out,
                           total_out2, t);
    } else {
        const int nb = nl_div_up(n, NL_SCAN_BLOCK);
        hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(NL_GEO_THREADS), 0, s, in, out, workspace, n, flag_mode);
        hipLaunchKernelGGL(k_scan_finish, dim3(nb), dim3(NL_GEO_THREADS), 0, s, in, out, workspace, n, ray_of_rank, total_out, total_out2);
    }
    NL_LAUNCH_CHECK();
    return NL_OK;
}

// exclusive scan of in[0..n) into out, total -> *total_out (device).  workspace: >= ceil(n/1024) ints.
int nl_exclusive_scan_i32(const int* in, int* out, int n, int flag_mode, int* total_out, int* workspace, void* stream)
{
    if (n <= 0 || !in || !out || !total_out || !workspace) return NL_ERR_INVALID_ARG;
    return scan_launch(in, out, n, flag_mode, nullptr, total_out, nullptr, workspace, stream);
}

/* hit-ray bookkeeping in one go (render_helpers.py:219-227): hit_rank = exclusive scan of (hit_count > 0),
 * ray_of_rank[rank] = ray, number of hit rays -> *total_out and (optional) *total_out2 */
int nl_scan_hit_rays(const int* hit_count, int* hit_rank, int* ray_of_rank, int N, int* total_out, int* total_out2, int* workspace,
                     void* stream)
{
    if (N <= 0 || !hit_count || !hit_rank || !ray_of_rank || !total_out || !workspace) return NL_ERR_INVALID_ARG;
    return scan_launch(hit_count, hit_rank, N, 1, ray_of_rank, total_out, total_out2, workspace, stream);
}

/* fused sampler kernel: 0 = sequential walk, one lane per ray; 1 = step-parallel, 8 lanes per ray; 2 = step-parallel up to
 * 8192 rays, sequential beyond (default); same results */
int nl_geometry_set_sampler_mode(int mode) { if (mode < 0 || mode > 2) return NL_ERR_INVALID_ARG; g_sampler_mode = mode; return NL_OK; }

/* lanes per ray of the work-list intersect kernel: 0 = the caller's choice, else by ray count (32 up to 4096 rays, 16 up to 16 384, else 8), or 4 / 8 / 16 / 32 for every call */
int nl_geometry_set_intersect_prune(int on) { g_isect_prune = on ? 1 : 0; return NL_OK; }
/* A/B / test aid: 0 = scans of 4097 .. 32 768 items as two launches (k_scan_blocks + k_scan_finish, the pre-round-5 path), 1 = one launch (default) */
int nl_geometry_set_scan_single(int on) { g_scan_single = on ? 1 : 0; return NL_OK; }
int nl_geometry_set_lanes_per_ray(int lpr) { if (lpr != 0 && lpr != 4 && lpr != 8 && lpr != 16 && lpr != 32) return NL_ERR_INVALID_ARG; g_isect_lpr = lpr; return NL_OK; }

/* profiling aid: device buffer [blocks][8] int64: cycle stamps (start, after set-up, after traversal, after finalise) and the
 * round count of wave 0 of every workgroup of k_ray_intersect_q (NULL disables) */
int nl_geometry_set_debug_buffer(void* dbg)
{
    long long* p = (long long*)dbg;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_isect_dbg), &p, sizeof(p)) == hipSuccess ? NL_OK : NL_ERR_LAUNCH;
}

int nl_compact_hit_rays(int N, const int* hit_count, const int* hit_rank, int* ray_of_rank, void* stream)
{
    if (N <= 0 || !hit_count || !hit_rank || !ray_of_rank) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_compact_hit_rays, dim3(nl_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, hit_count, hit_rank, ray_of_rank);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_sample_rays(int emit, int N, const int* hit_idx, const float* hit_t0, const float* hit_t1, const int* hit_count,
                   const int* hit_rank, const int* ray_of_rank, const float* cos_gt, const float* gt_dist,
                   float step_size, float tau, float max_depth, unsigned seed, int use_hash_noise, int tail_always, int ray_id_base,
                   const unsigned* seed_mix, const int* row_first, int* counters, int* samp_count, const int* samp_off, int capacity,
                   int* s_vox, float* s_depth, float* s_dist, int* s_ray, void* stream)
{
    if (N <= 0 || !hit_idx || !hit_t0 || !hit_t1 || !hit_count || !hit_rank || !ray_of_rank || !cos_gt || !gt_dist || !counters || !samp_count)
        return NL_ERR_INVALID_ARG;
    if (emit && (!samp_off || !s_vox || !s_depth || !s_dist || !s_ray)) return NL_ERR_INVALID_ARG;
    SampleArgs a;
    a.N = N; a.hit_idx = hit_idx; a.hit_t0 = hit_t0; a.hit_t1 = hit_t1; a.hit_count = hit_count; a.hit_rank = hit_rank;
    a.ray_of_rank = ray_of_rank; a.cos_gt = cos_gt; a.gt_dist = gt_dist; a.step_size = step_size; a.tau = tau; a.max_depth = max_depth;
    a.seed = seed; a.use_hash_noise = use_hash_noise; a.tail_always = tail_always; a.ray_id_base = ray_id_base;
    a.seed_mix = seed_mix; a.row_first = row_first;
    a.counters = counters; a.dcounters = (double*)(counters + NL_CNT_INTS);
    a.samp_count = samp_count; a.samp_off = samp_off; a.capacity = capacity;
    a.s_vox = s_vox; a.s_depth = s_depth; a.s_dist = s_dist; a.s_ray = s_ray;
    // step-parallel (SP_LPR lanes per ray): the emit pass at every size (131 072 rays: 22.7 us against 29.3 us sequential, 16 384: 12.1
    // against 15.3), the count pass in the latency-bound regime only - beyond 8192 rays its 1024 workgroups x 10 same-line atomics and
    // the extra threads cost more than the shorter chains save (16 384 rays: 19.5 against 15.5 us, 131 072: 77 against 24)
    const int smode = g_sampler_mode.load(std::memory_order_relaxed);
    if (smode == 1 || (smode == 2 && (emit || N <= NL_RAYS_FUSED_SAMPLER))) {
        const int nbk = nl_div_up(N, SP_RAYS) < 1024 ? nl_div_up(N, SP_RAYS) : 1024;
        if (emit) hipLaunchKernelGGL(k_sample_par<true>, dim3(nbk), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream, a);
        else      hipLaunchKernelGGL(k_sample_par<false>, dim3(nbk), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream, a);
    } else {
        if (emit) hipLaunchKernelGGL(k_sample<true>, dim3(nl_div_up(N, NL_GEO_THREADS)), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream, a);
        else      hipLaunchKernelGGL(k_sample<false>, dim3(nl_div_up(N, NL_GEO_THREADS)), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream, a);
    }
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_scan_samples_finalize(const int* samp_count, int* samp_off, int N, int* counters, void* loss_scalars, float fs_weight, float sdf_weight,
                             float tau, float max_depth, int capacity, int* workspace, void* stream);

/* count pass + sample-offset scan + loss normalisers + emit pass as ONE launch (k_sample_fused) up to 8192 rays; beyond, the four
 * launches.  state: >= 8 * (1 + ceil(N / 32)) bytes of device memory that only this function touches, zero-initialised once
 * (word 0 counts the launches: the look-back words are tagged with it and never cleared). */
int nl_sample_rays_fused(int N, const int* hit_idx, const float* hit_t0, const float* hit_t1, const int* hit_count,
                         const int* hit_rank, const int* ray_of_rank, const float* cos_gt, const float* gt_dist,
                         float step_size, float tau, float max_depth, unsigned seed, int use_hash_noise, int tail_always, int ray_id_base,
                         const unsigned* seed_mix, int* counters, int* samp_count, int* samp_off, int capacity,
                         int* s_vox, float* s_depth, float* s_dist, int* s_ray, void* loss_scalars, float fs_weight, float sdf_weight,
                         void* state, int* scan_ws, void* stream)
{
    if (N <= 0 || !hit_idx || !hit_t0 || !hit_t1 || !hit_count || !hit_rank || !ray_of_rank || !cos_gt || !gt_dist || !counters || !samp_count ||
        !samp_off || !s_vox || !s_depth || !s_dist || !s_ray || !loss_scalars || !scan_ws) return NL_ERR_INVALID_ARG;
    if (N > NL_RAYS_FUSED_SAMPLER || !state) {                             // (16 384 rays measured: +0.03 ms against the four launches of the sequential sampler)
        int rc = nl_sample_rays(0, N, hit_idx, hit_t0, hit_t1, hit_count, hit_rank, ray_of_rank, cos_gt, gt_dist, step_size, tau, max_depth, seed,
                                use_hash_noise, tail_always, ray_id_base, seed_mix, nullptr, counters, samp_count, nullptr, capacity, nullptr, nullptr,
                                nullptr, nullptr, stream);
        if (rc != NL_OK) return rc;
        rc = nl_scan_samples_finalize(samp_count, samp_off, N, counters, loss_scalars, fs_weight, sdf_weight, tau, max_depth, capacity, scan_ws, stream);
        if (rc != NL_OK) return rc;
        return nl_sample_rays(1, N, hit_idx, hit_t0, hit_t1, hit_count, hit_rank, ray_of_rank, cos_gt, gt_dist, step_size, tau, max_depth, seed,
                              use_hash_noise, tail_always, ray_id_base, seed_mix, nullptr, counters, samp_count, samp_off, capacity, s_vox, s_depth,
                              s_dist, s_ray, stream);
    }
    SampleFusedArgs fa;
    SampleArgs& a = fa.s;
    a.N = N; a.hit_idx = hit_idx; a.hit_t0 = hit_t0; a.hit_t1 = hit_t1; a.hit_count = hit_count; a.hit_rank = hit_rank;
    a.ray_of_rank = ray_of_rank; a.cos_gt = cos_gt; a.gt_dist = gt_dist; a.step_size = step_size; a.tau = tau; a.max_depth = max_depth;
    a.seed = seed; a.use_hash_noise = use_hash_noise; a.tail_always = tail_always; a.ray_id_base = ray_id_base;
    a.seed_mix = seed_mix; a.row_first = nullptr;
    a.counters = counters; a.dcounters = (double*)(counters + NL_CNT_INTS);
    a.samp_count = samp_count; a.samp_off = samp_off; a.capacity = capacity;
    a.s_vox = s_vox; a.s_depth = s_depth; a.s_dist = s_dist; a.s_ray = s_ray;
    fa.samp_off_out = samp_off; fa.wg_state = (unsigned long long*)state;
    fa.ls = (NlLossScalars*)loss_scalars; fa.fs_weight = fs_weight; fa.sdf_weight = sdf_weight;
    hipLaunchKernelGGL(k_sample_fused, dim3(nl_div_up(N, SP_RAYS)), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream, fa);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_loss_finalize(int* counters, void* loss_scalars, float fs_weight, float sdf_weight, float tau, float max_depth,
                     int capacity, void* stream)
{
    if (!counters || !loss_scalars) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, counters, (NlLossScalars*)loss_scalars,
                       fs_weight, sdf_weight, tau, max_depth, capacity);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* exchange 2 of the ray-sharded iteration: nl_dist_merge_counters_strided(stage 2) + nl_loss_finalize in one launch */
int nl_dist_merge_finalize(const int* gathered, int stride_ints, int world, int* counters, void* loss_scalars, float fs_weight, float sdf_weight,
                           float tau, float max_depth, int capacity, void* stream)
{
    if (!gathered || !counters || !loss_scalars || world <= 0 || stride_ints < NL_CNT_INTS + 2 * NL_CNT_DOUBLES || (stride_ints & 1))
        return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_dist_merge_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, gathered, stride_ints, world, counters,
                       (NlLossScalars*)loss_scalars, fs_weight, sdf_weight, tau, max_depth, capacity);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* nl_exclusive_scan_i32(samp_count -> samp_off, total -> counters[NLC_P]) + nl_loss_finalize: one launch up to
 * NL_SCAN_ONE_BLOCK_MAX rays, the same launches beyond */
int nl_scan_samples_finalize(const int* samp_count, int* samp_off, int N, int* counters, void* loss_scalars, float fs_weight, float sdf_weight,
                             float tau, float max_depth, int capacity, int* workspace, void* stream)
{
    if (N <= 0 || !samp_count || !samp_off || !counters || !loss_scalars || !workspace) return NL_ERR_INVALID_ARG;
    if (N <= NL_SCAN_ONE_BLOCK_MAX) {
        hipLaunchKernelGGL(k_scan_samples_finalize, dim3(1), dim3(NL_GEO_THREADS), 0, (hipStream_t)stream, samp_count, samp_off, N, counters,
                           (NlLossScalars*)loss_scalars, fs_weight, sdf_weight, tau, max_depth, capacity);
        NL_LAUNCH_CHECK();
        return NL_OK;
    }
    ScanTail tail = {};
    tail.counters = counters; tail.ls = (NlLossScalars*)loss_scalars; tail.fs_weight = fs_weight; tail.sdf_weight = sdf_weight; tail.tau = tau;
    tail.max_depth = max_depth; tail.capacity = capacity; tail.finalize = 1;
    bool finalized = false;                              // (up to NL_SCAN_SINGLE_MAX rays the scan's last workgroup writes the loss scalars)
    const int rc = scan_launch(samp_count, samp_off, N, 0, nullptr, counters + NLC_P, nullptr, workspace, stream, &tail, &finalized);
    if (rc != NL_OK || finalized) return rc;
    return nl_loss_finalize(counters, loss_scalars, fs_weight, sdf_weight, tau, max_depth, capacity, stream);
}

}  // extern "C"
