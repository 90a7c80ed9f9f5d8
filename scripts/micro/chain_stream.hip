// micro-benchmark for the register-chained decoder design: ONE wave per SIMD (256-thread workgroups, one per
// CU), every wave owns 32 samples whose activations stay in registers as MFMA B operands (lane = sample), and streams the
// bf16 weight planes (the A operands, 1 KB fragments in MFMA order) straight from L2 into registers:
//   forward : 8 output tiles x 16 k-steps x (3 A planes x 3 B planes) = 1152 v_mfma_f32_32x32x16_bf16, 384 fragments
//   dgrad   : 8 output tiles x 16 k-steps x 3 A planes x S sample sub-tiles (S = 1 or 2: mask fragments are cheap), 384 fragments
// Question: does the weight stream (each of the 4 waves of a CU loads every fragment itself: L1 / L2 served) keep the matrix pipe
// fed at one wave per SIMD?  Ideal = 32 cycles per MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 chain_stream.hip -o chain_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define PLANE (256 * 256 * 2)

__device__ __forceinline__ uint4 bload4(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }

// RING: k-steps of A fragments in flight (register ring); LOAD: stream from memory or use constants; SUB: dgrad sample sub-tiles
template <int RING, bool LOAD, int SUB>
__global__ __launch_bounds__(256, 1) void k_chain(const unsigned short* Wf, const unsigned short* Wd, const uint4* Hinit, float* out,
                                                  long long* cyc, int tiles)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wf), 0, 3 * PLANE, 0x00020000);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wd), 0, 3 * PLANE, 0x00020000);
    const int voff = lane * 16;
    uint4 hb[16][3];                                     // this wave's H1 operand planes: 192 registers, live through the forward
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) hb[s][p] = Hinit[(s * 3 + p) * 64 + lane];
    f32x16 total;
    for (int r = 0; r < 16; ++r) total[r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int tile = 0; tile < tiles; ++tile) {
        // ---------------- forward: 8 x 16 k-steps, 9 MFMAs per k-step ----------------
        {
            uint4 aq[RING][3];
#pragma unroll
            for (int j = 0; j < RING - 1; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) aq[j][p] = LOAD ? bload4(rf, voff, p * PLANE + j * 1024) : make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll 1
            for (int nt = 0; nt < 8; ++nt) {
                f32x16 c;
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int jn = nt * 16 + s + RING - 1;            // linear k-step to prefetch
                    if (LOAD) {
                        const int so = (jn & 127) * 1024;
#pragma unroll
                        for (int p = 0; p < 3; ++p) aq[(s + RING - 1) % RING][p] = bload4(rf, voff, p * PLANE + so);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                        for (int pb = 0; pb < 3; ++pb)
                            c = MFMA(__builtin_bit_cast(bf16x8, aq[s % RING][pa]), __builtin_bit_cast(bf16x8, hb[s][pb]), c);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) total[r] += c[r] > 0.f ? c[r] : 0.f;
            }
        }
        // ---------------- dgrad: 8 x 16 k-steps, 3 MFMAs per k-step and sample sub-tile ----------------
        {
            uint4 aq[RING][3];
            uint4 mk[SUB];
#pragma unroll
            for (int u = 0; u < SUB; ++u) mk[u] = hb[u][0];
#pragma unroll
            for (int j = 0; j < RING - 1; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) aq[j][p] = LOAD ? bload4(rd, voff, p * PLANE + j * 1024) : make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll 1
            for (int kt = 0; kt < 8; ++kt) {
                f32x16 c[SUB];
#pragma unroll
                for (int u = 0; u < SUB; ++u) for (int r = 0; r < 16; ++r) c[u][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int jn = kt * 16 + s + RING - 1;
                    if (LOAD) {
                        const int so = (jn & 127) * 1024;
#pragma unroll
                        for (int p = 0; p < 3; ++p) aq[(s + RING - 1) % RING][p] = bload4(rd, voff, p * PLANE + so);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                        for (int u = 0; u < SUB; ++u)
                            c[u] = MFMA(__builtin_bit_cast(bf16x8, aq[s % RING][pa]), __builtin_bit_cast(bf16x8, mk[u]), c[u]);
                }
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) total[r] += c[u][r];
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float sres = 0.f;
    for (int r = 0; r < 16; ++r) sres += total[r];
    out[blockIdx.x * 256 + tid] = sres;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int RING, bool LOAD, int SUB>
static void run(const char* tag, const unsigned short* Wf, const unsigned short* Wd, const uint4* Hinit)
{
    const int blocks = 256, tiles = 40;
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks); (void)hipMalloc(&cyc, 8 * 4 * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<RING, LOAD, SUB>), dim3(blocks), dim3(256), 0, 0, Wf, Wd, Hinit, out, cyc, 2);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<RING, LOAD, SUB>), dim3(blocks), dim3(256), 0, 0, Wf, Wd, Hinit, out, cyc, tiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double mf = (1152.0 + 384.0 * SUB) * tiles;
    const double samples = 32.0 * (SUB == 1 ? 1.0 : 1.0) * tiles * 4 * blocks;   // forward samples per launch
    printf("%-40s ring %d: %6.1f cycles/MFMA (ideal 32), %8.0f cycles/wave-tile, %.3f ms, %.2f us per 1e3 fwd samples\n", tag, RING,
           (double)h[0] / mf, (double)h[0] / tiles, ms, ms * 1e3 / (samples / 1e3));
    (void)hipFree(out); (void)hipFree(cyc);
}


// LDS-shared weight stream: the four waves of the workgroup load each fragment ONCE (LDS-DMA, 12 fragments per wave and stage of
// 16 k-steps), synchronise once per stage, and read their A operands from LDS (ds_read_b128).  Two 48 KB stages.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int SUB>
__global__ __launch_bounds__(256, 1) void k_chain_lds(const unsigned short* Wf, const unsigned short* Wd, const uint4* Hinit, float* out,
                                                      long long* cyc, int tiles)
{
    __shared__ __attribute__((aligned(16))) uint4 ring[2][48][64];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint4 hb[16][3];
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) hb[s][p] = Hinit[(s * 3 + p) * 64 + lane];
    f32x16 total;
    for (int r = 0; r < 16; ++r) total[r] = 0.f;
    // stage j of a tile: j < 8 forward tile j (planes Wf), else dgrad tile j - 8 (planes Wd); fragment f = 3 s + p
    auto issue = [&](int j, int buf) {
        const unsigned short* base = (j & 8) ? Wd : Wf;
        const int t = j & 7;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int f = w * 12 + i, s = f / 3, p = f % 3;
            const char* src = reinterpret_cast<const char*>(base) + (size_t)p * PLANE + (size_t)(t * 16 + s) * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)&ring[buf][f][0], 16, 0, 0);
        }
    };
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0)
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
        for (int j = 0; j < 16; ++j) {
            issue((j + 1) & 15, (j + 1) & 1);
            const uint4 (*rb)[64] = ring[j & 1];
            // A fragments of k-step s + 1 are read from LDS while the MFMAs of k-step s run (one wave per SIMD: nobody else hides it)
            uint4 af[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) af[0][p] = rb[p][lane];
            if (j < 8) {
                f32x16 c;
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s + 1 < 16) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) af[(s + 1) & 1][p] = rb[3 * (s + 1) + p][lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                        for (int pb = 0; pb < 3; ++pb) c = MFMA(__builtin_bit_cast(bf16x8, af[s & 1][pa]), __builtin_bit_cast(bf16x8, hb[s][pb]), c);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) total[r] += c[r] > 0.f ? c[r] : 0.f;
            } else {
                f32x16 c[SUB];
#pragma unroll
                for (int u = 0; u < SUB; ++u) for (int r = 0; r < 16; ++r) c[u][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s + 1 < 16) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) af[(s + 1) & 1][p] = rb[3 * (s + 1) + p][lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int u = 0; u < SUB; ++u) c[u] = MFMA(__builtin_bit_cast(bf16x8, af[s & 1][p]), __builtin_bit_cast(bf16x8, hb[u][0]), c[u]);
                }
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) total[r] += c[u][r];
            }
            __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0): this wave's DMA pieces of the next stage have landed
            __syncthreads();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float sres = 0.f;
    for (int r = 0; r < 16; ++r) sres += total[r];
    out[blockIdx.x * 256 + tid] = sres;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

// Register-staged LDS sharing: every wave loads a quarter of the NEXT stage's 48 fragments into registers (4 at a time, issued at
// k-steps 0 / 4 / 8 and written to LDS three k-steps later), one barrier per stage.  DG_LDS: the dgrad stages go through LDS too;
// otherwise they stream from L2 per wave (register ring of 4 k-steps) and only the forward planes are shared.
// Cycle counters: [0] whole loop, [1] forward stages, [2] dgrad stages.
template <int SUB, bool DG_LDS, int FILL, int SCHED = 0, int CHAINS = 1>      // CHAINS: forward accumulator chains; FILL: 0 none, 1 load + store, 2 loads only (folded into a register), 3 LDS stores only; SCHED 0: fills loaded at k-steps 0/4/8, stored 3 later; 1: loaded 0/5/10, stored 5 later
__global__ __launch_bounds__(256, 1) void k_chain_rs(const unsigned short* Wf, const unsigned short* Wd, const uint4* Hinit, float* out,
                                                     long long* cyc, int tiles)
{
    __shared__ __attribute__((aligned(16))) uint4 ring[2][48][64];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wf), 0, 3 * PLANE, 0x00020000);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wd), 0, 3 * PLANE, 0x00020000);
    const rsrc_t rfd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wf), 0, 6 * PLANE, 0x00020000);   // Wd == Wf + 3 planes
    const int voff = lane * 16;
    uint4 hb[16][3];
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) hb[s][p] = Hinit[(s * 3 + p) * 64 + lane];
    f32x16 total;
    for (int r = 0; r < 16; ++r) total[r] = 0.f;
    constexpr int NST = DG_LDS ? 16 : 8;                 // LDS stages per tile
    // this wave's share of stage j: fragments f = 12 w + i, k-step f / 3, plane f % 3
    uint4 st[5];
    auto fill_load = [&](int j, int b) {                 // b: batch 0..2 of four fragments
        // one resource over both plane sets (they are adjacent, as in the decoder workspace): a run-time choice between two
        // resources becomes a waterfall loop around every load
        const int t = j & 7, dg = (j & 8) ? 3 * PLANE : 0;
        if (FILL == 3) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = w * 12 + b * 4 + i;
            st[i] = bload4(rfd, voff, dg + (f % 3) * PLANE + (t * 16 + f / 3) * 1024);
        }
    };
    unsigned sink = 0;
    auto fill_load1 = [&](int j, int i) {                // SCHED 2: fragment i of this wave's twelve
        const int t = j & 7, dg = (j & 8) ? 3 * PLANE : 0;
        const int f = w * 12 + i;
        if (FILL != 3) st[i % 5] = bload4(rfd, voff, dg + (f % 3) * PLANE + (t * 16 + f / 3) * 1024);
    };
    auto fill_store1 = [&](int buf, int i) {
        if (FILL == 2) { sink ^= st[i % 5].x ^ st[i % 5].y ^ st[i % 5].z ^ st[i % 5].w; return; }
        ring[buf][w * 12 + i][lane] = (FILL == 3) ? make_uint4(0, 0, 0, 0) : st[i % 5];
    };
    auto fill_store = [&](int buf, int b) {
        if (FILL == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sink ^= st[i].x ^ st[i].y ^ st[i].z ^ st[i].w;
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[buf][w * 12 + b * 4 + i][lane] = (FILL == 3) ? make_uint4(0, 0, 0, 0) : st[i];
    };
    if (FILL) { for (int b = 0; b < 3; ++b) { fill_load(0, b); fill_store(0, b); } }
    __syncthreads();
    long long cf = 0, cd = 0;
    const long long t0 = __builtin_readcyclecounter();
    int lbuf = 0;                                        // LDS buffer holding the current LDS stage
    for (int tile = 0; tile < tiles; ++tile) {
        long long ta = __builtin_readcyclecounter();
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {                    // ---------------- forward stages ----------------
            const int jn = (j + 1) % NST;                // next LDS stage (forward 0 again after the last one when the dgrad streams from L2)
            const bool fill = FILL && (DG_LDS || j < 7);
            const uint4 (*rb)[64] = ring[lbuf];
            uint4 af[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) af[0][p] = rb[p][lane];
            __builtin_amdgcn_sched_barrier(0);           // keep the preamble reads out of the pipelined region (the group barriers would count them)
            f32x16 c, c2;
            for (int r = 0; r < 16; ++r) { c[r] = 0.f; c2[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) af[(s + 1) & 1][p] = rb[3 * (s + 1) + p][lane];
                }
                if (SCHED == 0) {
                    if (fill && (s == 0 || s == 4 || s == 8)) fill_load(jn, s / 4);
                    if (fill && (s == 3 || s == 7 || s == 11)) fill_store(lbuf ^ 1, s / 4);
                } else if (SCHED == 1) {
                    if (fill && (s == 5 || s == 10 || s == 15)) fill_store(lbuf ^ 1, s / 5 - 1);
                    if (fill && (s == 0 || s == 5 || s == 10)) fill_load(jn, s / 5);
                } else {
                    if (fill && s >= 4) fill_store1(lbuf ^ 1, s - 4);
                    if (fill && s < 12) fill_load1(jn, s);
                }
                if (SCHED != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                    for (int pb = 0; pb < 3; ++pb) {
                        if (CHAINS == 2 && ((3 * pa + pb) & 1)) c2 = MFMA(__builtin_bit_cast(bf16x8, af[s & 1][pa]), __builtin_bit_cast(bf16x8, hb[s][pb]), c2);
                        else c = MFMA(__builtin_bit_cast(bf16x8, af[s & 1][pa]), __builtin_bit_cast(bf16x8, hb[s][pb]), c);
                    }
                if (SCHED == 2) {                        // one memory instruction in the shadow of each MFMA
                    for (int q = 0; q < 3; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = CHAINS == 2 ? c[r] + c2[r] : c[r]; total[r] += v > 0.f ? v : 0.f; }
            if (DG_LDS || j < 7) { __syncthreads(); lbuf ^= 1; }
        }
        long long tb = __builtin_readcyclecounter();
        cf += tb - ta;
        if (DG_LDS) {
#pragma unroll 1
            for (int j = 8; j < 16; ++j) {               // ---------------- dgrad stages through LDS ----------------
                const int jn = (j + 1) % 16;
                const uint4 (*rb)[64] = ring[lbuf];
                uint4 af[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[0][p] = rb[p][lane];
                __builtin_amdgcn_sched_barrier(0);
                f32x16 c[SUB];
#pragma unroll
                for (int u = 0; u < SUB; ++u) for (int r = 0; r < 16; ++r) c[u][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s + 1 < 16) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) af[(s + 1) & 1][p] = rb[3 * (s + 1) + p][lane];
                    }
                    if (SCHED == 0) {
                        if (FILL && (s == 0 || s == 4 || s == 8)) fill_load(jn, s / 4);
                        if (FILL && (s == 3 || s == 7 || s == 11)) fill_store(lbuf ^ 1, s / 4);
                    } else if (SCHED == 1) {
                        if (FILL && (s == 5 || s == 10 || s == 15)) fill_store(lbuf ^ 1, s / 5 - 1);
                        if (FILL && (s == 0 || s == 5 || s == 10)) fill_load(jn, s / 5);
                    } else {
                        if (FILL && s >= 4) fill_store1(lbuf ^ 1, s - 4);
                        if (FILL && s < 12) fill_load1(jn, s);
                    }
                    if (SCHED != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int u = 0; u < SUB; ++u) c[u] = MFMA(__builtin_bit_cast(bf16x8, af[s & 1][p]), __builtin_bit_cast(bf16x8, hb[u][0]), c[u]);
                    if (SCHED == 2) {
                        for (int q = 0; q < 3; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                        if (SUB == 2) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        } else {
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) total[r] += c[u][r];
                __syncthreads(); lbuf ^= 1;
            }
        } else {                                         // ---------------- dgrad from L2, forward stage 0 of the next tile filled meanwhile ----------------
            constexpr int RING = 4;
            uint4 aq[RING][3];
#pragma unroll
            for (int j = 0; j < RING - 1; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) aq[j][p] = bload4(rd, voff, p * PLANE + j * 1024);
#pragma unroll 1
            for (int kt = 0; kt < 8; ++kt) {
                f32x16 c[SUB];
#pragma unroll
                for (int u = 0; u < SUB; ++u) for (int r = 0; r < 16; ++r) c[u][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int so = ((kt * 16 + s + RING - 1) & 127) * 1024;
#pragma unroll
                    for (int p = 0; p < 3; ++p) aq[(s + RING - 1) % RING][p] = bload4(rd, voff, p * PLANE + so);
                    if (FILL && kt == 7 && (s == 0 || s == 4 || s == 8)) fill_load(0, s / 4);
                    if (FILL && kt == 7 && (s == 3 || s == 7 || s == 11)) fill_store(lbuf ^ 1, s / 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                        for (int u = 0; u < SUB; ++u)
                            c[u] = MFMA(__builtin_bit_cast(bf16x8, aq[s % RING][pa]), __builtin_bit_cast(bf16x8, hb[u][0]), c[u]);
                }
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) total[r] += c[u][r];
            }
            __syncthreads(); lbuf ^= 1;
        }
        cd += __builtin_readcyclecounter() - tb;
    }
    const long long t1 = __builtin_readcyclecounter();
    float sres = (float)sink;
    for (int r = 0; r < 16; ++r) sres += total[r];
    out[blockIdx.x * 256 + tid] = sres;
    if (tid == 0) { cyc[blockIdx.x * 4 + 0] = t1 - t0; cyc[blockIdx.x * 4 + 1] = cf; cyc[blockIdx.x * 4 + 2] = cd; }
}

template <int SUB, bool DG_LDS, int FILL, int SCHED = 0, int CHAINS = 1>
static void run_rs(const char* tag, const unsigned short* Wf, const unsigned short* Wd, const uint4* Hinit)
{
    const int blocks = 256, tiles = 40;
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks); (void)hipMalloc(&cyc, 8 * 4 * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain_rs<SUB, DG_LDS, FILL, SCHED, CHAINS>), dim3(blocks), dim3(256), 0, 0, Wf, Wd, Hinit, out, cyc, 2);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain_rs<SUB, DG_LDS, FILL, SCHED, CHAINS>), dim3(blocks), dim3(256), 0, 0, Wf, Wd, Hinit, out, cyc, tiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-52s: fwd %5.1f cycles/MFMA, dgrad %5.1f cycles/MFMA, %8.0f cycles/wave-tile, %.3f ms  [err %s]\n", tag,
           (double)h[1] / (1152.0 * tiles), (double)h[2] / (384.0 * SUB * tiles), (double)h[0] / tiles, ms, hipGetErrorString(hipGetLastError()));
    (void)hipFree(out); (void)hipFree(cyc);
}

template <int SUB>
static void run_lds(const char* tag, const unsigned short* Wf, const unsigned short* Wd, const uint4* Hinit)
{
    const int blocks = 256, tiles = 40;
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks); (void)hipMalloc(&cyc, 8 * 4 * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain_lds<SUB>), dim3(blocks), dim3(256), 0, 0, Wf, Wd, Hinit, out, cyc, 2);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain_lds<SUB>), dim3(blocks), dim3(256), 0, 0, Wf, Wd, Hinit, out, cyc, tiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double mf = (1152.0 + 384.0 * SUB) * tiles;
    printf("%-40s        : %6.1f cycles/MFMA (ideal 32), %8.0f cycles/wave-tile, %.3f ms  [err %s]\n", tag, (double)h[0] / mf, (double)h[0] / tiles, ms,
           hipGetErrorString(hipGetLastError()));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    unsigned short *Wf, *Wd; uint4* H;
    (void)hipMalloc(&Wf, 6 * PLANE); Wd = Wf + 3 * PLANE / 2; (void)hipMalloc(&H, 48 * 64 * 16);
    (void)hipMemset(Wf, 0, 6 * PLANE); (void)hipMemset(H, 0, 48 * 64 * 16);
    run<2, false, 1>("no loads, dgrad 32 samples", Wf, Wd, H);
    run<4, true, 1>("weights from L2, dgrad 32 samples", Wf, Wd, H);
    run<4, true, 2>("weights from L2, dgrad 64 samples", Wf, Wd, H);
    run_lds<1>("weights shared through LDS, dgrad 32", Wf, Wd, H);
    run_lds<2>("weights shared through LDS, dgrad 64", Wf, Wd, H);
    run_rs<1, true, 0>("LDS reads + barriers only (no fill), dgrad 32", Wf, Wd, H);
    run_rs<2, true, 0>("LDS reads + barriers only (no fill), dgrad 64", Wf, Wd, H);
    run_rs<1, true, 1>("register-staged LDS sharing, fwd + dgrad 32", Wf, Wd, H);
    run_rs<2, true, 1>("register-staged LDS sharing, fwd + dgrad 64", Wf, Wd, H);
    run_rs<1, false, 1>("register-staged LDS fwd, dgrad 32 from L2", Wf, Wd, H);
    run_rs<2, false, 1>("register-staged LDS fwd, dgrad 64 from L2", Wf, Wd, H);
    run_rs<1, true, 1, 1>("register-staged LDS sharing, fwd + dgrad 32, late store", Wf, Wd, H);
    run_rs<2, true, 1, 1>("register-staged LDS sharing, fwd + dgrad 64, late store", Wf, Wd, H);
    run_rs<2, false, 1, 1>("register-staged LDS fwd, dgrad 64 from L2, late store", Wf, Wd, H);
    run_rs<1, true, 1, 2>("register-staged, one memory op per MFMA, dgrad 32", Wf, Wd, H);
    run_rs<2, true, 1, 2>("register-staged, one memory op per MFMA, dgrad 64", Wf, Wd, H);
    run_rs<2, true, 1, 2, 2>("register-staged, one op per MFMA, two fwd chains", Wf, Wd, H);
    run_rs<2, true, 0, 2>("no fill, one memory op per MFMA, dgrad 64", Wf, Wd, H);
    run_rs<2, true, 2, 2>("fill loads only, one memory op per MFMA, dgrad 64", Wf, Wd, H);
    run_rs<2, true, 3, 2>("fill stores only, one memory op per MFMA, dgrad 64", Wf, Wd, H);
    run_rs<2, true, 2, 1>("fill loads only (no LDS store), late", Wf, Wd, H);
    run_rs<2, true, 3, 1>("fill LDS stores only (no loads), late", Wf, Wd, H);
    return 0;
}
