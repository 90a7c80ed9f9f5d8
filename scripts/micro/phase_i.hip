// micro-benchmark behind DESIGN.md 4.2 / VERDICT r03 item 4(i): phases H + I of k_decoder<train> (dH1 -> LDS, then dX = dH1 W1 on four waves and
// dW1 += dH1^T X on the other four, both as chains of v_mfma_f32_16x16x4_f32) against formulations that take dW1 straight from phase F's
// accumulator layout.
//
// Phase F leaves dH1 in the C layout of 32x32 tiles: lane (l31 = hidden unit inside the wave's 32 columns, lh), register r = sample row
// d32_row(r, lh).  As the A operand of v_mfma_f32_32x32x2_f32 that IS A[i = hidden][k = lh] of a product over SAMPLES: with B[k = lh][n = l31] =
// X[row(r, lh)][channel n] (n < 16, zero beyond) the 32 registers give dW1[hidden 32][channel 16] += dH1^T X with no LDS round trip - at twice
// the matrix-pipe time of the 16x16x4 form (half of the 32 output columns are unused).  dX = dH1 W1 reduces over the hidden units, which lie
// ACROSS lanes in that layout: it needs the transposition through LDS either way.
//
//   variant 0  the product's phases: 32 ds_write_b32 per lane (dH1 tile), barrier, waves 0-3 dX (64 MFMA16, two chains), waves 4-7 dW1 (64 MFMA16), barrier
//   variant 1  dW1 from registers on all eight waves (32 MFMA 32x32x2, B from the X tile in LDS) interleaved with the dH1 stores, barrier, waves 0-3 dX, barrier
//   variant 2  variant 1 with dX split over K: waves 0-3 the hidden units 0..127, waves 4-7 128..255 (32 MFMA16 each), partials through LDS, barrier, add
// All with one 512-thread workgroup per CU (two waves per SIMD), 256 workgroups.  Prints shader cycles per tile (thread 0 of workgroup 0).
// Build: hipcc --offload-arch=gfx950 -O3 phase_i.hip -o phase_i
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define LDH 257
#define LDX 17
#define D32_RR(r) (((r) & 3) + 8 * ((r) >> 2))
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void k_phase(float* out, long long* cyc, int iters)
{
    __shared__ float sD[64 * LDH];
    __shared__ float sX[64 * LDX];
    __shared__ float sW1[256 * 16];
    __shared__ float sPart[4 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const int col = 32 * w + l31;
    for (int i = tid; i < 64 * LDX; i += 512) sX[i] = (float)((i * 7) & 15) * 0.01f;
    for (int i = tid; i < 256 * 16; i += 512) sW1[i] = (float)((i * 5) & 31) * 0.01f;
    f32x16 g0v, g1v;
    for (int r = 0; r < 16; ++r) { g0v[r] = 0.001f * (float)(tid + r); g1v[r] = 0.002f * (float)(tid - r); }
    f32x4 accW1[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) accW1[t][r] = 0.f;
    f32x16 accW;
    for (int r = 0; r < 16; ++r) accW[r] = 0.f;
    float sink = 0.f;
    __syncthreads();
    long long t0 = 0;
    for (int it = -2; it < iters; ++it) {
        if (it == 0) t0 = __builtin_readcyclecounter();
        // (the values phase F would deliver: keep them changing so that nothing is hoisted)
        for (int r = 0; r < 16; ++r) { g0v[r] = g0v[r] * 0.999f + 0.001f; g1v[r] = g1v[r] * 0.998f + 0.002f; }
        if (VARIANT >= 1) {
            // dW1 from the accumulator layout, interleaved with the dH1 stores (the stores only feed dX now)
            const float msk = l31 < 16 ? 1.f : 0.f;
            const float* xb = sX + opaque(4 * lh * LDX + (l31 & 15));
            float* db = sD + opaque(4 * lh * LDH + col);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float b0 = xb[D32_RR(r) * LDX] * msk, b1 = xb[(32 + D32_RR(r)) * LDX] * msk;
                db[D32_RR(r) * LDH] = g0v[r]; db[(32 + D32_RR(r)) * LDH] = g1v[r];
                accW = MFMA32(g0v[r], b0, accW);
                accW = MFMA32(g1v[r], b1, accW);
            }
        } else {
            float* db = sD + opaque(4 * lh * LDH + col);
#pragma unroll
            for (int r = 0; r < 16; ++r) { db[D32_RR(r) * LDH] = g0v[r]; db[(32 + D32_RR(r)) * LDH] = g1v[r]; }
        }
        __syncthreads();
        if (VARIANT <= 1) {
            if (w < 4) {
                f32x4 cxa = {0.f, 0.f, 0.f, 0.f}, cxb = {0.f, 0.f, 0.f, 0.f};
                const float* ap = sD + opaque((16 * w + l15) * LDH + lq);
                const float* bq = sW1 + opaque(lq * 16 + l15);
                float aA[8], bA[8], aB[8], bB[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * i]; bA[i] = bq[4 * i * 16]; }
#pragma unroll 1
                for (int q = 0; q < 64; q += 16) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { aB[i] = ap[4 * (q + 8 + i)]; bB[i] = bq[4 * (q + 8 + i) * 16]; }
#pragma unroll
                    for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aA[i], bA[i], cxa); cxb = MFMA16(aA[i + 1], bA[i + 1], cxb); }
                    if (q + 16 < 64) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * (q + 16 + i)]; bA[i] = bq[4 * (q + 16 + i) * 16]; }
                    }
#pragma unroll
                    for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aB[i], bB[i], cxa); cxb = MFMA16(aB[i + 1], bB[i + 1], cxb); }
                }
                for (int r = 0; r < 4; ++r) sink += cxa[r] + cxb[r];
            } else if (VARIANT == 0) {
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
                for (int ii = 0; ii < 16; ii += 4) {
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
                    if (ii + 4 < 16) {
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
        } else {
            // dX split over K: every wave 16 rows x 128 hidden units (32 MFMA16, two chains)
            const int wr = w & 3, kh = w >> 2;
            f32x4 cxa = {0.f, 0.f, 0.f, 0.f}, cxb = {0.f, 0.f, 0.f, 0.f};
            const float* ap = sD + opaque((16 * wr + l15) * LDH + 128 * kh + lq);
            const float* bq = sW1 + opaque((128 * kh + lq) * 16 + l15);
            float aA[8], bA[8], aB[8], bB[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * i]; bA[i] = bq[4 * i * 16]; }
#pragma unroll 1
            for (int q = 0; q < 32; q += 16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { aB[i] = ap[4 * (q + 8 + i)]; bB[i] = bq[4 * (q + 8 + i) * 16]; }
#pragma unroll
                for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aA[i], bA[i], cxa); cxb = MFMA16(aA[i + 1], bA[i + 1], cxb); }
                if (q + 16 < 32) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { aA[i] = ap[4 * (q + 16 + i)]; bA[i] = bq[4 * (q + 16 + i) * 16]; }
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) { cxa = MFMA16(aB[i], bB[i], cxa); cxb = MFMA16(aB[i + 1], bB[i + 1], cxb); }
            }
            if (kh == 1) { f32x4 p = cxa + cxb; *reinterpret_cast<f32x4*>(sPart + (wr * 64 + lane) * 4) = p; }
            __syncthreads();
            if (kh == 0) { const f32x4 p = *reinterpret_cast<const f32x4*>(sPart + (wr * 64 + lane) * 4); for (int r = 0; r < 4; ++r) sink += cxa[r] + cxb[r] + p[r]; }
        }
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) sink += accW1[t][r];
    for (int r = 0; r < 16; ++r) sink += accW[r] + g0v[r] + g1v[r];
    out[blockIdx.x * 512 + tid] = sink;
    if (blockIdx.x == 0 && tid == 0) { cyc[0] = t0; cyc[1] = t1; }
}

template <int V>
static void run(const char* what)
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * 512); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    hipLaunchKernelGGL(k_phase<V>, dim3(256), dim3(512), 0, 0, out, cyc, 16);
    hipLaunchKernelGGL(k_phase<V>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("variant %d  %-88s %8.0f cycles / tile\n", V, what, (double)(h[1] - h[0]) / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("product: dH1 -> LDS | barrier | waves 0-3 dX, waves 4-7 dW1 (16x16x4 chains) | barrier");
    run<1>("dW1 from F's accumulators (32 x 32x32x2 per wave) under the dH1 stores | barrier | waves 0-3 dX | barrier");
    run<2>("variant 1 with dX split over K on all eight waves (+ partials through LDS, one more barrier)");
    return 0;
}
