// micro-benchmark: the decoder's 32x256x256 GEMM main loop in isolation (A from LDS, B = a 256 KB matrix streamed from L2
// with buffer loads), 1 or 2 workgroups of 4 waves per CU, to see which operand stream keeps the MFMA pipe waiting.
// Build: hipcc --offload-arch=gfx950 -O3 gemm_stream.hip -o gemm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define NW 256
#define LDH 257

__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }

template <int KP, bool LOAD_B, bool LOAD_A>
__device__ __forceinline__ void gemm(rsrc_t rsrc, int voff, const float* ap, f32x16& c0, f32x16& c1)
{
    constexpr int RB = 2 * NW * 4;
    float bA[2 * KP], bB[2 * KP], aA[KP], aB[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) { bA[2 * i] = LOAD_B ? bload(rsrc, voff, i * RB) : 1.f; bA[2 * i + 1] = LOAD_B ? bload(rsrc, voff + 128, i * RB) : 2.f; }
#pragma unroll
    for (int i = 0; i < KP; ++i) aA[i] = LOAD_A ? ap[2 * i] : 0.5f;
#pragma unroll 1
    for (int g = 0; g < NW / 2; g += 2 * KP) {
        const int so = g * RB;
#pragma unroll
        for (int i = 0; i < KP; ++i) { bB[2 * i] = LOAD_B ? bload(rsrc, voff, so + (KP + i) * RB) : 1.f; bB[2 * i + 1] = LOAD_B ? bload(rsrc, voff + 128, so + (KP + i) * RB) : 2.f; }
#pragma unroll
        for (int i = 0; i < KP; ++i) aB[i] = LOAD_A ? ap[2 * (g + KP + i)] : 0.25f;
#pragma unroll
        for (int i = 0; i < KP; ++i) { c0 = MFMA32(aA[i], bA[2 * i], c0); c1 = MFMA32(aA[i], bA[2 * i + 1], c1); }
        if (g + 2 * KP < NW / 2) {
#pragma unroll
            for (int i = 0; i < KP; ++i) { bA[2 * i] = LOAD_B ? bload(rsrc, voff, so + (2 * KP + i) * RB) : 1.f; bA[2 * i + 1] = LOAD_B ? bload(rsrc, voff + 128, so + (2 * KP + i) * RB) : 2.f; }
#pragma unroll
            for (int i = 0; i < KP; ++i) aA[i] = LOAD_A ? ap[2 * (g + 2 * KP + i)] : 0.5f;
        }
#pragma unroll
        for (int i = 0; i < KP; ++i) { c0 = MFMA32(aB[i], bB[2 * i], c0); c1 = MFMA32(aB[i], bB[2 * i + 1], c1); }
    }
}

template <int KP, bool LOAD_B, bool LOAD_A>
__global__ __launch_bounds__(256, 2) void k_stream(const float* W, float* out, long long* cyc, int reps)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 32 * LDH; i += 256) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, NW * NW * 4, 0x00020000);
    const int colA = 64 * w + l31;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        f32x16 c0, c1;
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        gemm<KP, LOAD_B, LOAD_A>(rs, (lh * NW + colA) * 4, lds + l31 * LDH + lh, c0, c1);
        for (int r = 0; r < 16; ++r) { acc0[r] += c0[r]; acc1[r] += c1[r]; }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KP, bool LOAD_B, bool LOAD_A>
static void run(int blocks, const char* tag, const float* W)
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks); (void)hipMalloc(&cyc, 8 * blocks);
    const int reps = 200;
    const size_t shm = 77 * 1024;
    (void)hipFuncSetAttribute((const void*)k_stream<KP, LOAD_B, LOAD_A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_stream<KP, LOAD_B, LOAD_A>), dim3(blocks), dim3(256), shm, 0, W, out, cyc, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_stream<KP, LOAD_B, LOAD_A>), dim3(blocks), dim3(256), shm, 0, W, out, cyc, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double mfma = 256.0 * reps;
    const double tf = mfma * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("%-44s WGs %3d (x%d per CU): %6.1f cycles/MFMA/wave, %.3f ms, %.1f TFLOP/s\n", tag, blocks, blocks / 256, (double)h[0] / mfma, ms, tf);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    float* W; (void)hipMalloc(&W, NW * NW * 4); (void)hipMemset(W, 0, NW * NW * 4);
    run<8, false, false>(256, "no loads", W);
    run<8, false, true>(256, "A from LDS only", W);
    run<8, true, false>(256, "B from L2 only, KP=8", W);
    run<8, true, true>(256, "A LDS + B L2, KP=8", W);
    run<4, true, true>(256, "A LDS + B L2, KP=4", W);
    run<8, false, false>(512, "no loads", W);
    run<8, false, true>(512, "A from LDS only", W);
    run<8, true, false>(512, "B from L2 only, KP=8", W);
    run<8, true, true>(512, "A LDS + B L2, KP=8", W);
    run<4, true, true>(512, "A LDS + B L2, KP=4", W);
    run<2, true, true>(512, "A LDS + B L2, KP=2", W);
    return 0;
}
