// micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 per SIMD on gfx950 for 1 / 2 waves per SIMD and 1 / 2 / 4
// independent accumulator chains (no memory traffic).  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int CHAINS>
__global__ void k_rate(float* out, long long* cyc, int iters)
{
    f32x16 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) c[i] = MFMA32(a, b, c[i]);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
static void run(int threads, int blocks, const char* tag)
{
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks); hipMalloc(&cyc, 8 * blocks);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    const double mfma_per_wave = 16.0 * iters;
    const double waves_per_simd = threads / 256.0 * (blocks / 256.0);
    const double tf = mfma_per_wave * 4096.0 * (threads / 64.0) * blocks / (ms * 1e-3) / 1e12;
    printf("%-28s threads %4d blocks %4d: %.1f cycles per MFMA per wave (x%.0f waves/SIMD), %.3f ms, %.1f TFLOP/s\n", tag, threads, blocks,
           (double)h[0] / mfma_per_wave, waves_per_simd, ms, tf);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1>(256, 256, "1 chain, 1 wave/SIMD");
    run<2>(256, 256, "2 chains, 1 wave/SIMD");
    run<4>(256, 256, "4 chains, 1 wave/SIMD");
    run<1>(512, 256, "1 chain, 2 waves/SIMD");
    run<2>(512, 256, "2 chains, 2 waves/SIMD");
    run<2>(256, 512, "2 chains, 2 WGs x 1 wave/SIMD");
    run<4>(512, 256, "4 chains, 2 waves/SIMD");
    return 0;
}
