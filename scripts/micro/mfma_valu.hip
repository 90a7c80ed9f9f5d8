// micro-benchmark: does VALU / LDS work of the SAME wave hide in the shadow of its MFMAs on gfx950 (one wave per SIMD)?
// The inner sequence is one asm block (nothing for the compiler to merge or move): 4 x { v_mfma_f32_32x32x16_bf16 on its own
// accumulator ; N independent v_fma_f32 }.  Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define V1 "v_fma_f32 %4, %4, %12, %13\n"
#define V2 V1 "v_fma_f32 %5, %5, %12, %13\n"
#define V3 V2 "v_fma_f32 %6, %6, %12, %13\n"
#define V4 V3 "v_fma_f32 %7, %7, %12, %13\n"
#define V5 V4 "v_fma_f32 %8, %8, %12, %13\n"
#define V6 V5 "v_fma_f32 %9, %9, %12, %13\n"
#define V7 V6 "v_fma_f32 %10, %10, %12, %13\n"
#define V8 V7 "v_fma_f32 %11, %11, %12, %13\n"
#define BODY(V)                                                                                                                   \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %14, %15, %0\n" V "v_mfma_f32_32x32x16_bf16 %1, %14, %15, %1\n" V                       \
                 "v_mfma_f32_32x32x16_bf16 %2, %14, %15, %2\n" V "v_mfma_f32_32x32x16_bf16 %3, %14, %15, %3\n" V                       \
                 : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                 : "v"(a), "v"(b), "v"(ha), "v"(hb))

template <int N, int THREADS>
__global__ __launch_bounds__(THREADS) void k_mv(float* out, long long* cyc, int iters)
{
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    const float a = 0.5f, b = threadIdx.x * 1e-3f;
    float v0 = b, v1 = b + 1, v2 = b + 2, v3 = b + 3, v4 = b + 4, v5 = b + 5, v6 = b + 6, v7 = b + 7;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (N == 0) BODY("");
        if (N == 1) BODY(V1);
        if (N == 2) BODY(V2);
        if (N == 4) BODY(V4);
        if (N == 6) BODY(V6);
        if (N == 8) BODY(V8);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int N, int THREADS = 256>
static void run()
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * THREADS); (void)hipMalloc(&cyc, 8 * 256);
    const int iters = 4096;
    hipLaunchKernelGGL((k_mv<N, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, 16);
    hipLaunchKernelGGL((k_mv<N, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long hh[16]; (void)hipMemcpy(hh, cyc, 128, hipMemcpyDeviceToHost);
    long long lo = hh[0], hi = hh[1];
    for (int w = 0; w < THREADS / 64; ++w) { lo = hh[2 * w] < lo ? hh[2 * w] : lo; hi = hh[2 * w + 1] > hi ? hh[2 * w + 1] : hi; }
    long long h[1] = {hi - lo};                          // first start to last end over the waves of workgroup 0
    printf("%d wave(s) per SIMD, 32x32x16 bf16 MFMA + %d independent v_fma_f32: %6.1f cycles per MFMA and SIMD\n", THREADS / 256, N,
           (double)h[0] / (4.0 * iters * (THREADS / 256)));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>(); run<1>(); run<2>(); run<4>(); run<6>(); run<8>();
    run<0, 512>(); run<2, 512>(); run<4, 512>(); run<6, 512>(); run<8, 512>();
    return 0;
}
