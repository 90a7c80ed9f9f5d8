// micro-benchmark: cycles per MFMA on gfx950, one wave per SIMD, for the shapes the decoder uses and for the small-K fp32 / bf16 shapes the
// K = 16 layers could use instead (4 independent accumulator chains, no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256, 1) void k_shape(float* out, long long* cyc, int iters)
{
    f32x16 c32[4]; f32x4 c16[4];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) c32[i][r] = 0.f; for (int r = 0; r < 4; ++r) c16[i][r] = 0.f; }
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
    float vv[8];
    for (int v = 0; v < 8; ++v) vv[v] = a + v;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) c32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c32[i], 0, 0, 0);
                if (KIND == 1) c16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c16[i], 0, 0, 0);
                if (KIND == 2) c32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c32[i], 0, 0, 0);
                if (KIND == 3) c16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c16[i], 0, 0, 0);
                if (KIND == 4) {            // the dgrad stage's mix: one 32x32x16 bf16 + one 16x16x4 f32 per slot
                    c32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c32[i], 0, 0, 0);
                    c16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c16[i], 0, 0, 0);
                }
                if (KIND >= 6 && KIND <= 9) {  // one 32x32x16 bf16 MFMA + (KIND - 5) * 2 independent VALU ops (does VALU work hide in the MFMA's shadow?)
                    c32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c32[i], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < (KIND - 5) * 2; ++v) vv[v] = __builtin_fmaf(vv[v], a, b);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (KIND == 5) {
                    c32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c32[i], 0, 0, 0);
                    c16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c16[i], 0, 0, 0);
                }
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int v = 0; v < 8; ++v) s += vv[v];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += c32[i][r]; for (int r = 0; r < 4; ++r) s += c16[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* tag, double per_iter)
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * 256); (void)hipMalloc(&cyc, 8 * 256);
    const int iters = 2048;
    hipLaunchKernelGGL(k_shape<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, 16);
    hipLaunchKernelGGL(k_shape<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-44s: %6.1f cycles per MFMA (or pair)\n", tag, (double)h[0] / (per_iter * iters));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("v_mfma_f32_32x32x2_f32", 16);
    run<1>("v_mfma_f32_16x16x4_f32", 16);
    run<2>("v_mfma_f32_32x32x16_bf16", 16);
    run<3>("v_mfma_f32_16x16x32_bf16", 16);
    run<4>("pair 32x32x16_bf16 + 16x16x4_f32", 16);
    run<5>("pair 32x32x16_bf16 + 16x16x32_bf16", 16);
    run<6>("32x32x16_bf16 + 2 independent v_fma_f32", 16);
    run<7>("32x32x16_bf16 + 4 independent v_fma_f32", 16);
    run<8>("32x32x16_bf16 + 6 independent v_fma_f32", 16);
    run<9>("32x32x16_bf16 + 8 independent v_fma_f32", 16);
    return 0;
}
