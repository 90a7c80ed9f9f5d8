// micro-benchmark (round 6): do the VALU instructions of ONE wave issue under the matrix instructions of ANOTHER wave of the same SIMD on gfx950?
// One 512-thread workgroup per CU (two waves per SIMD).  Waves 0-3 ("M") run a pure stream of independent v_mfma_f32_32x32x16_f16 (four accumulators),
// waves 4-7 ("V") a pure stream of v_fma_f32 - independent (8 chains) or one dependent chain - or LDS reads.  Each role's own cycle count is reported alone and
// together; a wave-priority variant raises the V waves.  What k_decoder2 (two independent 4-wave workgroups per CU) needs is "together ~ max(alone)";
// what it measures on the real kernel is "together ~ sum".
// Build: hipcc --offload-arch=gfx950 -O3 wave_roles.hip -o wave_roles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define M4 "v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
#define V8I "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define V8D "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"

// mode bits: 1 = M waves work, 2 = V waves work; vkind 0 = independent FMAs, 1 = dependent chain, 2 = mixed VALU + ds_read_b128; prio: V waves' priority
template <int VKIND>
__global__ __launch_bounds__(512, 2) void k_roles(float* out, long long* cyc, int iters, int mode, int prio)
{
    __shared__ float lds[4096];
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 1e-3f;
    __syncthreads();
    float s = 0.f;
    long long t0 = 0, t1 = 0;
    if (w < 4) {
        if (mode & 1) {
            f32x16 c0, c1, c2, c3;
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
            f16x8 ha, hb;
            for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)0.5f; hb[i] = (_Float16)(threadIdx.x * 1e-3f); }
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it)
                asm volatile(M4 M4 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(ha), "v"(hb));
            t1 = __builtin_readcyclecounter();
            for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
        }
    } else if (mode & 2) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        const float a = 0.5f, b = threadIdx.x * 1e-3f;
        float v0 = b, v1 = b + 1, v2 = b + 2, v3 = b + 3, v4 = b + 4, v5 = b + 5, v6 = b + 6, v7 = b + 7;
        t0 = __builtin_readcyclecounter();
        if (VKIND == 0) for (int it = 0; it < iters; ++it)
            asm volatile(V8I V8I V8I V8I V8I V8I V8I V8I : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
        if (VKIND == 1) for (int it = 0; it < iters; ++it)
            asm volatile(V8D V8D V8D V8D V8D V8D V8D V8D : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
        if (VKIND == 2) for (int it = 0; it < iters; ++it) {               // 8 x (ds_read_b128 + 7 FMAs on its values): 64 instructions
            const float4* p = reinterpret_cast<const float4*>(lds) + ((threadIdx.x + it) & 1023);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float4 q = p[(u * 64) & 1023];
                asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));
                v0 = fmaf(v0, q.x, q.y); v1 = fmaf(v1, q.z, q.w); v2 = fmaf(v2, q.x, a); v3 = fmaf(v3, q.y, a); v4 = fmaf(v4, q.z, a); v5 = fmaf(v5, q.w, a); v6 = fmaf(v6, q.x, b);
            }
        }
        t1 = __builtin_readcyclecounter();
        s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * w] = t0; cyc[2 * w + 1] = t1; }
}

template <int VKIND>
static void run(const char* what, int mode, int prio)
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * 512); (void)hipMalloc(&cyc, 8 * 16);
    const int iters = 2048;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_roles<VKIND>), dim3(256), dim3(512), 0, 0, out, cyc, iters, mode, prio);
    (void)hipDeviceSynchronize();
    long long h[16]; (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int w = 0; w < 4; ++w) { m += (double)(h[2 * w + 1] - h[2 * w]) / 4; v += (double)(h[2 * (w + 4) + 1] - h[2 * (w + 4)]) / 4; }
    printf("%-34s mode %d prio %d:  M waves %7.1f cycles per MFMA (8 per iteration)   V waves %6.2f cycles per VALU instruction (64 per iteration)\n", what, mode, prio,
           m / (8.0 * iters), v / (64.0 * iters));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("independent FMAs", 1, 0); run<0>("independent FMAs", 2, 0); run<0>("independent FMAs", 3, 0); run<0>("independent FMAs", 3, 1);
    run<1>("dependent FMA chain", 2, 0); run<1>("dependent FMA chain", 3, 0); run<1>("dependent FMA chain", 3, 1);
    run<2>("ds_read_b128 + 7 FMAs", 2, 0); run<2>("ds_read_b128 + 7 FMAs", 3, 0); run<2>("ds_read_b128 + 7 FMAs", 3, 1);
    return 0;
}
