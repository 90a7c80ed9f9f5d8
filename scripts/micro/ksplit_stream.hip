// micro-benchmark: the chained decoder's forward stage (48 weight fragments shared through LDS per stage, 144 bf16 MFMAs per SIMD and
// stage, one barrier per stage) at ONE wave per SIMD owning all 16 k-steps (hb: 192 registers) against TWO waves per SIMD that split
// K (8 k-steps = 96 registers each; their partial accumulators would be exchanged through LDS once per stage - not modelled here),
// with VPM independent VALU instructions next to every MFMA (the real kernel carries ~2.6: epilogues, splits, mask work).
// Question: how much of the per-instruction tax of a lone wave (mfma_valu.hip) does the second wave per SIMD buy back?
// Build: hipcc --offload-arch=gfx950 -O3 ksplit_stream.hip -o ksplit_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define PLANE (256 * 256 * 2)
__device__ __forceinline__ uint4 bload4(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int WAVES, int VPM>
__global__ __launch_bounds__(WAVES * 64) void k_stage(const unsigned short* W, const uint4* Hinit, float* out, long long* cyc, int stages)
{
    __shared__ __attribute__((aligned(16))) uint4 ring[2][48][64];
    constexpr int KS = 64 / WAVES;                       // k-steps per wave and stage: 16 (one wave per SIMD) or 8 (K split over a wave pair)
    constexpr int NF = 48 / WAVES;                       // fragments this wave moves per stage
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = WAVES == 8 ? (w >> 2) : 0;            // which half of K
    const rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, 3 * PLANE, 0x00020000);
    const int voff = lane * 16;
    uint4 hb[KS][3];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) hb[s][p] = Hinit[((s + KS * kh) * 3 + p) * 64 + lane];
    float vv[4] = {1.f + lane, 2.f, 3.f, 4.f};
    const float va = 0.999f, vb = 1e-3f;
    uint4 st[3];
    auto fill_load = [&](int j, int i) { const int f = w * NF + i; st[i % 3] = bload4(rw, voff, (f % 3) * PLANE + ((j & 7) * 16 + f / 3) * 1024); };
    auto fill_store = [&](int buf, int i) { ring[buf][w * NF + i][lane] = st[i % 3]; };
    for (int i = 0; i < NF; ++i) { fill_load(0, i); fill_store(0, i); }
    __syncthreads();
    f32x16 total;
    for (int r = 0; r < 16; ++r) total[r] = 0.f;
    int lbuf = 0;
    uint4 af[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) af[0][p] = ring[0][3 * (KS * kh) + p][lane];
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int j = 0; j < stages; ++j) {
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                c = MFMA(__builtin_bit_cast(bf16x8, af[s & 1][m / 3]), __builtin_bit_cast(bf16x8, hb[s][m % 3]), c);
                if (m < 3 && s + 1 < KS) af[(s + 1) & 1][m] = ring[lbuf][3 * (KS * kh + s + 1) + m][lane];
                // fills: one fragment every (KS / NF)-th k-step... NF fragments over KS k-steps: load at slot 4, store two k-steps later at slot 3
                if (m == 3 && s >= 2 && s - 2 < NF) fill_store(lbuf ^ 1, s - 2);
                if (m == 4 && s < NF) fill_load(j + 1, s);
#pragma unroll
                for (int v = 0; v < VPM; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(vv[(m * VPM + v) & 3]) : "v"(va), "v"(vb));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();
        lbuf ^= 1;
#pragma unroll
        for (int p = 0; p < 3; ++p) af[0][p] = ring[lbuf][3 * (KS * kh) + p][lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) total[r] += c[r];
    }
    const long long t1 = __builtin_readcyclecounter();
    float sres = vv[0] + vv[1] + vv[2] + vv[3];
    for (int r = 0; r < 16; ++r) sres += total[r];
    out[blockIdx.x * WAVES * 64 + tid] = sres;
    if ((tid & 63) == 0 && blockIdx.x == 0) { cyc[2 * (tid >> 6)] = t0; cyc[2 * (tid >> 6) + 1] = t1; }
}

template <int WAVES, int VPM>
static void run(const unsigned short* W, const uint4* H)
{
    const int blocks = 256, stages = 320;
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * WAVES * 64 * blocks); (void)hipMalloc(&cyc, 8 * 16);
    hipLaunchKernelGGL((k_stage<WAVES, VPM>), dim3(blocks), dim3(WAVES * 64), 0, 0, W, H, out, cyc, 16);
    hipLaunchKernelGGL((k_stage<WAVES, VPM>), dim3(blocks), dim3(WAVES * 64), 0, 0, W, H, out, cyc, stages);
    (void)hipDeviceSynchronize();
    long long hh[16]; (void)hipMemcpy(hh, cyc, 128, hipMemcpyDeviceToHost);
    long long lo = hh[0], hi = hh[1];
    for (int w = 0; w < WAVES; ++w) { lo = hh[2 * w] < lo ? hh[2 * w] : lo; hi = hh[2 * w + 1] > hi ? hh[2 * w + 1] : hi; }
    printf("%d wave(s) per SIMD, %d VALU per MFMA: %6.1f cycles per MFMA and SIMD (%7.0f per stage of 144)  [%s]\n", WAVES / 4, VPM,
           (double)(hi - lo) / (144.0 * stages), (double)(hi - lo) / stages, hipGetErrorString(hipGetLastError()));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    unsigned short* W; uint4* H;
    (void)hipMalloc(&W, 3 * PLANE); (void)hipMalloc(&H, 48 * 64 * 16);
    (void)hipMemset(W, 0, 3 * PLANE); (void)hipMemset(H, 0, 48 * 64 * 16);
    run<4, 0>(W, H); run<4, 2>(W, H); run<4, 3>(W, H); run<4, 4>(W, H);
    run<8, 0>(W, H); run<8, 2>(W, H); run<8, 3>(W, H); run<8, 4>(W, H);
    return 0;
}
