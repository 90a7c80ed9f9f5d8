// f16_pair.hip -- what the "f16 pair" decoder arithmetic relies on, checked on the device:
//   (1) v_mfma_f32_32x32x16_f16 keeps SUBNORMAL f16 inputs (the low term of a two-term fp16 split of a small fp32 value is subnormal);
//   (2) its issue rate next to v_mfma_f32_32x32x16_bf16 (same 8 passes);
//   (3) the error of a 64 x 256 x 256 product formed from two-term fp16 splits (3 and 4 partial products) against fp64, next to the
//       eight-product bf16 split and a plain fp32 fmaf chain.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/f16_pair.hip -o scripts/micro/f16_pair && scripts/micro/f16_pair
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_subnormal(float* out)
{
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[m][k]: lane (m = lane & 31, kgroup = lane >> 5) holds k = 8 * kgroup + i.  Row 0: one subnormal 2^-20 at k = 0; row 1: 2^-24 (the
    // smallest subnormal) at k = 0; row 2: normal 2^-14.  B[k][n] = 1 at k = 0 for every n; B[k=1][n] = 2^-20 (subnormal) with A[3][1] = 2^-20.
    if (lane == 0) a[0] = (_Float16)9.5367431640625e-07f;        // 2^-20
    if (lane == 1) a[0] = (_Float16)5.9604644775390625e-08f;     // 2^-24
    if (lane == 2) a[0] = (_Float16)6.103515625e-05f;            // 2^-14
    if (lane == 3) a[1] = (_Float16)9.5367431640625e-07f;
    if (lane < 32) { b[0] = (_Float16)1.f; b[1] = (_Float16)9.5367431640625e-07f; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    // C[row][col]: lane = col (+32: rows 4..7 of each group of 8), reg r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (lane == 0) { out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3]; }
}

template <bool F16>
__global__ void k_rate(float* out, int iters)
{
    h8 ah, bh; b8 ab, bb;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(0.001f * (threadIdx.x + i)); bh[i] = (_Float16)(0.002f * i); ab[i] = (__bf16)(0.001f * (threadIdx.x + i)); bb[i] = (__bf16)(0.002f * i); }
    f32x16 c[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) c[t][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            c[t] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[t], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[t], 0, 0, 0);
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += c[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave: C[32][32] = A[32][256] B[256][32] from split planes; mode 0: f16 pair 3 products, 1: f16 pair 4 products, 2: bf16 eight products
__device__ inline unsigned short f16_bits(float x) { _Float16 h = (_Float16)x; return __builtin_bit_cast(unsigned short, h); }
__global__ void k_gemm(const float* A, const float* B, float* C, int mode, float sa, float sb)
{
    const int lane = threadIdx.x & 63, m = lane & 31, g = lane >> 5;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    for (int s = 0; s < 16; ++s) {
        float av[8], bv[8];
        for (int i = 0; i < 8; ++i) { av[i] = A[m * 256 + 16 * s + 8 * g + i]; bv[i] = B[(16 * s + 8 * g + i) * 32 + m]; }
        if (mode < 2) {
            h8 ah, al, bh, bl;
            for (int i = 0; i < 8; ++i) {
                const float x = av[i] * sa, y = bv[i] * sb;
                ah[i] = (_Float16)x; al[i] = (_Float16)(x - (float)ah[i]);
                bh[i] = (_Float16)y; bl[i] = (_Float16)(y - (float)bh[i]);
            }
            if (mode == 1) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
        } else {
            b8 ap[3], bp[3];
            for (int i = 0; i < 8; ++i) {
                float x = av[i], y = bv[i];
                for (int p = 0; p < 3; ++p) {
                    const float tx = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u), ty = __uint_as_float(__float_as_uint(y) & 0xFFFF0000u);
                    ap[p][i] = (__bf16)tx; bp[p][i] = (__bf16)ty; x -= tx; y -= ty;
                }
            }
            for (int pa = 2; pa >= 0; --pa)
                for (int pb = 2; pb >= 0; --pb)
                    if (pa + pb <= 3) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[pa], bp[pb], c, 0, 0, 0);
        }
    }
    const float inv = mode < 2 ? 1.0f / (sa * sb) : 1.0f;
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + m] = c[r] * inv;
}

int main()
{
    float* d; hipMalloc(&d, 1 << 22);
    float h[4];
    hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("subnormal f16 inputs through v_mfma_f32_32x32x16_f16: 2^-20 * 1 = %g (want %g), 2^-24 * 1 = %g (want %g), 2^-14 * 1 = %g, 2^-20 * 2^-20 = %g (want %g)\n",
           h[0], ldexp(1.0, -20), h[1], ldexp(1.0, -24), h[2], h[3], ldexp(1.0, -40));
    printf("  -> subnormal inputs %s\n", (h[0] == (float)ldexp(1.0, -20) && h[1] == (float)ldexp(1.0, -24) && h[3] == (float)ldexp(1.0, -40)) ? "KEPT" : "FLUSHED");
    // rate: 256 CUs x 4 waves
    for (int which = 0; which < 2; ++which) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which) hipLaunchKernelGGL(k_rate<true>, dim3(256), dim3(256), 0, 0, d, iters);
            else       hipLaunchKernelGGL(k_rate<false>, dim3(256), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("%s 32x32x16: %.3f ms, %.0f TFLOP/s\n", which ? "f16 " : "bf16", ms, flops / ms * 1e-9);
    }
    // accuracy
    std::vector<float> A(32 * 256), B(256 * 32), C(32 * 32);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX; };
    for (auto& x : A) { float v = (rnd() + rnd() + rnd() - 1.5f); x = v > 0 ? v : 0.f; }       // relu-like activations
    for (auto& x : B) x = (rnd() * 2 - 1) / 16;
    float *dA = d, *dB = d + 32 * 256, *dC = d + 2 * 32 * 256;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(32 * 32, 0.0);
    std::vector<float> chain(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; float f = 0.f;
        for (int k = 0; k < 256; ++k) { s += (double)A[i * 256 + k] * B[k * 32 + j]; f = fmaf(A[i * 256 + k], B[k * 32 + j], f); }
        ref[i * 32 + j] = s; chain[i * 32 + j] = f;
    }
    auto report = [&](const char* name, const float* got) {
        double mx = 0, rms = 0;
        for (int i = 0; i < 32 * 32; ++i) { const double e = got[i] - ref[i]; mx = fmax(mx, fabs(e)); rms += e * e; }
        printf("%-34s max |err| %.3e  rms %.3e\n", name, mx, sqrt(rms / 1024));
    };
    report("fp32 fmaf chain (host)", chain.data());
    const char* names[3] = {"f16 pair, 3 products (2^4, 2^8)", "f16 pair, 4 products (2^4, 2^8)", "bf16 split, 8 products"};
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode, 16.f, 256.f);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        report(names[mode], C.data());
    }
    hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0, 1.f, 1.f);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    report("f16 pair, 3 products, unscaled", C.data());
    return 0;
}
