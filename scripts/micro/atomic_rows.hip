// micro-benchmark: what bounds the embedding scatter on an accumulated map (nl_field.hip k_trilinear_bwd: 16 lanes add one 64-byte fp32 row
// with global_atomic_add_f32, rows scattered over a [E,16] fp32 array)?  Rate of such row atomics against (a) the size of the array the rows
// fall in (3.4 MB = the single-scan map, 71 MB = the 150-scan map, 1 GB), (b) the number of DISTINCT rows the kernel touches (a compact
// accumulator would keep the same rows in a small array), (c) plain stores of the same rows as the ceiling.
// Build: hipcc --offload-arch=gfx950 -O3 atomic_rows.hip -o atomic_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every 16-lane quarter of a wave adds `per_quarter` rows; row ids = hash(counter) % distinct, spread over the array by `stride_rows`
template <int MODE>                                   // 0 = atomicAdd, 1 = plain store
__global__ __launch_bounds__(256) void k_rows(float* acc, unsigned distinct, unsigned stride_rows, int per_quarter, unsigned seed)
{
    const unsigned q = (blockIdx.x * 256u + threadIdx.x) >> 4, c = threadIdx.x & 15;
    for (int i = 0; i < per_quarter; ++i) {
        const unsigned row = (hash32(q * 977u + (unsigned)i * 7919u + seed) % distinct) * stride_rows;
        float* p = acc + (size_t)row * 16 + c;
        if (MODE == 0) atomicAdd(p, 1.0f);
        else *p = (float)i;
    }
}

// round 5: SAME-ROW contention and what replicated accumulators buy.  Of every quarter's rows a fraction `hot_pct` goes to one of `n_hot` hot rows (the
// near-sensor voxels every ray of a scan crosses), the rest to distinct rows; with `replicas` copies of the accumulator a wave adds into copy (wave id % replicas).
__global__ __launch_bounds__(256) void k_hot(float* acc, unsigned distinct, unsigned n_hot, unsigned hot_pct, unsigned replicas, size_t rep_stride, int per_quarter, unsigned seed)
{
    const unsigned q = (blockIdx.x * 256u + threadIdx.x) >> 4, c = threadIdx.x & 15;
    const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    float* base = acc + (size_t)(wave % replicas) * rep_stride;
    for (int i = 0; i < per_quarter; ++i) {
        const unsigned h = hash32(q * 977u + (unsigned)i * 7919u + seed);
        const unsigned row = (h % 100u) < hot_pct ? (hash32(h) % n_hot) : n_hot + (hash32(h ^ 0x9E3779B9u) % distinct);
        atomicAdd(base + (size_t)row * 16 + c, 1.0f);
    }
}

int main()
{
    const size_t max_rows = (size_t)1 << 24;                              // 16.8 M rows = 1 GB
    float* acc = nullptr;
    if (hipMalloc(&acc, max_rows * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(acc, 0, max_rows * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char* name; unsigned distinct, stride; };
    const Case cases[] = {
        {"53 k rows, dense (3.4 MB: the single-scan map)", 53478u, 1u},
        {"1.11 M rows, dense (71 MB: the 150-scan map)", 1111128u, 1u},
        {"16.8 M rows, dense (1 GB)", 1u << 24, 1u},
        {"30 k distinct rows spread over 71 MB (a 2048-ray step on the 150-scan map)", 30000u, 37u},
        {"30 k distinct rows, compact (1.9 MB)", 30000u, 1u},
        {"200 k distinct rows spread over 1 GB", 200000u, 83u},
        {"200 k distinct rows, compact (12.8 MB)", 200000u, 1u},
    };
    const int blocks_list[] = {750, 2048};                                // the scatter's launch at 2048 rays on the large map / its full grid
    for (int mode = 0; mode < 2; ++mode)
        for (const Case& cs : cases)
            for (int blocks : blocks_list) {
                const int per_quarter = 32;                                // a wave table's 128 rows = 4 quarters x 32
                const double rows = (double)blocks * 16 * per_quarter;
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k_rows<0>, dim3(blocks), dim3(256), 0, 0, acc, cs.distinct, cs.stride, per_quarter, 17u + rep);
                    else           hipLaunchKernelGGL(k_rows<1>, dim3(blocks), dim3(256), 0, 0, acc, cs.distinct, cs.stride, per_quarter, 17u + rep);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("%-7s %-78s %5d workgroups: %8.1f us for %7.0f k rows = %6.2f G rows/s\n", mode == 0 ? "atomic" : "store", cs.name, blocks, best * 1e3, rows / 1e3,
                       rows / (best * 1e-3) / 1e9);
            }
    // hot rows: 750 workgroups x 16 quarters x 32 rows = 384 k row atomics (a 2048-ray step on the 150-scan map), 30 k cold rows
    for (unsigned hot_pct : {0u, 10u, 30u})
        for (unsigned n_hot : {48u, 512u})
            for (unsigned replicas : {1u, 2u, 4u, 8u, 16u, 64u}) {
                if (hot_pct == 0 && (replicas > 1 || n_hot > 48)) continue;
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_hot, dim3(750), dim3(256), 0, 0, acc, 30000u, n_hot, hot_pct, replicas, (size_t)40000 * 16, 32, 17u + rep);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("hot     %2u %% of the rows on %3u hot rows, %2u accumulator copies: %8.1f us for 384 k row atomics\n", hot_pct, n_hot, replicas, best * 1e3);
            }
    hipFree(acc);
    return 0;
}
