// nl_mesh.hip -- marching cubes over the per-voxel SDF grids of get_scores ("optionally a HIP marching-cubes to replace per-voxel CPU skimage").
//
// Reference behaviour: MeshExtractor.marching_cubes, /root/reference/src/utils/mesh_util.py:145-169 - for every surface voxel whose res^3 grid changes sign
// (:158-159), skimage.measure.marching_cubes(volume, 0, spacing = 1 / (res - 1)) on the CPU, one voxel at a time after a .cpu() per voxel; then
// verts = (verts - 0.5) * voxel_size + centre, faces offset by the running vertex count, concatenated in voxel order.
//
// Here: one workgroup per voxel, the grid in LDS, two launches around two prefix scans (nl_exclusive_scan_i32):
//   count: per voxel, the lattice edges whose end values change sign (= vertices) and the triangles of its (res - 1)^3 cells;
//   emit:  vertices in lattice-edge order (axis, ix, iy, iz), at v0 / (v0 - v1) along the edge (linear interpolation, what every marching-cubes variant
//          does), shared by the cells around the edge like skimage's per-volume vertex list; faces in cell order from the 256-case table nl_mc_table.h.
// Outputs land at the offsets the scans give: the order is the reference's voxel order, the launch is deterministic.
// The case table is derived from the cube's geometry (scripts/gen_mc_table.py); skimage's Lewiner tables are not available in this image, so the
// TRIANGULATION is not pinned to the reference (oracle/mc_oracle.py states what is: vertex set, skip rule, world map, closed oriented surface).
#include "nl_common.h"
#include "nl_mc_table.h"

#define NL_MC_THREADS 256
#define NL_MC_MAX_RES 16                                    // LDS: 16 res^3 bytes (grid + three vertex-id planes)

struct McArgs {
    const float* sdf; const float* centres; int centre_stride; int n_vox, res; float voxel_size;
    int* n_verts; int* n_tris;
    const int* vert_off; const int* tri_off; float* verts; int* faces;
};

__device__ __forceinline__ int mc_wave_incl_scan(int v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if ((int)(threadIdx.x & 63) >= d) v += o;
    }
    return v;
}

// exclusive prefix of v over the workgroup's 256 threads in thread order, total -> *total; tmp: 8 ints of LDS.  Two barriers.
__device__ __forceinline__ int mc_block_excl_scan(int v, int* tmp, int* total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int incl = mc_wave_incl_scan(v);
    __syncthreads();                                         // (tmp of the previous call has been read)
    if (lane == 63) tmp[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NL_MC_THREADS / 64; ++i) { const int t = tmp[i]; if (i < w) base += t; tot += t; }
    *total = tot;
    return base + incl - v;
}

__device__ __forceinline__ int mc_cell_config(const float* vol, int res, int i, int j, int k)
{
    int cfg = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (vol[((i + (c & 1)) * res + j + ((c >> 1) & 1)) * res + k + ((c >> 2) & 1)] < 0.f) cfg |= 1 << c;
    return cfg;
}

// lattice edge slot s = axis * res^3 + (i * res + j) * res + k: the edge from point (i, j, k) one step along `axis` (no edge where that leaves the grid)
__device__ __forceinline__ bool mc_edge(const float* vol, int res, int r3, int s, int* axis, int* p, float* v0, float* v1)
{
    const int a = s / r3, q = s - a * r3;
    p[0] = q / (res * res); p[1] = (q / res) % res; p[2] = q % res;
    *axis = a;
    if (p[a] + 1 >= res) return false;
    const int step = a == 0 ? res * res : (a == 1 ? res : 1);
    *v0 = vol[q]; *v1 = vol[q + step];
    return (*v0 < 0.f) != (*v1 < 0.f);
}

template <bool EMIT>
__global__ __launch_bounds__(NL_MC_THREADS) void k_marching_cubes(McArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mc_lds[];
    const int res = a.res, r3 = res * res * res, v = blockIdx.x, tid = threadIdx.x;
    float* vol = reinterpret_cast<float*>(mc_lds);
    int* vid = reinterpret_cast<int*>(mc_lds) + r3;           // [3][res^3] vertex id of an edge slot within this voxel (emit pass)
    __shared__ int tmp[8];
    __shared__ float red[2][NL_MC_THREADS / 64];
    const float* src = a.sdf + (size_t)v * r3;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < r3; i += NL_MC_THREADS) { const float x = src[i]; vol[i] = x; mn = fminf(mn, x); mx = fmaxf(mx, x); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { mn = fminf(mn, __shfl_xor(mn, d, 64)); mx = fmaxf(mx, __shfl_xor(mx, d, 64)); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = mn; red[1][tid >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const bool skip = mn > 0.f || mx < 0.f;                  // mesh_util.py:158-159
    if (!EMIT) {
        int nv = 0, nt = 0;
        if (!skip) {
            for (int s = tid; s < 3 * r3; s += NL_MC_THREADS) { int ax, p[3]; float v0, v1; nv += mc_edge(vol, res, r3, s, &ax, p, &v0, &v1) ? 1 : 0; }
            const int rc = res - 1, nc = rc * rc * rc;
            for (int c = tid; c < nc; c += NL_MC_THREADS) nt += NL_MC_NTRI[mc_cell_config(vol, res, c / (rc * rc), (c / rc) % rc, c % rc)];
        }
        int tv, tt;
        mc_block_excl_scan(nv, tmp, &tv);
        mc_block_excl_scan(nt, tmp, &tt);
        if (tid == 0) { a.n_verts[v] = tv; a.n_tris[v] = tt; }
        return;
    }
    if (skip) return;
    const int voff = a.vert_off[v], toff = a.tri_off[v];
    const float cx = a.centres[(size_t)v * a.centre_stride], cy = a.centres[(size_t)v * a.centre_stride + 1], cz = a.centres[(size_t)v * a.centre_stride + 2];
    const float spacing = 1.0f / (float)(res - 1);           // mesh_util.py:149 (python float -> skimage's float32 vertices)
    int base = 0;
    for (int s0 = 0; s0 < 3 * r3; s0 += NL_MC_THREADS) {
        const int s = s0 + tid;
        int ax = 0, p[3] = {0, 0, 0}; float v0 = 0.f, v1 = 0.f;
        const bool on = s < 3 * r3 && mc_edge(vol, res, r3, s, &ax, p, &v0, &v1);
        int tot;
        const int id = base + mc_block_excl_scan(on ? 1 : 0, tmp, &tot);
        if (s < 3 * r3) vid[s] = on ? id : -1;
        if (on) {
            const float t = v0 / (v0 - v1);
            float q[3] = {(float)p[0], (float)p[1], (float)p[2]};
            q[ax] = q[ax] + t;
            float* o = a.verts + 3 * (size_t)(voff + id);
            o[0] = (q[0] * spacing - 0.5f) * a.voxel_size + cx;   // :162-164
            o[1] = (q[1] * spacing - 0.5f) * a.voxel_size + cy;
            o[2] = (q[2] * spacing - 0.5f) * a.voxel_size + cz;
        }
        base += tot;
    }
    __syncthreads();
    const int rc = res - 1, nc = rc * rc * rc;
    base = 0;
    for (int c0 = 0; c0 < nc; c0 += NL_MC_THREADS) {
        const int c = c0 + tid;
        int cfg = 0, i = 0, j = 0, k = 0;
        if (c < nc) { i = c / (rc * rc); j = (c / rc) % rc; k = c % rc; cfg = mc_cell_config(vol, res, i, j, k); }
        const int n = NL_MC_NTRI[cfg];
        int tot;
        const int first = base + mc_block_excl_scan(n, tmp, &tot);
        for (int t = 0; t < n; ++t) {
            int* f = a.faces + 3 * (size_t)(toff + first + t);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int e = NL_MC_TRI[cfg][3 * t + m], ax = e >> 2, b0 = e & 1, b1 = (e >> 1) & 1;
                // the edge's lower corner inside the cell: the other two axes' bits (lower axis first)
                const int di = ax == 0 ? 0 : b0, dj = ax == 0 ? b0 : (ax == 1 ? 0 : b1), dk = ax == 2 ? 0 : b1;
                f[m] = voff + vid[ax * r3 + ((i + di) * res + j + dj) * res + k + dk];
            }
        }
        base += tot;
    }
}

extern "C" {

/* count pass: n_verts[v], n_tris[v] of every voxel (0 for a voxel the reference skips) */
int nl_mc_count(const float* sdf, int n_vox, int res, int* n_verts, int* n_tris, void* stream)
{
    if (n_vox < 0 || res < 2 || res > NL_MC_MAX_RES || (n_vox > 0 && (!sdf || !n_verts || !n_tris))) return NL_ERR_INVALID_ARG;
    if (n_vox == 0) return NL_OK;
    McArgs a = {};
    a.sdf = sdf; a.n_vox = n_vox; a.res = res; a.n_verts = n_verts; a.n_tris = n_tris;
    hipLaunchKernelGGL((k_marching_cubes<false>), dim3(n_vox), dim3(NL_MC_THREADS), (size_t)4 * res * res * res, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* emit pass: vert_off / tri_off = exclusive scans of the count pass; verts[total_verts][3] (world), faces[total_tris][3] (indices into verts) */
int nl_mc_emit(const float* sdf, const float* centres, int centre_stride, int n_vox, int res, float voxel_size, const int* vert_off, const int* tri_off,
               float* verts, int* faces, void* stream)
{
    if (n_vox < 0 || res < 2 || res > NL_MC_MAX_RES || centre_stride < 3 || (n_vox > 0 && (!sdf || !centres || !vert_off || !tri_off))) return NL_ERR_INVALID_ARG;
    if (n_vox == 0) return NL_OK;
    McArgs a = {};
    a.sdf = sdf; a.centres = centres; a.centre_stride = centre_stride; a.n_vox = n_vox; a.res = res; a.voxel_size = voxel_size;
    a.vert_off = vert_off; a.tri_off = tri_off; a.verts = verts; a.faces = faces;
    const size_t lds = (size_t)16 * res * res * res;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_marching_cubes<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return NL_ERR_LAUNCH;
    hipLaunchKernelGGL((k_marching_cubes<true>), dim3(n_vox), dim3(NL_MC_THREADS), lds, (hipStream_t)stream, a);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

}
