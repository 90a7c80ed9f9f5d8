// tests/host_harness.cpp -- TEST ONLY.  Compiles nerf_loam_amd/csrc/nl_device_math.h (the exact
// inline functions the HIP kernels execute per lane) for the host with g++, so the CPU test-suite
// can check the device logic against the oracle without a GPU.  Not linked into the product.
#include <cstring>
#include <vector>

#include "../nerf_loam_amd/csrc/nl_device_math.h"

extern "C" {

void hh_ray_intersect(int N, const float* o, const float* d, const float* centres, const int* structure,
                      float voxel_size, float max_distance, int* idx, float* t0, float* t1, int* count)
{
    for (int r = 0; r < N; ++r) {
        int hi[NL_MAX_HITS]; float h0[NL_MAX_HITS], h1[NL_MAX_HITS];
        NlLocalStack stk;
        int cnt = nl_octree_walk(centres, structure, o[3 * r], o[3 * r + 1], o[3 * r + 2], d[3 * r], d[3 * r + 1], d[3 * r + 2],
                                 voxel_size * 0.5f, NL_MAX_HITS, stk, hi, h0, h1);
        int valid = nl_sort_cull_hits(cnt, hi, h0, h1, max_distance);
        for (int l = 0; l < NL_MAX_HITS; ++l) {
            bool v = l < cnt;
            idx[r * NL_MAX_HITS + l] = v ? hi[l] : -1;
            t0[r * NL_MAX_HITS + l] = v ? h0[l] : max_distance;
            t1[r * NL_MAX_HITS + l] = v ? h1[l] : max_distance;
        }
        count[r] = valid;
    }
}

// raw DFS-order hits (grid.svo_intersect semantics)
void hh_svo_intersect_raw(int N, const float* o, const float* d, const float* centres, const int* structure,
                          float voxel_size, int n_max, int* idx, float* t0, float* t1)
{
    for (int r = 0; r < N; ++r) {
        int hi[NL_MAX_HITS]; float h0[NL_MAX_HITS], h1[NL_MAX_HITS];
        NlLocalStack stk;
        int cnt = nl_octree_walk(centres, structure, o[3 * r], o[3 * r + 1], o[3 * r + 2], d[3 * r], d[3 * r + 1], d[3 * r + 2],
                                 voxel_size * 0.5f, n_max, stk, hi, h0, h1);
        for (int l = 0; l < n_max; ++l) {
            bool v = l < cnt;
            idx[r * n_max + l] = v ? hi[l] : -1; t0[r * n_max + l] = v ? h0[l] : 0.f; t1[r * n_max + l] = v ? h1[l] : 0.f;
        }
    }
}

// hit arrays [R,20] for the R hit rays in rank order; P = batch max hits; ray_ids[R] = noise ids
void hh_sample(int R, const int* idx, const float* t0, const float* t1, int P, float step_size, unsigned seed,
               int use_hash, int tail_always, const unsigned* ray_ids, int S_cap,
               int* s_idx, float* s_depth, float* s_dist, int* count)
{
    for (int r = 0; r < R; ++r) {
        NlTailCtx tc; int first;
        nl_sampler_layout(r, R, &tc.j_in_row, &tc.rays_in_row, &first);
        tc.row_first_idx = idx + (size_t)first * NL_MAX_HITS;
        tc.row_first_count = NL_MAX_HITS;
        tc.row_first_bias = 0;
        tc.tail_always = (tail_always & 1) != 0;                       // bit 1 selects the step-parallel formulation
        const unsigned rid = ray_ids[r];
        auto noise = [&](int s) -> float { return use_hash ? nl_noise(seed, rid, (unsigned)s) : 0.5f; };
        auto emit = [&](int s, int v, float depth, float dist) {
            if (s < S_cap) { s_idx[(size_t)r * S_cap + s] = v; s_depth[(size_t)r * S_cap + s] = depth; s_dist[(size_t)r * S_cap + s] = dist < 0.f ? 0.f : dist; }
        };
        count[r] = (tail_always & 2) ? nl_sample_walk_steps(idx + (size_t)r * NL_MAX_HITS, t0 + (size_t)r * NL_MAX_HITS, t1 + (size_t)r * NL_MAX_HITS, P,
                                                             step_size, tc, noise, emit)
                                     : nl_sample_walk(idx + (size_t)r * NL_MAX_HITS, t0 + (size_t)r * NL_MAX_HITS, t1 + (size_t)r * NL_MAX_HITS, P,
                                                      step_size, tc, noise, emit);
    }
}

void hh_trilinear(int P, const float* x, const float* c, float vs, float* p_out, float* w_out)
{
    for (int i = 0; i < P; ++i) { nl_trilinear_p(x + 3 * i, c + 3 * i, vs, p_out + 3 * i); nl_trilinear_w(p_out + 3 * i, w_out + 8 * i); }
}
void hh_trilinear_dp(int P, const float* p, const float* dot, float* dp)
{
    for (int i = 0; i < P; ++i) nl_trilinear_dp(p + 3 * i, dot + 8 * i, dp + 3 * i);
}

void hh_adam_bf16(int n, uint16_t* p, const uint16_t* g, uint16_t* m, uint16_t* v, double lr, int step)
{
    NlAdamHyper h = nl_adam_hyper(lr, step, 0.9, 0.999, 1e-8);
    for (int i = 0; i < n; ++i) nl_adam_bf16(p + i, g[i], m + i, v + i, h);
}
void hh_adam_f32(int n, float* p, const float* g, float* m, float* v, double lr, int step)
{
    NlAdamHyper h = nl_adam_hyper(lr, step, 0.9, 0.999, 1e-8);
    for (int i = 0; i < n; ++i) nl_adam_f32(p + i, g[i], m + i, v + i, h);
}
void hh_rodrigues(const float* w, float* R) { nl_rodrigues(w, R); }
void hh_rodrigues_bwd(const float* w, const float* G, float* gw) { nl_rodrigues_bwd(w, G, gw); }
void hh_noise(unsigned seed, int n_rays, const unsigned* rays, int n_steps, float* out)
{
    for (int r = 0; r < n_rays; ++r) for (int s = 0; s < n_steps; ++s) out[(size_t)r * n_steps + s] = nl_noise(seed, rays[r], (unsigned)s);
}
// loss: masks + gradient for packed valid samples
void hh_loss(int P, const float* sdf, const float* z, const float* d, float w_fs, float w_sdf, float two_over_n,
             float fs_weight, float sdf_weight, float tau, float max_depth, float* ds, int* front, int* sdfm)
{
    NlLossScalars ls; ls.w_fs = w_fs; ls.w_sdf = w_sdf; ls.two_over_n = two_over_n; ls.fs_weight = fs_weight; ls.sdf_weight = sdf_weight;
    ls.tau = tau; ls.max_depth = max_depth;
    for (int i = 0; i < P; ++i) {
        bool f, m; float q1, q2;
        nl_loss_masks(z[i], d[i], tau, max_depth, &f, &m);
        ds[i] = nl_loss_grad(sdf[i], z[i], d[i], f, m, ls, &q1, &q2);
        front[i] = f; sdfm[i] = m;
    }
}

void hh_split3_bf16(int n, const float* v, uint16_t* hi, uint16_t* mid, uint16_t* lo)
{
    for (int i = 0; i < n; ++i) nl_split3_bf16(v[i], &hi[i], &mid[i], &lo[i]);
}

void hh_f16_convert(int n, const float* v, uint16_t* h, float* back)
{
    for (int i = 0; i < n; ++i) { h[i] = nl_f32_to_f16(v[i]); back[i] = nl_f16_to_f32(h[i]); }
}

void hh_split2_f16(int n, const float* v, float scale, uint16_t* hi, uint16_t* lo)
{
    for (int i = 0; i < n; ++i) nl_split2_f16(v[i], scale, &hi[i], &lo[i]);
}

void hh_select_key(int n, uint32_t seed, uint32_t* out)
{
    for (int i = 0; i < n; ++i) out[i] = nl_select_key(seed, (uint32_t)i);
}

void hh_unit_dirs(int n, const float* p, float* d, float* norm)
{
    for (int i = 0; i < n; ++i) norm[i] = nl_unit_dir(p[3 * i], p[3 * i + 1], p[3 * i + 2], &d[3 * i], &d[3 * i + 1], &d[3 * i + 2]);
}

}  // extern "C"
