/* nerfloam_hip_debug.h -- test, profiling and same-box A/B aids of libnerfloam_hip.so.  NOT part of the product surface
 * (include/nerfloam_hip.h): everything here reads or writes PROCESS-GLOBAL state of the library (relaxed atomics: no data race, but a
 * setter changes what every later call of every thread does), so a caller that needs the library re-entrant across threads and streams -
 * what nerfloam_hip.h promises - simply never calls these.  Every selection that matters to a caller has a per-call form there:
 * kernel_modes (decoder kernels), NlIterDesc.isect_lanes / nl_ray_intersect_lanes (lanes per ray).  Used by tests/, scripts/ and the
 * NL_* environment switches of nerf_loam_amd/_lib.py. */
#ifndef NERFLOAM_HIP_DEBUG_H
#define NERFLOAM_HIP_DEBUG_H
#include "nerfloam_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* process-wide DEFAULTS of the decoder kernel selection, used by the entry points without a kernel_modes argument and by kernel_modes
 * fields left 0 (modes: nerfloam_hip.h, nl_decoder_transpose_w2) */
int nl_decoder_set_gemm_mode(int mode);
int nl_decoder_get_gemm_mode(void);
int nl_decoder_set_wgrad2_mode(int mode);
int nl_decoder_get_wgrad2_mode(void);
int nl_decoder_set_layout(int layout);          /* process default of NL_KERNEL_LAYOUT: 0 = by slab count (default), 1 = one 8-wave workgroup per CU, 2 = two 4-wave workgroups */
int nl_decoder_get_layout(void);

int nl_geometry_set_sampler_mode(int mode);     /* nl_sample_rays: 0 = sequential walk per ray, 1 = step-parallel, 2 = by ray count (default); same results */
int nl_geometry_set_intersect_prune(int on);    /* nl_ray_intersect (tests): 0 = rays with more hits than the work-list kernel's list holds go to the sequential fallback instead of being pruned to the first 20 in place; 1 = default */
int nl_geometry_set_scan_single(int on);        /* prefix scans of 4097 .. 32 768 items: 1 = one launch (k_scan_single, default), 0 = the two launches of larger scans (equality tests, A/B) */
int nl_geometry_set_lanes_per_ray(int lpr);     /* nl_ray_intersect: 0 = the caller's choice / by ray count (default), or 4 / 8 / 16 / 32 lanes per ray for every call */
int nl_geometry_set_debug_buffer(void* dbg);   /* [blocks][8] int64 stamps of nl_ray_intersect's work-list kernel */
int nl_field_set_debug_buffer(void* dbg);      /* [blocks][8] int64 stamps of nl_trilinear_bwd's workgroups */
/* nl_trilinear_bwd is a latency chain per wave: while ONE round of resident workgroups (4 per compute unit) covers the samples with at
 * most 6 per 8-lane group, the launch's other workgroups leave at once (default on; 0 = every workgroup takes samples: A/B aid) */
int nl_field_set_one_round(int on);
int nl_field_set_midspan_flush(int min_steps_left);   /* A/B aid: nl_trilinear_bwd writes a full wave table out mid-span when its 8-lane groups have at least this
                                                         many sample steps left (default 2; < 0 = never: overflowing runs go to memory from their lane) */
int nl_field_set_probes(int n);               /* A/B aid: open-addressing probes of nl_trilinear_bwd's wave tables before a run goes straight to memory */
/* profiling aid: (256 + 8 x nslabs) x int64 device buffer receiving per-phase shader-clock stamps of the decoder kernel's workgroup 0 in its first 256 words and, under
 * layout 2, one 8-word record per workgroup behind them (start / end wall clock and shader cycles, HW_ID, XCC_ID, tiles) (NULL = off) */
int nl_decoder_set_debug_buffer(void* dbg);
/* MFMA lane-map self test (debug) */
int nl_mfma_selftest(const float* A32, const float* B32, float* D32, const float* A16, const float* B16, float* D16, void* stream);

#ifdef __cplusplus
}
#endif
#endif
