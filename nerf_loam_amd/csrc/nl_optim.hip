// nl_optim.hip -- fused optimiser step: Adam on the bf16 voxel embeddings (bf16 state, the seven
// rounding points of torch's bf16 tensors), on the fp32 decoder block, and on the SE3 pose 6-vectors
// (with the Rodrigues tail of the pose gradient).  HBM-bound element-wise kernels.
//
// Reference behaviour: torch.optim.Adam as constructed in src/variations/render_helpers.py:341-353
// and :448-450 (default betas/eps, fresh state per call), src/se3pose.py:18-35 for the pose tail.
#include "nl_common.h"
#include "../../include/nerfloam_hip.h"

// embeddings: g = bf16(fp32 accumulator) ; accumulator is reset for the next iteration
// optimiser state block on the device: int32 step counter (+3 pad) followed by NlAdamHyper[3] = {embeddings, decoder,
// pose}.  k_adam_prepare advances the step and recomputes the bias corrections ON THE DEVICE, so the whole optimiser
// step has no host-side scalar: a captured hipGraph replays correctly for every iteration of a call.
__global__ void k_adam_prepare(int* __restrict__ state, double lr_emb, double lr_dec, double lr_pose)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int step = state[0] + 1;
    state[0] = step;
    NlAdamHyper* h = reinterpret_cast<NlAdamHyper*>(state + 4);
    h[0] = nl_adam_hyper(lr_emb, step, 0.9, 0.999, 1e-8);
    h[1] = nl_adam_hyper(lr_dec, step, 0.9, 0.999, 1e-8);
    h[2] = nl_adam_hyper(lr_pose, step, 0.9, 0.999, 1e-8);
}

__global__ void k_adam_emb(uint16_t* __restrict__ p, float* __restrict__ g_acc, uint16_t* __restrict__ m, uint16_t* __restrict__ v,
                           long long n, const NlAdamHyper* __restrict__ hp)
{
    const NlAdamHyper h = *hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float ga = g_acc[i];
        const uint16_t pm = m[i], pv = v[i];
        if (ga == 0.0f && pm == 0 && pv == 0) continue;      // 0/(0+eps) = 0: untouched rows never move
        g_acc[i] = 0.0f;
        uint16_t pp = p[i], mm = pm, vv = pv;
        nl_adam_bf16(&pp, nl_f32_to_bf16(ga), &mm, &vv, h);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// convert fp32 accumulators to the bf16 gradient the reference optimiser sees (for tests / RCCL path)
__global__ void k_emb_grad_bf16(const float* __restrict__ g_acc, uint16_t* __restrict__ g, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        g[i] = nl_f32_to_bf16(g_acc[i]);
}

__global__ void k_adam_f32(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                           int n, const NlAdamHyper* __restrict__ hp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NlAdamHyper h = *hp;
    float pp = p[i], mm = m[i], vv = v[i];
    nl_adam_f32(&pp, g[i], &mm, &vv, h);
    p[i] = pp; m[i] = mm; v[i] = vv;
}

// W2T[k][j] = W2[j][k]  (the forward GEMM streams W2^T rows as coalesced MFMA B operands)
__global__ void k_transpose_w2(const float* __restrict__ params, float* __restrict__ W2T)
{
    __shared__ float t[32][33];
    const float* W2 = params + NL_OFF_W2;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    // a rebuild from scratch: the clipped-plane flag starts again (k_prepare_w2x, the next launch, raises it); the sticky status word stays
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(W2T + NL_DEC_WS_RANGE_OFF)[NLR_PLANE_SAT] = 0u;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) t[r][threadIdx.x] = W2[(by + r) * NL_W + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) W2T[(bx + r) * NL_W + by + threadIdx.x] = t[threadIdx.x][r];
}

// bf16 operand planes of the decoder's 256-deep GEMMs (nl_decoder.hip: gemm_x9 / gemm_mask_x), rebuilt with W2T after every
// optimiser step.  A value v is split by truncation into hi + mid + lo (8 + 8 + 8 significand bits: v == hi + mid + lo
// exactly, each term a bf16).  Both matrices are stored in MFMA-fragment order so a wave fetches one B fragment with one
// contiguous 1 KB load:  [plane(3)][column tile t(8)][k-step s(16)][lane(64)][8 bf16],  lane = 32 h + c:
//     W2X  (dgrad,   dH1 = dH2 W2):    element e <-> j = 16 s + 8 h + e, k = 32 t + c, value w3_j * W2[j][k]
//     W2TX (forward, H2 = H1 W2^T):    element e <-> k = 16 s + 8 h + e, n = 32 t + c, value W2[n][k]
// The same two matrices as fp16 pairs (gemm modes 4 / 5; nl_split2_f16): [plane(2)][t(8)][s(16)][lane(64)][8 f16], the same element order,
//     W2H  (dgrad):   (w3_j * W2[j][k]) * 2^10 (NL_F16_SG),    W2TH (forward):  W2[n][k] * 2^8 (NL_F16_SW2).
// W1[k][c] * 2^8 as an fp16 pair into the two operand forms k_decoder2 reads (nl_common.h: W1F, W1X); ws16 = the weight workspace as 16-bit elements
__device__ __forceinline__ void nl_store_w1_planes(uint16_t* ws16, int k, int c, float v)
{
    uint16_t hi, lo;
    nl_split2_f16(v, NL_F16_SW1, &hi, &lo);
    uint16_t* f = ws16 + 2 * NL_DEC_WS_W1F_OFF, *x = ws16 + 2 * NL_DEC_WS_W1X_OFF;
    f[NL_W1F_INDEX(k, c, 0)] = hi; f[NL_W1F_INDEX(k, c, 1)] = lo;
    x[NL_W1X_INDEX(k, c, 0)] = hi; x[NL_W1X_INDEX(k, c, 1)] = lo;
}

__global__ void k_prepare_w2x(const float* __restrict__ params, uint16_t* __restrict__ W2X, uint16_t* __restrict__ W2TX)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;         // one thread per (tile, s, lane)
    if (t >= 8 * 16 * 64) return;
    const int lane = t & 63, s = (t >> 6) & 15, tl = t >> 10;
    const int c = 32 * tl + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
    unsigned* range = reinterpret_cast<unsigned*>(reinterpret_cast<float*>(W2X) - NL_W * NL_W + NL_DEC_WS_RANGE_OFF);
    bool clipped = false;
    for (int which = 0; which < 2; ++which) {
        uint16_t* dst = (which ? W2TX : W2X) + (size_t)t * 8;
        for (int e = 0; e < 8; ++e) {
            const int kk = k0 + e;
            const float v = which ? params[NL_OFF_W2 + c * NL_W + kk] : params[NL_OFF_W3 + kk] * params[NL_OFF_W2 + kk * NL_W + c];
            clipped |= !(fabsf(v) * (which ? NL_F16_SW2 : NL_F16_SG) <= NL_F16_MAX);
            nl_split3_bf16(v, &dst[e], &dst[e + NL_W * NL_W], &dst[e + 2 * NL_W * NL_W]);
            uint16_t* dh = W2X + NL_DEC_WS_W2H_OFF16 + (which ? 2 * NL_W * NL_W : 0) + (size_t)t * 8 + e;      // W2H | W2TH behind the bf16 planes
            nl_split2_f16(v, which ? NL_F16_SW2 : NL_F16_SG, &dh[0], &dh[NL_W * NL_W]);
        }
    }
    if (t < NL_W * NL_C) {
        const float w = params[NL_OFF_W1 + t];
        nl_store_w1_planes(reinterpret_cast<uint16_t*>(reinterpret_cast<float*>(W2X) - NL_W * NL_W), t >> 4, t & 15, w);
        clipped |= !(fabsf(w) * NL_F16_SW1 <= NL_F16_MAX);
    }
    if (clipped) atomicOr(range + NLR_PLANE_SAT, NL_SAT_PLANES);         // (negated comparisons: a NaN weight counts)
}

// poses12[f] = [R(w) row-major | t]   from pose6[f] = [t, w]     (se3pose.py:18-35)
__global__ void k_pose_matrix(const float* __restrict__ pose6, float* __restrict__ poses12, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float R[9];
    nl_rodrigues(pose6 + 6 * f + 3, R);
    for (int i = 0; i < 9; ++i) poses12[12 * f + i] = R[i];
    for (int i = 0; i < 3; ++i) poses12[12 * f + 9 + i] = pose6[6 * f + i];
}

// pose gradient tail + Adam: g_pose[f] = (dL/dt[3], dL/dR[9]) -> dL/d(t,w) -> Adam (if enabled) ->
// refreshed pose matrices.  grad6_out (optional) receives the 6-vector gradient; g_pose is cleared.
__device__ __forceinline__ void pose_step_one(int f, float* __restrict__ pose6, double* __restrict__ g_pose, float* __restrict__ m,
                                              float* __restrict__ v, const int* __restrict__ enable, float* __restrict__ grad6_out,
                                              float* __restrict__ poses12, const NlAdamHyper& h, int apply)
{
    float g6[6], gw[3], G[9];
    for (int i = 0; i < 9; ++i) G[i] = (float)g_pose[12 * f + 3 + i];          // fp64 sums (nl_field.hip), one rounding to fp32 here
    nl_rodrigues_bwd(pose6 + 6 * f + 3, G, gw);
    for (int i = 0; i < 3; ++i) { g6[i] = (float)g_pose[12 * f + i]; g6[3 + i] = gw[i]; }
    for (int i = 0; i < 12; ++i) g_pose[12 * f + i] = 0.0;
    if (grad6_out) for (int i = 0; i < 6; ++i) grad6_out[6 * f + i] = g6[i];
    if (apply && (!enable || enable[f])) {
        for (int i = 0; i < 6; ++i) nl_adam_f32(&pose6[6 * f + i], g6[i], &m[6 * f + i], &v[6 * f + i], h);
    }
    float R[9];
    nl_rodrigues(pose6 + 6 * f + 3, R);
    for (int i = 0; i < 9; ++i) poses12[12 * f + i] = R[i];
    for (int i = 0; i < 3; ++i) poses12[12 * f + 9 + i] = pose6[6 * f + i];
}

__global__ void k_pose_step(float* __restrict__ pose6, double* __restrict__ g_pose, float* __restrict__ m, float* __restrict__ v,
                            const int* __restrict__ enable, float* __restrict__ grad6_out, float* __restrict__ poses12,
                            int F, const NlAdamHyper* __restrict__ hp, int apply)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    pose_step_one(f, pose6, g_pose, m, v, enable, grad6_out, poses12, *hp, apply);
}

// ---- the whole optimiser step as ONE launch (the iteration is launch-bound at small ray counts: six ~5 us launches -> one).
// Workgroup roles by index: [embeddings | decoder rows of W2 | rest of the decoder | poses].  Every workgroup derives its group's
// bias corrections from the device step counter itself (nobody writes it while they run: a single-workgroup launch - the pose
// step of tracking - advances it at its end, a larger one leaves that to the one-thread kernel launched right behind it).  A W2-row workgroup j also owns w3_j: all its threads evaluate the update of w3_j
// from the old values (they need the NEW w3_j for the dgrad operand planes, value w3_j * W2[j][k]), one stores it after a barrier.
// Each thread then writes its updated element into W2T and into the six bf16 operand planes (layouts: k_prepare_w2x above).
struct OptimArgs {
    int* state; double lr_emb, lr_dec, lr_pose;
    uint16_t* emb; float* g_emb; uint16_t* emb_m; uint16_t* emb_v; long long n_emb; int nb_emb;
    float* params; const float* grad; float* dm; float* dv; float* W2T; uint16_t* W2X; uint16_t* W2TX; int nb_dec;
    float* pose6; double* g_pose; float* pm; float* pv; const int* enable; float* grad6_out; float* poses12; int F; int apply_pose;
    const int* counters; int skip_mode;
    int* snap_src; int* snap_dst;        // optional: the counter block is copied to snap_dst and CLEARED by the launch's last step, so that the
                                         // next iteration needs no memset launch (nl_iteration); the host reads the copy
    const int* touched_list; const int* touched_count;     // optional: the embedding group sweeps these rows instead of the whole table
    int emb_copies; long long emb_copy_stride;             // replicated accumulators (NlTouchedRows.copies): summed - and cleared - by the sweep over the touched rows
};

// "The iteration was unusable" decided ON THE DEVICE, so that the host loop needs no per-iteration read-back.  The reference skips
// the optimiser step when render_rays returns None (no ray hit a voxel: render_helpers.py:216-217; the sampler's guard
// :232-233) - mapping carries on with the next iteration (:407-410), tracking stops (:486-489).  Here: no hit ray anywhere
// (R_GLOBAL == 0), the guard flag, or a sample-buffer overflow (reported to the host at the end of the call, which raises).
// skip_mode 1 = skip this step only, 2 = sticky: once one step was skipped every later one is (tracking's `break`).
// A skipped step clears the gradient accumulators, touches no parameter or moment and does not advance the step counter;
// state[2] counts skipped steps, state[3] latches the overflow flag.
__device__ __forceinline__ bool optim_skip(const int* __restrict__ counters, const int* __restrict__ state, int skip_mode)
{
    if (!counters || skip_mode == 0) return false;
    if (counters[NLC_R_GLOBAL] == 0 || counters[NLC_GUARD] != 0 || counters[NLC_OVERFLOW] != 0) return true;
    return skip_mode == 2 && reinterpret_cast<volatile const int*>(state)[2] > 0;
}
__device__ __forceinline__ void optim_note_skip(const int* __restrict__ counters, int* __restrict__ state)
{
    state[2] = state[2] + 1;
    if (counters[NLC_OVERFLOW] != 0) state[3] = state[3] | 1;
}
// state[3] bit 1 latches the decoder's range status (nl_common.h NL_SAT_*: an operand of the fp16-pair arithmetic left - or may have left - its range during
// this call): the host reads it with the call's one status read-back and raises, like a sample overflow.  The status word itself stays set until the next
// begin_call clears it (forward-only callers read it directly).
__device__ __forceinline__ void optim_latch_range(const float* __restrict__ ws, int* __restrict__ state)
{
    if (ws && reinterpret_cast<const volatile unsigned*>(ws + NL_DEC_WS_RANGE_OFF)[NLR_STATUS]) state[3] = state[3] | 2;
}
// end of an iteration: hand the counter block to the host-visible copy and leave it zeroed for the next iteration.  Called by the first
// NL_CNT_BYTES / 4 threads of a workgroup AFTER a barrier behind the last read of the block: one word each (one thread walking the 24 words
// put ~2.5 us of dependent round trips at the end of every launch-bound iteration)
__device__ __forceinline__ void counters_hand_over(int* src, int* __restrict__ dst, int tid)
{
    if (!src || !dst || tid >= NL_CNT_BYTES / 4) return;
    dst[tid] = src[tid]; src[tid] = 0;
}
#define OPT_DEC_REST (NL_DEC_PARAMS - NL_W * NL_W - NL_W)            // W1, b1, b2, b3 (w3 rides with the W2 rows)
#define OPT_DEC_BLOCKS (NL_W + (OPT_DEC_REST + 255) / 256)

__global__ void __launch_bounds__(256) k_optim_step(OptimArgs a)
{
    __shared__ NlAdamHyper s_h;
    const int step = *reinterpret_cast<volatile const int*>(a.state) + 1;       // read once, before anybody can advance it
    const int b = blockIdx.x, tid = threadIdx.x;
    const int role = b < a.nb_emb ? 0 : (b < a.nb_emb + a.nb_dec ? 1 : 2);
    if (optim_skip(a.counters, a.state, a.skip_mode)) {             // uniform over the launch: every thread reads the same words
        if (role == 0 && a.touched_list) {
            const long long n = (long long)*a.touched_count * NL_C;
            for (long long e = (long long)b * 256 + tid; e < n; e += (long long)a.nb_emb * 256)
                for (int c = 0; c < a.emb_copies; ++c) a.g_emb[c * a.emb_copy_stride + (long long)a.touched_list[e >> 4] * NL_C + (e & 15)] = 0.0f;
        } else if (role == 0) {
            for (long long i = (long long)b * 256 + tid; i < a.n_emb; i += (long long)a.nb_emb * 256)
                if (a.g_emb[i] != 0.0f) a.g_emb[i] = 0.0f;
        } else if (role == 2) {
            const int f = (b - a.nb_emb - a.nb_dec) * 256 + tid;
            if (f < a.F) for (int i = 0; i < 12; ++i) a.g_pose[12 * f + i] = 0.0;
        }
        if (gridDim.x == 1) {
            __syncthreads();
            if (tid == 0) { optim_note_skip(a.counters, a.state); optim_latch_range(a.W2T, a.state); }
            __syncthreads();                                           // (optim_note_skip reads the overflow word)
            counters_hand_over(a.snap_src, a.snap_dst, tid);
        }
        return;
    }
    if (tid == 0) s_h = nl_adam_hyper(role == 0 ? a.lr_emb : (role == 1 ? a.lr_dec : a.lr_pose), step, 0.9, 0.999, 1e-8);
    __syncthreads();
    const NlAdamHyper h = s_h;
    if (role == 0 && a.touched_list) {
        // the rows touched since this optimiser was created (16 lanes per row); every other row has zero gradient and zero moments,
        // i.e. the dense sweep below would leave it alone: same bits, cost proportional to the touched rows instead of the table
        const long long n = (long long)*a.touched_count * NL_C;
        for (long long e = (long long)b * 256 + tid; e < n; e += (long long)a.nb_emb * 256) {
            const long long i = (long long)a.touched_list[e >> 4] * NL_C + (e & 15);
            float ga = a.g_emb[i];
            for (int c = 1; c < a.emb_copies; ++c) {             // the waves' accumulator copies, in copy order: fp32 sums, ONE bf16 rounding below
                const float gc = a.g_emb[c * a.emb_copy_stride + i];
                if (gc != 0.0f) { ga += gc; a.g_emb[c * a.emb_copy_stride + i] = 0.0f; }
            }
            const uint16_t pm = a.emb_m[i], pv = a.emb_v[i];
            if (ga == 0.0f && pm == 0 && pv == 0) continue;
            a.g_emb[i] = 0.0f;
            uint16_t pp = a.emb[i], mm = pm, vv = pv;
            nl_adam_bf16(&pp, nl_f32_to_bf16(ga), &mm, &vv, h);
            a.emb[i] = pp; a.emb_m[i] = mm; a.emb_v[i] = vv;
        }
    } else if (role == 0) {
        for (long long i = (long long)b * 256 + tid; i < a.n_emb; i += (long long)a.nb_emb * 256) {
            const float ga = a.g_emb[i];
            const uint16_t pm = a.emb_m[i], pv = a.emb_v[i];
            if (ga == 0.0f && pm == 0 && pv == 0) continue;
            a.g_emb[i] = 0.0f;
            uint16_t pp = a.emb[i], mm = pm, vv = pv;
            nl_adam_bf16(&pp, nl_f32_to_bf16(ga), &mm, &vv, h);
            a.emb[i] = pp; a.emb_m[i] = mm; a.emb_v[i] = vv;
        }
    } else if (role == 1) {
        const int r = b - a.nb_emb;
        if (r < NL_W) {
            const int j = r, k = tid, iw = NL_OFF_W3 + j, i = NL_OFF_W2 + j * NL_W + k;
            float w3 = a.params[iw], w3m = a.dm[iw], w3v = a.dv[iw];
            nl_adam_f32(&w3, a.grad[iw], &w3m, &w3v, h);
            float p = a.params[i], m = a.dm[i], v = a.dv[i];
            nl_adam_f32(&p, a.grad[i], &m, &v, h);
            a.params[i] = p; a.dm[i] = m; a.dv[i] = v;
            __syncthreads();                                         // every thread holds the old w3_j
            if (k == 0) { a.params[iw] = w3; a.dm[iw] = w3m; a.dv[iw] = w3v; }
            a.W2T[k * NL_W + j] = p;
            {   // forward planes: value W2[n = j][kk = k]
                uint16_t* d = a.W2TX + (size_t)((((j >> 5) * 16 + (k >> 4)) * 64) + 32 * ((k >> 3) & 1) + (j & 31)) * 8 + (k & 7);
                nl_split3_bf16(p, &d[0], &d[NL_W * NL_W], &d[2 * NL_W * NL_W]);
                uint16_t* dh = a.W2X + NL_DEC_WS_W2H_OFF16 + 2 * NL_W * NL_W + (d - a.W2TX);
                nl_split2_f16(p, NL_F16_SW2, &dh[0], &dh[NL_W * NL_W]);
            }
            {   // dgrad planes: value w3_j * W2[j][k] at row index j (the reduction index), column k
                uint16_t* d = a.W2X + (size_t)((((k >> 5) * 16 + (j >> 4)) * 64) + 32 * ((j >> 3) & 1) + (k & 31)) * 8 + (j & 7);
                nl_split3_bf16(w3 * p, &d[0], &d[NL_W * NL_W], &d[2 * NL_W * NL_W]);
                uint16_t* dh = a.W2X + NL_DEC_WS_W2H_OFF16 + (d - a.W2X);
                nl_split2_f16(w3 * p, NL_F16_SG, &dh[0], &dh[NL_W * NL_W]);
            }
            // range block (nl_common.h): a plane that clips raises the flag (one atomic when it happens - never, on a decoder inside the arithmetic's range)
            if (!(fabsf(p) * NL_F16_SW2 <= NL_F16_MAX) || !(fabsf(w3 * p) * NL_F16_SG <= NL_F16_MAX))
                atomicOr(reinterpret_cast<unsigned*>(a.W2T + NL_DEC_WS_RANGE_OFF) + NLR_PLANE_SAT, NL_SAT_PLANES);
        } else {
            const int g = (r - NL_W) * 256 + tid;
            int i = -1;
            if (g < NL_OFF_W2) i = g;                                 // W1, b1
            else if (g < NL_OFF_W2 + NL_W) i = NL_OFF_B2 + (g - NL_OFF_W2);
            else if (g == NL_OFF_W2 + NL_W) i = NL_OFF_B3;
            if (i >= 0) {
                float p = a.params[i], m = a.dm[i], v = a.dv[i];
                nl_adam_f32(&p, a.grad[i], &m, &v, h);
                a.params[i] = p; a.dm[i] = m; a.dv[i] = v;
                if (i < NL_OFF_B1) nl_store_w1_planes(reinterpret_cast<uint16_t*>(a.W2T), i >> 4, i & 15, p);       // W1's operand planes (k_decoder2)
            }
            if (i >= 0 && i < NL_OFF_B1 && !(fabsf(a.params[i]) * NL_F16_SW1 <= NL_F16_MAX))       // W1's planes clip
                atomicOr(reinterpret_cast<unsigned*>(a.W2T + NL_DEC_WS_RANGE_OFF) + NLR_PLANE_SAT, NL_SAT_PLANES);
        }
    } else {
        const int f = (b - a.nb_emb - a.nb_dec) * 256 + tid;
        if (f < a.F) pose_step_one(f, a.pose6, a.g_pose, a.pm, a.pv, a.enable, a.grad6_out, a.poses12, h, a.apply_pose);
    }
    if (gridDim.x == 1) {                                            // (every thread read the state / counter words before the barrier above)
        if (tid == 0) { a.state[0] = step; optim_latch_range(a.W2T, a.state); }
        counters_hand_over(a.snap_src, a.snap_dst, tid);
    }
}

// advance the step counter after a multi-workgroup k_optim_step (a last-workgroup ticket costs more than this launch: thousands
// of same-address device-scope atomics, profiles/r01_m_optimiser_step.txt)
__global__ void k_adam_advance(int* __restrict__ state, const int* counters, int skip_mode, int* snap_src,      // (snap_src IS the counter block)
                               int* __restrict__ snap_dst, const float* __restrict__ dec_ws)
{
    if (threadIdx.x == 0) {
        if (optim_skip(counters, state, skip_mode)) optim_note_skip(counters, state);
        else state[0] = state[0] + 1;
        optim_latch_range(dec_ws, state);
    }
    __syncthreads();
    counters_hand_over(snap_src, snap_dst, threadIdx.x);
}

// Multi-GPU (nerf_loam_amd/dist.py): every rank all-gathers its whole counter block (one small collective) and this kernel
// folds the gathered blocks into the local one - instead of one collective per quantity and a dozen tiny torch kernels.
//   stage 1 (after intersect): global hit-ray count, this rank's hit-rank offset, global max hits per ray
//   stage 2 (after counting):  summed loss normalisers / flags, max samples per ray, summed padded-slot constants
__global__ void k_dist_merge(const int* __restrict__ gathered, int STRIDE, int world, int rank, int stage, int* __restrict__ counters)
{
    const int t = threadIdx.x;
    if (stage == 1) {
        if (t == 0) {
            int tot = 0, off = 0, hmax = 0;
            for (int r = 0; r < world; ++r) {
                const int v = gathered[r * STRIDE + NLC_R];
                tot += v; if (r < rank) off += v;
                hmax = max(hmax, gathered[r * STRIDE + NLC_HMAX]);
            }
            counters[NLC_R_GLOBAL] = tot; counters[NLC_R_OFFSET] = off; counters[NLC_HMAX] = hmax;
        }
        return;
    }
    if (t >= NLC_NFS && t <= NLC_GUARD) {                        // NFS, NSDF, INV_* (4), OVERFLOW, GUARD: contiguous
        int sum = 0;
        for (int r = 0; r < world; ++r) sum += gathered[r * STRIDE + t];
        counters[t] = sum;
    } else if (t == NLC_SMAX) {
        int m = 0;
        for (int r = 0; r < world; ++r) m = max(m, gathered[r * STRIDE + NLC_SMAX]);
        counters[NLC_SMAX] = m;
    } else if (t == 32 || t == 33) {
        const int d = t == 32 ? NLD_INV_D2 : NLD_INV_D2CNT;
        double sum = 0.0;
        for (int r = 0; r < world; ++r) sum += reinterpret_cast<const double*>(gathered + r * STRIDE + NL_CNT_INTS)[d];
        reinterpret_cast<double*>(counters + NL_CNT_INTS)[d] = sum;
    }
}

extern "C" {

int nl_dist_merge_counters_strided(const int* gathered, int stride_ints, int world, int rank, int stage, int* counters, void* stream)
{
    if (!gathered || !counters || world <= 0 || rank < 0 || rank >= world || (stage != 1 && stage != 2) ||
        stride_ints < NL_CNT_INTS + 2 * NL_CNT_DOUBLES || (stride_ints & 1))            // (the doubles of a block stay 8-byte aligned)
        return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_dist_merge, dim3(1), dim3(64), 0, (hipStream_t)stream, gathered, stride_ints, world, rank, stage, counters);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_dist_merge_counters(const int* gathered, int world, int rank, int stage, int* counters, void* stream)
{
    return nl_dist_merge_counters_strided(gathered, NL_CNT_INTS + 2 * NL_CNT_DOUBLES, world, rank, stage, counters, stream);
}

int nl_adam_prepare(int* state, double lr_emb, double lr_dec, double lr_pose, void* stream)
{
    if (!state) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr_emb, lr_dec, lr_pose);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

static const NlAdamHyper* hyper_of(const int* state, int which) { return reinterpret_cast<const NlAdamHyper*>(state + 4) + which; }

int nl_adam_embeddings(void* emb, float* g_acc, void* m, void* v, long long n_elems, const int* state, void* stream)
{
    if (!emb || !g_acc || !m || !v || n_elems <= 0 || !state) return NL_ERR_INVALID_ARG;
    const int blocks = (int)((n_elems + 255) / 256 < 4096 ? (n_elems + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_adam_emb, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)emb, g_acc, (uint16_t*)m, (uint16_t*)v,
                       n_elems, hyper_of(state, 0));
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_embedding_grad_bf16(const float* g_acc, void* g_bf16, long long n_elems, void* stream)
{
    if (!g_acc || !g_bf16 || n_elems <= 0) return NL_ERR_INVALID_ARG;
    const int blocks = (int)((n_elems + 255) / 256 < 4096 ? (n_elems + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_emb_grad_bf16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_acc, (uint16_t*)g_bf16, n_elems);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_adam_f32(float* p, const float* g, float* m, float* v, int n, const int* state, int which, void* stream)
{
    if (!p || !g || !m || !v || n <= 0 || !state || which < 0 || which > 2) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_adam_f32, dim3(nl_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper_of(state, which));
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_dec_ws_floats(void) { return NL_DEC_WS_TOTAL; }
int nl_decoder_range_status(float* W2T, unsigned* status_out, int clear, void* stream)
{
    if (!W2T || !status_out) return NL_ERR_INVALID_ARG;
    unsigned* word = reinterpret_cast<unsigned*>(W2T + NL_DEC_WS_RANGE_OFF) + NLR_STATUS;
    if (hipMemcpyAsync(status_out, word, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return NL_ERR_LAUNCH;
    if (clear && hipMemsetAsync(word, 0, 4, (hipStream_t)stream) != hipSuccess) return NL_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return NL_ERR_LAUNCH;
    return NL_OK;
}
int nl_abi_version(void) { return NL_ABI_VERSION; }

int nl_decoder_transpose_w2(const float* params, float* W2T, void* stream)
{
    if (!params || !W2T) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_transpose_w2, dim3(NL_W / 32, NL_W / 32), dim3(32, 8), 0, (hipStream_t)stream, params, W2T);
    hipLaunchKernelGGL(k_prepare_w2x, dim3(8 * 16 * 64 / 256), dim3(256), 0, (hipStream_t)stream, params,
                       reinterpret_cast<uint16_t*>(W2T + NL_W * NL_W), reinterpret_cast<uint16_t*>(W2T + NL_W * NL_W + 3 * NL_W * NL_W / 2));
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_pose_matrices(const float* pose6, float* poses12, int F, void* stream)
{
    if (!pose6 || !poses12 || F <= 0) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_pose_matrix, dim3(nl_div_up(F, 64)), dim3(64), 0, (hipStream_t)stream, pose6, poses12, F);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_pose_step(float* pose6, double* g_pose, float* m, float* v, const int* enable, float* grad6_out, float* poses12,
                 int F, const int* state, int apply, void* stream)
{
    if (!pose6 || !g_pose || !m || !v || !poses12 || F <= 0 || !state) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_pose_step, dim3(nl_div_up(F, 64)), dim3(64), 0, (hipStream_t)stream, pose6, g_pose, m, v, enable, grad6_out,
                       poses12, F, hyper_of(state, 2), apply);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

/* nl_optimiser_step + the end-of-iteration hand-over of the counter block: its words are copied to counters_copy and CLEARED by
 * the launch's last step (nl_iteration: the next iteration then starts without a memset launch; the host reads the copy) */
int nl_optimiser_step_t(int* state, double lr_emb, double lr_dec, double lr_pose,
                        void* emb, float* g_emb, void* emb_m, void* emb_v, long long n_emb,
                        float* dec_params, const float* dec_grad, float* dec_m, float* dec_v, float* dec_ws,
                        float* pose6, double* g_pose, float* pose_m, float* pose_v, const int* pose_enable, float* grad6_out,
                        float* poses12, int F, int apply_pose, const int* counters, int skip_mode, int* counters_rw, int* counters_copy,
                        const NlTouchedRows* touched, void* stream)
{
    if (!state || skip_mode < 0 || skip_mode > 2 || (skip_mode && !counters)) return NL_ERR_INVALID_ARG;
    if (touched && (touched->struct_size != (int)sizeof(NlTouchedRows) || (touched->list && !touched->count))) return NL_ERR_INVALID_ARG;
    if (emb && (!g_emb || !emb_m || !emb_v || n_emb <= 0)) return NL_ERR_INVALID_ARG;
    if (dec_params && (!dec_grad || !dec_m || !dec_v || !dec_ws)) return NL_ERR_INVALID_ARG;
    if (pose6 && (!g_pose || !pose_m || !pose_v || !poses12 || F <= 0)) return NL_ERR_INVALID_ARG;
    if (!emb && !dec_params && !pose6) return NL_ERR_INVALID_ARG;
    OptimArgs a;
    a.state = state; a.lr_emb = lr_emb; a.lr_dec = lr_dec; a.lr_pose = lr_pose;
    a.emb = (uint16_t*)emb; a.g_emb = g_emb; a.emb_m = (uint16_t*)emb_m; a.emb_v = (uint16_t*)emb_v; a.n_emb = emb ? n_emb : 0;
    a.nb_emb = emb ? (int)((n_emb + 255) / 256 < 4096 ? (n_emb + 255) / 256 : 4096) : 0;
    a.touched_list = (emb && touched) ? touched->list : nullptr; a.touched_count = (emb && touched) ? touched->count : nullptr;
    a.emb_copies = (a.touched_list && touched->copies > 1) ? touched->copies : 1;
    a.emb_copy_stride = a.emb_copies > 1 ? touched->copy_stride : 0;
    if (a.emb_copies > 1 && a.emb_copy_stride <= 0) return NL_ERR_INVALID_ARG;
    if (a.touched_list && a.nb_emb > 1024) a.nb_emb = 1024;        // the row count is on the device: a fixed grid strides over it
    a.params = dec_params; a.grad = dec_grad; a.dm = dec_m; a.dv = dec_v; a.W2T = dec_ws;
    a.W2X = dec_ws ? reinterpret_cast<uint16_t*>(dec_ws + NL_W * NL_W) : nullptr;
    a.W2TX = dec_ws ? reinterpret_cast<uint16_t*>(dec_ws + NL_W * NL_W + 3 * NL_W * NL_W / 2) : nullptr;
    a.nb_dec = dec_params ? OPT_DEC_BLOCKS : 0;
    a.pose6 = pose6; a.g_pose = g_pose; a.pm = pose_m; a.pv = pose_v; a.enable = pose_enable; a.grad6_out = grad6_out;
    a.poses12 = poses12; a.F = pose6 ? F : 0; a.apply_pose = apply_pose;
    a.counters = counters; a.skip_mode = skip_mode;
    a.snap_src = counters_copy ? counters_rw : nullptr; a.snap_dst = counters_rw ? counters_copy : nullptr;
    const int nb_pose = pose6 ? nl_div_up(F, 256) : 0;
    const int nb = a.nb_emb + a.nb_dec + nb_pose;
    hipLaunchKernelGGL(k_optim_step, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    if (nb > 1) hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, state, counters, skip_mode, a.snap_src, a.snap_dst, (const float*)dec_ws);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_optimiser_step_ex(int* state, double lr_emb, double lr_dec, double lr_pose,
                         void* emb, float* g_emb, void* emb_m, void* emb_v, long long n_emb,
                         float* dec_params, const float* dec_grad, float* dec_m, float* dec_v, float* dec_ws,
                         float* pose6, double* g_pose, float* pose_m, float* pose_v, const int* pose_enable, float* grad6_out,
                         float* poses12, int F, int apply_pose, const int* counters, int skip_mode, int* counters_rw, int* counters_copy,
                         void* stream)
{
    return nl_optimiser_step_t(state, lr_emb, lr_dec, lr_pose, emb, g_emb, emb_m, emb_v, n_emb, dec_params, dec_grad, dec_m, dec_v, dec_ws, pose6,
                               g_pose, pose_m, pose_v, pose_enable, grad6_out, poses12, F, apply_pose, counters, skip_mode, counters_rw, counters_copy,
                               nullptr, stream);
}

// begin of an optimisation call: the rows the PREVIOUS call touched get their gradient accumulators, moments and flag words
// cleared (the reference constructs a fresh torch.optim.Adam per call, render_helpers.py:353), then the list is emptied - cost
// proportional to the touched rows, not to the table
__global__ void k_touched_reset(const int* __restrict__ list, const int* __restrict__ count, unsigned* __restrict__ flags,
                                float* __restrict__ g_emb, uint16_t* __restrict__ emb_m, uint16_t* __restrict__ emb_v, int copies, long long copy_stride)
{
    const long long n = (long long)*count * NL_C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int row = list[e >> 4];
        const long long i = (long long)row * NL_C + (e & 15);
        g_emb[i] = 0.0f; emb_m[i] = 0; emb_v[i] = 0;
        for (int c = 1; c < copies; ++c) g_emb[c * copy_stride + i] = 0.0f;
        if ((e & 15) == 0) flags[row >> 5] = 0u;                     // every bit of a word belongs to a listed row: all writers store 0
    }
}
__global__ void k_touched_count_clear(int* count) { *count = 0; }

int nl_touched_rows_reset(const NlTouchedRows* touched, float* g_emb, void* emb_m, void* emb_v, void* stream)
{
    if (!touched || touched->struct_size != (int)sizeof(NlTouchedRows) || !touched->list || !touched->count || !touched->flags || !g_emb || !emb_m || !emb_v) return NL_ERR_INVALID_ARG;
    if (touched->copies > 1 && touched->copy_stride <= 0) return NL_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_touched_reset, dim3(1024), dim3(256), 0, (hipStream_t)stream, touched->list, touched->count, touched->flags, g_emb,
                       (uint16_t*)emb_m, (uint16_t*)emb_v, touched->copies > 1 ? touched->copies : 1, touched->copy_stride);
    hipLaunchKernelGGL(k_touched_count_clear, dim3(1), dim3(1), 0, (hipStream_t)stream, touched->count);
    NL_LAUNCH_CHECK();
    return NL_OK;
}

int nl_optimiser_step(int* state, double lr_emb, double lr_dec, double lr_pose,
                      void* emb, float* g_emb, void* emb_m, void* emb_v, long long n_emb,
                      float* dec_params, const float* dec_grad, float* dec_m, float* dec_v, float* dec_ws,
                      float* pose6, double* g_pose, float* pose_m, float* pose_v, const int* pose_enable, float* grad6_out,
                      float* poses12, int F, int apply_pose, const int* counters, int skip_mode, void* stream)
{
    return nl_optimiser_step_ex(state, lr_emb, lr_dec, lr_pose, emb, g_emb, emb_m, emb_v, n_emb, dec_params, dec_grad, dec_m, dec_v, dec_ws, pose6,
                                g_pose, pose_m, pose_v, pose_enable, grad6_out, poses12, F, apply_pose, counters, skip_mode, nullptr, nullptr, stream);
}

int nl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int nl_version(void) { return 100; }

}  // extern "C"
