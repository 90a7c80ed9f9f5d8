"""bundle_adjust_frames / track_frame / render_rays on the MI355X SdfEngine.

Host-side mirror of /root/reference/src/variations/render_helpers.py: same function names, argument
order and meaning, in-place mutation of the embedding parameter, decoder module and frame poses,
`None` to signal "skip this iteration" (render_helpers.py:216-217,232-233, B9), the tracker's
learning-rate rule (render_helpers.py:449-450).  What differs is everything underneath: one call
into SdfEngine per iteration (a fixed sequence of HIP kernels, device-resident state) instead of
~100 torch ops, autograd and a torch.optim.Adam object.

The loops run WITHOUT a host synchronisation: the ray subset of every iteration is drawn on the device
(RAY_SELECTION), the sampler jitter is re-drawn per iteration from the device-side step counter, an
unusable iteration is recognised and skipped by the optimiser kernel itself, the decoder block stays
resident between calls, and one small read-back at the end of the call reports what happened.
Dropped (no observable effect): torch.cuda.empty_cache() calls, the unused autograd.grad pass of
render_helpers.py:293-297 (B10), decoder .grad accumulation during tracking (B11)."""
from collections import OrderedDict
from copy import deepcopy
import os

import numpy as np
import torch

from . import _lib as L
from . import ops
from .lidar_frame import scan_of
from .pipeline import BYTES_PER_SAMPLE, DecoderDevice, IterConfig, MapDevice, SdfEngine, samples_per_ray_bound

_ENGINES = OrderedDict()
MAX_CACHED_ENGINES = 4          # (device, ray capacity, frame capacity) -> SdfEngine, least recently used evicted

# "device" (default): nl_select_rays draws every iteration's ray subset on the GPU from the resident scan - the reference's
# distribution (a uniformly random N-subset, dataset order kept; lidarFrame.py:55-57), no host RNG, no top-k on the CPU, no H2D
# copy.  "host": the reference's own Gumbel top-k on the global CPU torch generator (LidarFrame.sample_rays) + upload: a seeded
# run then picks exactly the rays the reference's CPU path would pick (the sampler jitter comes from a private generator and
# does not advance the global one).
RAY_SELECTION = os.environ.get("NL_RAY_SELECTION", "device")

# Sampler jitter (the reference draws torch `uniform_()` noise per iteration, voxel_helpers.py:298-303).  None (default): a seed
# from the private generator per call, re-drawn on the device every iteration.  (seed, fresh): reproducible runs and the parity
# tests - `seed` keys the counter-based noise hash(seed, ray index, step) the oracle and tests/golden/make_golden.py use;
# fresh=False keeps the jitter a function of (seed, ray, step) only, i.e. the same in every iteration, like the goldens.
SAMPLER_NOISE = None

_PRIVATE_GEN = None
_PRIVATE_SEED = None


def _draw_seed():
    """seeds of the device-side random streams (ray subsets, sampler jitter) come from a PRIVATE generator derived from torch's global
    seed: reproducible under torch.manual_seed - a new manual_seed starts the private stream again, whatever was drawn before - and the
    global CPU stream the reference's sample_rays consumes is left untouched"""
    global _PRIVATE_GEN, _PRIVATE_SEED
    if _PRIVATE_GEN is None or _PRIVATE_SEED != torch.initial_seed():
        _PRIVATE_SEED = torch.initial_seed()
        _PRIVATE_GEN = torch.Generator()
        _PRIVATE_GEN.manual_seed((_PRIVATE_SEED * 0x9E3779B1 + 0x7F4A7C15) & 0x7FFFFFFFFFFFFFFF)
    return int(torch.randint(0, 2 ** 31 - 1, (1,), generator=_PRIVATE_GEN).item())


def reseed():
    """start the private seed stream of _draw_seed again from torch's current global seed (a torch.manual_seed with a NEW value does
    that by itself; re-seeding with the same value cannot be told from not seeding): for runs that must not depend on what the
    process drew before"""
    global _PRIVATE_GEN
    _PRIVATE_GEN = None


# Replicated embedding-gradient accumulators of the API's engines (SdfEngine(emb_grad_copies=), include/nerfloam_hip.h NlTouchedRows.copies).  Default 1 = a single
# array: on the 150-scan map 16 copies take 45 us off the scatter of a 4096 x 4 bundle adjustment (the near-sensor rows' same-address atomics) but the optimiser's
# fold over the copies costs 64 us (bench.py large_map.*_16_accumulator_copies, profiles/experiments/README.md round 5).
EMB_GRAD_COPIES = int(os.environ.get("NL_EMB_GRAD_COPIES", "1"))
SAMPLE_MEMORY_FRACTION = 0.5     # of the device memory that is free when an engine is built: the most its per-sample workspace may take


def _samples_per_ray_cap(n_rays, voxel_size, step_size, device):
    """-> (samples per ray the engine provides for, True when device memory clipped it).  The capacity is DERIVED from the call's
    settings (pipeline.samples_per_ray_bound: the most samples the reference's sampler can give one ray at this voxel / step size),
    so no legal configuration overflows it - the tracker steps of the shipped configs (tracking.py:36: 0.2 x voxel, ncd 0.1 x voxel)
    on a map of any density included.  Only the device's memory limits it: 184 B per sample, i.e. 2048 rays x 368 samples = 139 MB."""
    need = samples_per_ray_bound(voxel_size, step_size)
    try:
        free = torch.cuda.mem_get_info(device)[0]
    except Exception:                                       # noqa: BLE001 - no memory query: trust the bound
        return need, False
    fit = int(free * SAMPLE_MEMORY_FRACTION / (BYTES_PER_SAMPLE * max(int(n_rays), 1)))
    return (need, False) if fit >= need else (max(fit, 1), True)


def _engine(n_rays, n_frames, device, voxel_size, step_size):
    key = (str(device), int(n_rays), max(2, int(n_frames)))
    eng = _ENGINES.get(key)
    need = samples_per_ray_bound(voxel_size, step_size)
    if eng is None or (eng.samples_per_ray_cap < need and not eng.samples_clipped):
        # (a cached engine built for a coarser step - the mapper's - is replaced by one that also holds the tracker's samples)
        if eng is not None:
            del _ENGINES[key], eng
        cap, clipped = _samples_per_ray_cap(key[1], voxel_size, step_size, device)
        eng = SdfEngine(max_rays=key[1], samples_per_ray_cap=cap, max_frames=key[2], device=device, emb_grad_copies=EMB_GRAD_COPIES)
        eng.samples_clipped = clipped
        _ENGINES[key] = eng
        while len(_ENGINES) > MAX_CACHED_ENGINES:
            _ENGINES.popitem(last=False)
    else:
        _ENGINES.move_to_end(key)
    return eng


def _map_device(map_states, voxel_size, device):
    md = map_states.get("_device")
    emb = map_states["voxel_vertex_emb"]
    if md is None or md.emb.data_ptr() != emb.data_ptr() or md.n_nodes != map_states["voxel_center_xyz"].shape[0]:
        md = MapDevice.from_tensors(map_states["voxel_center_xyz"], map_states["voxel_structure"], map_states["voxel_vertex_idx"],
                                    map_states["voxel_id2embedding_id"], emb, voxel_size, device)
        map_states["_device"] = md
    return md


def _param_list(sdf_network):
    """the six parameter tensors of the shipped decoder configuration (W1 b1 W2 b2 W3 b3) of ANY module laid out like the reference's
    variations/lidar.py:105-107 (`pts_linears[0..1]`, `sdf_out`): nerf_loam_amd.decoder.Decoder and the reference's own Decoder alike"""
    if hasattr(sdf_network, "param_list"):
        return sdf_network.param_list()
    lin = list(getattr(sdf_network, "pts_linears", []))
    out = getattr(sdf_network, "sdf_out", None)
    shapes = [tuple(l.weight.shape) for l in lin] + ([tuple(out.weight.shape)] if out is not None else [])
    if shapes != [(L.NL_W, L.NL_C), (L.NL_W, L.NL_W), (1, L.NL_W)] or getattr(sdf_network, "skips", []) not in ([], ()) or \
            type(getattr(sdf_network, "pe", None)).__name__ not in ("NoneType", "Same"):
        raise NotImplementedError("the MI355X decoder kernels implement the reference's shipped configuration only: depth 2, width 256, "
                                  f"in_dim 16, skips [], embedder none (got layers {shapes})")
    return [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, out.weight, out.bias]


def _decoder_version(sdf_network):
    return tuple((p.data_ptr(), p._version) for p in _param_list(sdf_network))


def _decoder_device(sdf_network, device):
    """device-resident parameter block + operand planes of the decoder module, kept between calls: rebuilt only when somebody
    else wrote the module's parameters (tensor version counters) - no per-call D2H / H2D round trip.  A module that already lives on
    the device has its parameters turned into VIEWS of the block: the optimiser kernels then update the module in place and a call
    ends without the six write-back copies (an outside write - load_state_dict, an optimiser of the caller's - goes through the same
    views, bumps the version counters and the block is rebuilt with its planes at the next call)."""
    cached = getattr(sdf_network, "_nl_device", None)
    ver = _decoder_version(sdf_network)
    if cached is not None and cached[0] == ver and cached[1].params.device == torch.device(device):
        return cached[1]
    plist = _param_list(sdf_network)
    p = torch.cat([q.detach().reshape(-1).float() for q in plist]).to(device).contiguous()
    dec = DecoderDevice.from_flat(p)
    if all(q.device == dec.params.device and q.dtype == torch.float32 for q in plist):
        with torch.no_grad():
            off = 0
            for q in plist:
                n = q.numel()
                q.data = dec.params[off:off + n].view(q.shape)
                off += n
    sdf_network._nl_device = (_decoder_version(sdf_network), dec)
    return dec


def _decoder_aliased(sdf_network, dec):
    off, base = 0, dec.params.data_ptr()
    for q in _param_list(sdf_network):
        if q.data_ptr() != base + 4 * off:
            return False
        off += q.numel()
    return True


def _decoder_writeback(sdf_network, dec):
    if not _decoder_aliased(sdf_network, dec):               # (a module on another device / dtype: copy the block back)
        with torch.no_grad():
            off = 0
            for q in _param_list(sdf_network):
                n = q.numel()
                q.copy_(dec.params[off:off + n].view_as(q))
                off += n
    sdf_network._nl_device = (_decoder_version(sdf_network), dec)


def _cfg(loss_criteria, voxel_size, step_size, max_distance, lrs=(0.0, 0.0, 0.0)):
    return IterConfig(voxel_size=float(voxel_size), step_size=float(step_size), max_distance=float(max_distance),
                      truncation=float(loss_criteria.truncation), sdf_weight=float(loss_criteria.sdf_weight),
                      fs_weight=float(loss_criteria.fs_weight), lr_emb=lrs[0], lr_dec=lrs[1], lr_pose=lrs[2],
                      noise_seed=_draw_seed() if SAMPLER_NOISE is None else int(SAMPLER_NOISE[0]))


def _fresh_noise():
    return True if SAMPLER_NOISE is None else bool(SAMPLER_NOISE[1])


def _gather_rays(eng, frames, N_rays, track=False):
    """host ray selection: every frame draws its boolean `sample_mask` the reference's way (frame.sample_rays, lidarFrame.py:55-57 - torch's
    global CPU generator, or whatever a test patched in); the M-byte mask is all that is uploaded - points / cos are gathered from the
    resident scan and the unit directions of the selected returns come from nl_unit_dirs (lidarFrame.py:47-52 on the device)."""
    d, p, c, f = [], [], [], []
    for i, fr in enumerate(frames):
        fr.sample_rays(N_rays, track=track) if track else fr.sample_rays(N_rays)
        sc = scan_of(fr, eng.dev)
        idx = fr.sample_mask.reshape(-1).to(eng.dev).nonzero().squeeze(1)
        pts = sc["points"][idx]
        if sc["dirs"] is None:
            dirs = torch.empty_like(pts)
            ops.unit_dirs(pts, dirs)
        else:
            dirs = sc["dirs"][idx]
        d.append(dirs); p.append(pts); c.append(sc["cos"][idx])
        f.append(torch.full((idx.numel(),), i, dtype=torch.int32, device=eng.dev))
    return torch.cat(d), torch.cat(p), torch.cat(c), torch.cat(f)


def _ray_plan(eng, frames, N_rays, seeds, track=False):
    """-> (use(it), done()): use makes iteration `it` of the call run on its ray subset, done() leaves every frame's boolean `sample_mask`
    (the reference's attribute, lidarFrame.py:55-57) = the LAST iteration's selection in a buffer the frame owns.  Frames are anything with
    .points / .pointsCos (scan_of: resident once, directions derived in the selection kernel).  Device selection: the subsets of ALL
    iterations are drawn up front (SdfEngine.predraw: two launches per eight (iteration, frame) pairs), an iteration then only points the
    descriptor at its slice.  Shapes outside the window method's range and host selection: one draw per iteration (_ray_drawer)."""
    if RAY_SELECTION == "device":
        scans = [scan_of(fr, eng.dev) for fr in frames]
        if eng.predraw(scans, N_rays, seeds):
            def done():
                # the predrawn masks live in an engine-wide buffer the next call overwrites: a frame keeps its own copy (M bytes, D2D)
                for f, (fr, sc) in enumerate(zip(frames, scans)):
                    sc["mask_u8"].copy_(eng._pre["masks"][f][len(seeds) - 1])
                    fr.sample_mask = sc["mask_u8"].view(torch.bool).view(-1, 1)
            return eng.use_predrawn, done
    draw = _ray_drawer(eng, frames, N_rays, track=track)
    return (lambda it: draw(seeds[it])), (lambda: None)


def _ray_drawer(eng, frames, N_rays, track=False):
    """-> draw(seed): puts the iteration's ray subset of every frame into the engine.  Device selection: the frame list is
    marshalled once per call, every draw is one C call (two launches for all frames); the frames' boolean `sample_mask`
    attribute (the reference's, lidarFrame.py:55-57) is a view of the frame-owned buffer the selection kernel writes."""
    if RAY_SELECTION != "device":
        return lambda seed: eng.set_rays(*_gather_rays(eng, frames, N_rays, track=track))
    scans = [scan_of(fr, eng.dev) for fr in frames]
    for fr, sc in zip(frames, scans):
        fr.sample_mask = sc["mask_u8"].view(torch.bool).view(-1, 1)
    if eng.prepare_selection(scans, N_rays):
        # reselect() returns False WITHOUT launching when the shapes leave the window method's range (n >= M, or a candidate
        # window beyond its capacity: ~17 k rays per frame at M = 131 k): the per-frame radix path handles every shape
        return lambda seed: eng.reselect(seed) or eng.select_rays(scans, N_rays, seed)
    return lambda seed: eng.select_rays(scans, N_rays, seed)


def _finish_call(eng, what):
    """the ONE host read-back of a call (the iterations themselves never synchronise): how many optimiser steps the device
    skipped as unusable, whether a sample buffer overflowed (not the reference's "returns None" case: fail loudly), and the
    frames' poses"""
    (steps, skipped, overflow), poses = eng.call_status_and_poses()
    if overflow:
        # the sample workspace holds the sampler's worst case for the call's voxel / step size (_samples_per_ray_cap), so this is reachable
        # only when device memory clipped that capacity, or when the on-device ray selection missed its threshold window
        raise L.NerfLoamHipError(f"the call is invalid: an iteration produced more than {eng.P_cap} valid samples ({eng.P_cap // max(eng.N_cap, 1)} "
                                 "per ray are provided for" + (": device memory clipped the sample workspace below the sampler's worst case for this "
                                 "step_size" if getattr(eng, "samples_clipped", False) else "") + "), or the on-device ray selection missed its "
                                 "threshold window")
    if eng.saturated and os.environ.get("NL_ON_SATURATION", "raise") != "ignore":
        # the reference's decoder is unbounded fp32 (lidar.py:109-123); the default arithmetic here clips - and a clipped operand trains on a wrong gradient
        raise L.NerfLoamHipError("the call is invalid: an operand of the decoder left the range of the fp16-pair arithmetic (|X| < 1023 - 256 with a trainable decoder -, "
                                 "|W1|, |W2| < 256, H1 < 4094, |w3_j W2[j][k]| < 64, dgrad sums < 64; DecoderDevice.range_status() names which, two of the four as "
                                 "conservative bounds) and was clipped.  Run this map with the exact-product arithmetic: SdfEngine(gemm_mode=3) / NL_GEMM_MODE=3 "
                                 "(NL_ON_SATURATION=ignore keeps the clipped results)")
    for _ in range(skipped if what == "Mapping" else min(skipped, 1)):
        print(f"Encouter a bug while {what}, currently not be fixed, " + ("Continue!!" if what == "Mapping" else "Restarting!!"))
    return steps, skipped, poses


def bundle_adjust_frames(keyframe_graph, embeddings, map_states, sdf_network, loss_criteria, voxel_size, step_size,
                         N_rays=512, num_iterations=10, truncation=0.1, max_voxel_hit=10, max_distance=10,
                         learning_rate=[1e-2, 1e-2, 5e-3], update_pose=True, update_decoder=True, profiler=None):
    device = embeddings.device
    if profiler is not None:
        profiler.tick("mapping_add_optim")
    assert map_states["voxel_vertex_emb"].data_ptr() == embeddings.data_ptr(), "embeddings must be map_states['voxel_vertex_emb']"
    m = _map_device(map_states, voxel_size, device)
    dec = _decoder_device(sdf_network, device)
    eng = _engine(N_rays * len(keyframe_graph), len(keyframe_graph), device, voxel_size, step_size)
    cfg = _cfg(loss_criteria, voxel_size, step_size, max_distance, learning_rate)
    optimise = [int(kf.index != 0 and update_pose) for kf in keyframe_graph]
    eng.set_poses(np.stack([kf.pose.data.detach().cpu().numpy() for kf in keyframe_graph]), optimise)
    eng.begin_call(m, dec)                                # fresh Adam per call (render_helpers.py:353)
    if profiler is not None:
        profiler.tok("mapping_add_optim")
    seed0 = _draw_seed()
    use, masks_done = _ray_plan(eng, keyframe_graph, N_rays, [seed0 + it for it in range(num_iterations)])
    # one C call per iteration (nl_iteration: ~15 launches); no host synchronisation inside the loop: an unusable iteration is
    # recognised and skipped by the optimiser kernel itself (skip_mode), fresh sampler jitter comes from the device step counter
    eng.bind(m, dec, cfg, train_decoder=update_decoder, want_emb_grad=True, want_pose_grad=any(optimise), update_emb=True,
             update_decoder=update_decoder, update_pose=any(optimise), skip_mode=1, fresh_noise=_fresh_noise())
    for it in range(num_iterations):
        use(it)
        eng.run_bound()
    masks_done()
    _, _, p6 = _finish_call(eng, "Mapping")
    with torch.no_grad():
        if update_decoder:
            _decoder_writeback(sdf_network, dec)
        for i, kf in enumerate(keyframe_graph):
            if optimise[i]:
                kf.pose.data.copy_(p6[i].to(kf.pose.data.device))


def track_frame(frame_pose, curr_frame, map_states, sdf_network, loss_criteria, voxel_size, N_rays=512, step_size=0.05,
                num_iterations=10, truncation=0.1, learning_rate=1e-3, max_voxel_hit=10, max_distance=10, profiler=None,
                depth_variance=False):
    emb = map_states["voxel_vertex_emb"]
    device = emb.device
    m = _map_device(map_states, voxel_size, device)
    dec = _decoder_device(sdf_network, device)
    eng = _engine(N_rays, 1, device, voxel_size, step_size)
    cfg = _cfg(loss_criteria, voxel_size, step_size, max_distance)
    # (the reference deep-copies the module, render_helpers.py:445; a module holding one 6-vector is rebuilt directly - 0.1 ms of the call)
    init_pose = type(frame_pose)(frame_pose.data.detach().clone()) if type(frame_pose).__name__ == "OptimizablePose" else deepcopy(frame_pose)
    lr = learning_rate * 2 if curr_frame.index < 2 else learning_rate / 3
    eng.set_poses(init_pose.data.detach().cpu().numpy()[None], [1])
    eng.begin_call(m, None, emb_state=False)
    seed0 = _draw_seed()
    use, masks_done = _ray_plan(eng, [curr_frame], N_rays, [seed0 + it for it in range(num_iterations)], track=True)
    eng.bind(m, dec, cfg, train_decoder=False, want_emb_grad=False, want_pose_grad=True, update_emb=False, update_decoder=False,
             update_pose=True, lr_pose=lr, skip_mode=2, fresh_noise=_fresh_noise())       # sticky skip = the reference's `break`
    for it in range(num_iterations):
        use(it)
        eng.run_bound()
    masks_done()
    _, skipped, p6 = _finish_call(eng, "Tracking")
    hit_mask = None if skipped else (eng.hit_count[:eng.N] > 0)
    with torch.no_grad():
        init_pose.data.copy_(p6[0].to(init_pose.data.device))
    return init_pose, hit_mask


@torch.no_grad()
def render_rays(rays_o, rays_d, map_states, sdf_network, step_size, voxel_size, truncation, max_voxel_hit, max_distance,
                chunk_size=10000, profiler=None, return_raw=False):
    """Forward rendering of world-space rays (render_helpers.py:190-318): any origins - every distinct origin becomes one "frame" of
    the engine (identity rotation, that translation), so a batch of several frames' rays (bundle_adjust_frames' layout, :385-388) or
    rays with individual origins render in one pass - up to NL_MAX_FRAMES = 32 distinct origins per call (the field kernels keep the
    frames' poses in registers); more raise here, render them in groups.  The returned tensors stay on the device."""
    emb = map_states["voxel_vertex_emb"]
    device = emb.device
    o = rays_o.reshape(-1, 3).to(device, torch.float32)
    d = rays_d.reshape(-1, 3).to(device, torch.float32).contiguous()
    if o.shape[0] != d.shape[0]:
        o = o.expand(d.shape[0], 3)
    if bool((o == o[:1]).all()):
        origins, fid = o[:1], None
    else:
        origins, inv = torch.unique(o, dim=0, return_inverse=True)
        fid = inv.to(torch.int32)
        if origins.shape[0] > L.NL_MAX_FRAMES:
            raise L.NerfLoamHipError(f"render_rays: {origins.shape[0]} distinct ray origins in one call, the field kernels take at most "
                                     f"{L.NL_MAX_FRAMES} frames (csrc/nl_field.hip NL_MAX_FRAMES): render the rays in groups of origins")
    m = _map_device(map_states, voxel_size, device)
    dec = _decoder_device(sdf_network, device)
    eng = _engine(d.shape[0], origins.shape[0], device, voxel_size, step_size)
    crit = type("C", (), dict(truncation=truncation, sdf_weight=1.0, fs_weight=1.0))
    cfg = _cfg(crit, voxel_size, step_size, max_distance)
    pose6 = torch.cat([origins, torch.zeros_like(origins)], 1).cpu().numpy()
    eng.set_poses(pose6, [0] * origins.shape[0])
    eng.set_rays(d, torch.zeros_like(d), torch.ones(d.shape[0], device=device), fid)
    eng.forward_only(m, dec, cfg)
    r = eng.export_render_device()
    if r is None:
        return None
    P = r["stats"]["P"]
    return {"z_vals": r["z_vals"], "sdf": r["sdf"], "ray_mask": r["ray_mask"].view(1, -1), "valid_mask": r["valid_mask"],
            "sampled_xyz": m.centres[eng.s_vox[:P].long()], "_engine": eng, "_cfg": cfg}


SCORES_CHUNK_POINTS = 1 << 24          # points per gather + decoder launch pair of get_scores (16.8 M points = 1 GB of X rows; the reference's chunk is 10 000 voxels)


@torch.no_grad()
def get_scores(sdf_network, map_states, voxel_size, bits=8, device_out=False):
    """Dense SDF grid of every voxel passed in map_states (mesh extraction, reference render_helpers.py:96-153):
    res^3 points on linspace(-0.5, 0.5, res)^3 * voxel_size around each voxel centre -> [n_voxels, res, res, res, 1]
    on the host, like the reference returns (device_out=True: left on the device).  The points are generated inside the gather
    kernel (nl_gather_grid) from torch's own linspace values - no xyz / voxel-id tensors -, a chunk is up to 16.8 M points = one gather +
    one matrix-core forward launch, the grid stays on the device and leaves in ONE copy (the reference: a .cpu() per 10 000 voxels)."""
    from . import ops
    emb = map_states["voxel_vertex_emb"]
    device = emb.device
    m = MapDevice.from_tensors(map_states["voxel_center_xyz"], map_states["voxel_structure"], map_states["voxel_vertex_idx"],
                               map_states["voxel_id2embedding_id"], emb, voxel_size, device, traversal=False)
    dec = _decoder_device(sdf_network, device)
    res = int(bits)
    lin = torch.linspace(-0.5, 0.5, res).to(device)
    n = m.centres.shape[0]
    r3 = res ** 3
    sdf = torch.empty(n * r3, dtype=torch.float32, device=device)
    chunk = max(1, SCORES_CHUNK_POINTS // r3)
    X = torch.empty(min(n, chunk) * r3, L.NL_C, dtype=torch.float32, device=device)
    hint = L.lib().nl_decoder_grid_hint()
    dec.W2T[L.NL_DEC_WS_RANGE_STATUS:L.NL_DEC_WS_RANGE_STATUS + 1].zero_()            # (the range status of THIS call)
    for i in range(0, n, chunk):
        k = min(chunk, n - i)
        ops.gather_grid(k, i, res, lin, m.centres, m.vertex_rows, m.emb, voxel_size, X)
        ops.decoder_forward(X, dec.params, dec.W2T, k * r3, sdf[i * r3:], hint)
    if dec.range_status() and os.environ.get("NL_ON_SATURATION", "raise") != "ignore":
        raise L.NerfLoamHipError("get_scores: an operand of the decoder left the range of the fp16-pair arithmetic and was clipped (DecoderDevice.range_status()); "
                                 "use NL_GEMM_MODE=3")
    out = sdf.view(-1, res, res, res, 1)
    return out if device_out else out.cpu()
