"""GPU (-m gpu): the reference's operator surface (Mapping / Tracking / bundle_adjust_frames /
track_frame / render_rays / Decoder / Criterion / svo.Octree) on the HIP path: a short
mapping-then-tracking run on a synthetic sector scan behaves like the reference's loop does -
loss goes down, embeddings/decoder/pose are mutated in place, a perturbed pose is pulled back."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _reseed():
    from nerf_loam_amd import render_helpers as RH
    RH.reseed()


def make_args():
    return Namespace(
        criteria=dict(sdf_weight=10000.0, fs_weight=1, eiko_weight=0.1, sdf_truncation=0.30),
        data_specs=dict(max_depth=50.0, min_depth=1.5),
        decoder_specs=dict(depth=2, width=256, in_dim=16, skips=[], embedder="none", multires=0),
        tracker_specs=dict(N_rays=1024, learning_rate=0.005, step_size=0.2, max_voxel_hit=20, num_iterations=10),
        mapper_specs=dict(N_rays_each=1024, use_local_coord=False, voxel_size=0.2, step_size=0.5, window_size=4, num_iterations=10,
                          max_voxel_hit=20, final_iter=True, mesh_res=2, learning_rate_emb=0.03, learning_rate_decorder=0.005,
                          learning_rate_pose=0.001, freeze_frame=5, keyframe_gap=8, remove_back=False, key_distance=12),
        debug_args=dict(verbose=False, mesh_freq=100))


def test_mapping_then_tracking_like_the_reference_loop():
    from nerf_loam_amd.lidar_frame import LidarFrame
    from nerf_loam_amd.mapping import Mapping
    from nerf_loam_amd.tracking import Tracking
    from nerf_loam_amd.render_helpers import render_rays
    torch.manual_seed(777)
    _reseed()                                                 # the device-side seed stream from its start: the outcome does not depend on test order
    pts, cos = H.scene_points(64, 64, 11)
    args = make_args()
    mapper = Mapping(args)
    f0 = LidarFrame(0, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
    mapper.create_voxels(f0)
    assert mapper.svo.count_leaf_nodes() > 500 and mapper.dynamic_embeddings.dtype == torch.bfloat16
    assert mapper.dynamic_embeddings.is_cuda and mapper.dynamic_embeddings.shape[0] == int((mapper.voxel_id2embedding_id >= 0).sum())
    emb_before = mapper.dynamic_embeddings.clone()
    w_before = mapper.decoder.pts_linears[1].weight.detach().clone()
    share = Namespace(decoder=None, states=None)

    def rendered_loss():
        pose = f0.get_pose()
        assert f0.rays_d.device == f0.points.device and f0.rays_norm.shape == (len(pts), 1)     # the reference's attributes, on the points' device
        assert (f0.points[f0.rays_norm.reshape(-1) <= 30.0].shape[0]) > 0                        # mapping.py:260-262 style indexing
        d = f0.get_rays().reshape(-1, 3) @ pose[:3, :3].T.cuda()
        o = pose[:3, 3].reshape(1, 3).expand_as(d).cuda().contiguous()
        out = render_rays(o[None], d[None], mapper.map_states, mapper.decoder, mapper.step_size, 0.2, 0.3, 20, 50.0)
        z = out["z_vals"]; sdf = out["sdf"]; m = out["valid_mask"]
        gt = torch.from_numpy(np.linalg.norm(pts, axis=1) * cos).cuda()[out["ray_mask"].view(-1)]
        zc = z * torch.from_numpy(cos).cuda()[out["ray_mask"].view(-1)][:, None]
        near = m & ((zc - gt[:, None]).abs() < 0.3)
        return float((((zc + sdf * 0.3) - gt[:, None])[near] ** 2).mean())

    l0 = rendered_loss()
    for _ in range(3):
        mapper.do_mapping(share, f0, selection_method="current")
    l1 = rendered_loss()
    assert l1 < 0.5 * l0, (l0, l1)                                     # the map learns the surface
    assert not torch.equal(emb_before, mapper.dynamic_embeddings)      # in-place parameter updates
    assert not torch.equal(w_before, mapper.decoder.pts_linears[1].weight.detach())
    assert share.decoder is mapper.decoder and share.states["voxel_vertex_emb"] is mapper.dynamic_embeddings

    # mesh extraction (reference mapping.py:354-378 -> mesh_util.py:80-169): dense SDF grids of the surface voxels + marching cubes, both on the device
    mesh = mapper.extract_mesh(res=8, clean_mesh=False)
    mv, mt = np.asarray(mesh.vertices), np.asarray(mesh.triangles)
    assert len(mt) > 1000 and mt.min() == 0 and mt.max() == len(mv) - 1 and mv.dtype == np.float32
    mv = mv + np.float32(2000)                                          # (create_mesh shifts the vertices by the reference's offset of -2000)
    from scipy.spatial import cKDTree
    centres = mapper.map_states["voxel_center_xyz"][~mapper.map_states["voxel_vertex_idx"].eq(-1).any(-1)].cpu().numpy()
    dc, _ = cKDTree(centres).query(mv)
    assert dc.max() <= 0.5 * 0.2 * np.sqrt(3) + 1e-4                    # every vertex lies in a surface voxel
    dp, _ = cKDTree(pts).query(mv - np.float32(2000))                  # (the map lives at the frames' +2000 m offset, lidarFrame.py:18: the mesh's offset undoes it)
    assert np.median(dp) < 0.15, float(np.median(dp))                   # and the zero level the map has learnt runs along the scanned surface
    mapper.decoder.train()

    # tracking: perturb the pose of a second frame looking at the same scene, refine it
    tracker = Tracking(args)
    tracker.last_frame = f0
    P4 = np.eye(4); P4[:3, 3] = [0.06, -0.05, 0.02]
    f1 = LidarFrame(1, torch.from_numpy(pts), torch.from_numpy(cos), P4)
    err0 = float((f1.pose.translation() - f0.pose.translation()).norm())
    tracker.last_frame = f1                                             # start from the perturbed pose (no motion model yet)
    f2 = LidarFrame(2, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
    import queue
    kf_buffer = queue.Queue()
    out = tracker.do_tracking(share, f2, kf_buffer)
    assert kf_buffer.get_nowait() is out and kf_buffer.empty()         # tracking.py:144-147: every tracked frame goes to the mapper's queue
    err1 = float((out.pose.translation() - f0.pose.translation()).norm())
    assert out.pose.data.shape == (6,) and torch.isfinite(out.pose.data).all()
    assert err1 < 0.8 * err0, (err0, err1)                              # 10 steps on freshly drawn rays pull it back towards the true pose
    assert 0.9 < float(out.hit_ratio) <= 1.0


def test_render_rays_with_per_frame_origins_matches_the_oracle():
    """render_rays (render_helpers.py:190-318) takes world-space rays with ANY origins - bundle_adjust_frames hands it the concatenated
    rays of several frames (:385-388).  Two frames' rays in one call: hit mask, depths, validity bit for bit and sdf to 5e-6 against the
    oracle's two-frame iteration; tensors come back on the device."""
    from oracle import oracle as O
    from nerf_loam_amd import render_helpers as RH, synthetic as S
    from nerf_loam_amd.lidar_frame import LidarFrame
    from nerf_loam_amd.mapping import Mapping
    torch.manual_seed(3)
    pts, cos = H.scene_points(64, 64, 11)
    mapper = Mapping(make_args())
    f0 = LidarFrame(0, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
    mapper.create_voxels(f0)
    with torch.no_grad():
        mapper.dynamic_embeddings.copy_(torch.randn_like(mapper.dynamic_embeddings, dtype=torch.float32).mul_(0.01).to(torch.bfloat16))
    poses = [S.scan_pose(), S.scan_pose(tx=0.4, ty=-0.25, tz=0.05)]
    sel = [np.arange(0, len(pts), 3), np.arange(1, len(pts), 4)]
    dirs = S.unit_dirs(pts)
    o = torch.from_numpy(np.concatenate([np.broadcast_to(p[:3], (len(s_), 3)) for p, s_ in zip(poses, sel)]).astype(np.float32)).cuda()
    d = torch.from_numpy(np.concatenate([dirs[s_] for s_ in sel])).cuda()
    monkey_noise = RH.SAMPLER_NOISE
    RH.SAMPLER_NOISE = (777, False)
    try:
        out = RH.render_rays(o[None], d[None], mapper.map_states, mapper.decoder, mapper.step_size, 0.2, 0.3, 20, 50.0)
    finally:
        RH.SAMPLER_NOISE = monkey_noise
    assert all(out[k].is_cuda for k in ("z_vals", "sdf", "ray_mask", "valid_mask", "sampled_xyz"))
    mp = mapper.map_states
    ms = O.MapState(mp["voxel_center_xyz"].cpu().numpy(), mp["voxel_structure"].cpu().numpy(), mp["voxel_vertex_idx"].cpu().numpy(),
                    mp["voxel_id2embedding_id"].cpu().numpy().astype(np.int32), mapper.dynamic_embeddings.view(torch.int16).cpu().numpy().view(np.uint16), 0.2)
    dp = O.DecoderParams(*[p.detach().cpu().numpy() for p in mapper.decoder.param_list()])
    frames = [O.Frame(dirs[s_], pts[s_], cos[s_], p.copy()) for p, s_ in zip(poses, sel)]
    ref = O.render_and_grad(ms, dp, frames, O.IterCfg(step_size=mapper.step_size), want_emb_grad=False, want_dec_grad=False)
    assert np.array_equal(out["ray_mask"].view(-1).cpu().numpy(), ref["hits"])
    assert np.array_equal(out["valid_mask"].cpu().numpy(), ref["valid"]) and np.array_equal(out["z_vals"].cpu().numpy(), ref["z_vals"])
    assert float(np.abs(out["sdf"].cpu().numpy() - ref["sdf"]).max()) < 5e-6
    assert ref["hits"][:len(sel[0])].any() and ref["hits"][len(sel[0]):].any()


def _select_key_np(seed, n):
    """numpy restatement of nl_select_key (nl_device_math.h): lowbias32(i ^ (seed * 0x9E3779B9 + 0x7F4A7C15))"""
    x = np.arange(n, dtype=np.uint64) ^ np.uint64((seed * 0x9E3779B9 + 0x7F4A7C15) & 0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


@pytest.mark.parametrize("M,n", [(131072, 2048), (4097, 4096), (1000, 1), (300, 300), (50, 80), (70000, 35000)])
def test_device_ray_selection_is_the_exact_top_n_subset(M, n):
    """nl_select_rays: exactly n distinct rays, dataset order, == the n largest keys of a bijective hash"""
    from nerf_loam_amd import pipeline as P
    rng = np.random.default_rng(M)
    dirs = rng.normal(size=(M, 3)).astype(np.float32); pts = rng.normal(size=(M, 3)).astype(np.float32); cos = rng.random(M).astype(np.float32)
    sc = dict(dirs=torch.from_numpy(dirs).cuda(), points=torch.from_numpy(pts).cuda(), cos=torch.from_numpy(cos).cuda())
    eng = P.SdfEngine(max_rays=2 * min(n, M) + 8, samples_per_ray_cap=4)
    seed = 12345
    masks = eng.select_rays([sc, sc], n, seed, want_masks=True)
    k = min(n, M)
    assert eng.N == 2 * k
    for f in range(2):
        keys = _select_key_np((seed * 1000003 + f) & 0xFFFFFFFF, M)
        assert len(np.unique(keys)) == M                                    # bijection: no ties
        want = np.zeros(M, bool); want[np.argsort(keys)[M - k:]] = True
        got = masks[f].cpu().numpy().astype(bool)
        assert got.sum() == k and np.array_equal(got, want)
        sl = slice(f * k, (f + 1) * k)
        assert np.array_equal(eng.rays_d_sensor[sl].cpu().numpy(), dirs[want])   # dataset order kept
        assert np.array_equal(eng.points_gt[sl].cpu().numpy(), pts[want])
        assert np.array_equal(eng.cos_gt[sl].cpu().numpy(), cos[want])
        assert (eng.frame_id[sl].cpu().numpy() == f).all()
    # different seeds give different, equally sized subsets; every ray is picked about n/M of the time
    if M == 131072:
        hits = np.zeros(M)
        for sd in range(40):
            mk = eng.select_rays([sc], n, sd, want_masks=True)[0].cpu().numpy()
            assert mk.sum() == n
            hits += mk
        assert abs(hits.mean() - 40 * n / M) < 1e-9 and hits.max() <= 8     # Binomial(40, 1/64): P(> 8) ~ 1e-8 per ray


@pytest.mark.parametrize("shapes", [[(131072, 2048)], [(131072, 4096)] * 4, [(90001, 2048), (131072, 2048), (65536, 1024)], [(20000, 512)]])
def test_two_launch_selection_equals_the_radix_selection(shapes):
    """nl_select_rays_batch (window around the expected threshold, all frames in two launches) picks exactly the subset
    nl_select_rays (4-pass radix select per frame) picks: same rays, same order, same masks; its fail word stays clear"""
    from nerf_loam_amd import _lib as L, ops, pipeline as P
    rng = np.random.default_rng(len(shapes))
    scans = []
    for M, _ in shapes:
        scans.append(dict(dirs=torch.from_numpy(rng.normal(size=(M, 3)).astype(np.float32)).cuda(),
                          points=torch.from_numpy(rng.normal(size=(M, 3)).astype(np.float32)).cuda(),
                          cos=torch.from_numpy(rng.random(M).astype(np.float32)).cuda()))
    n = shapes[0][1]
    assert all(nn == n for _, nn in shapes) or True
    total = sum(nn for _, nn in shapes)
    eng = P.SdfEngine(max_rays=total + 8, samples_per_ray_cap=4, max_frames=8)
    for seed in (1, 77, 123456):
        # reference: the radix path, frame by frame
        ref = dict(d=torch.zeros(total, 3, device="cuda"), p=torch.zeros(total, 3, device="cuda"), c=torch.zeros(total, device="cuda"),
                   f=torch.zeros(total, dtype=torch.int32, device="cuda"))
        off, ref_masks = 0, []
        for f, (sc, (M, nn)) in enumerate(zip(scans, shapes)):
            ws = torch.empty(264 + 2 * M + (M + 1023) // 1024 + 8, dtype=torch.int32, device="cuda")
            mk = torch.empty(M, dtype=torch.uint8, device="cuda")
            ops.select_rays(M, nn, (seed * 1000003 + f) & 0xFFFFFFFF, sc["dirs"], sc["points"], sc["cos"], f, ref["d"][off:], ref["p"][off:],
                            ref["c"][off:], ref["f"][off:], mk, ws)
            ref_masks.append(mk); off += nn
        if len({nn for _, nn in shapes}) == 1:
            masks = eng.select_rays(scans, n, seed, want_masks=True)
        else:                                                     # per-frame counts differ: drive the batch entry point directly
            ws = torch.zeros(L.NL_SEL_MAX_FRAMES * L.NL_SEL_BATCH_WS_INTS_PER_FRAME, dtype=torch.int32, device="cuda")
            masks = [torch.empty(M, dtype=torch.uint8, device="cuda") for M, _ in shapes]
            offs = [sum(nn for _, nn in shapes[:f]) for f in range(len(shapes))]
            assert ops.select_rays_batch([M for M, _ in shapes], [nn for _, nn in shapes], [(seed * 1000003 + f) & 0xFFFFFFFF for f in range(len(shapes))],
                                         [sc["dirs"] for sc in scans], [sc["points"] for sc in scans], [sc["cos"] for sc in scans], masks, offs,
                                         eng.rays_d_sensor, eng.points_gt, eng.cos_gt, eng.frame_id, ws, 0)
            assert not ws.view(L.NL_SEL_MAX_FRAMES, -1)[:, 2].any()
        torch.cuda.synchronize()
        assert torch.equal(eng.rays_d_sensor[:total], ref["d"]) and torch.equal(eng.points_gt[:total], ref["p"])
        assert torch.equal(eng.cos_gt[:total], ref["c"]) and torch.equal(eng.frame_id[:total], ref["f"])
        for a_, b_ in zip(masks, ref_masks):
            assert torch.equal(a_, b_)
    assert getattr(eng, "_selb_ws", None) is None or not eng._selb_ws.view(L.NL_SEL_MAX_FRAMES, -1)[:, 2].any()


def test_mapping_and_tracking_with_device_ray_selection():
    """the reference loop with rays re-drawn ON THE DEVICE every iteration (render_helpers.RAY_SELECTION = "device")"""
    from nerf_loam_amd import render_helpers as RH
    from nerf_loam_amd.lidar_frame import LidarFrame
    from nerf_loam_amd.mapping import Mapping
    from nerf_loam_amd.tracking import Tracking
    torch.manual_seed(5)
    _reseed()
    pts, cos = H.scene_points(64, 64, 11)
    args = make_args()
    old = RH.RAY_SELECTION
    RH.RAY_SELECTION = "device"
    try:
        mapper = Mapping(args)
        f0 = LidarFrame(0, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
        mapper.create_voxels(f0)
        share = Namespace(decoder=None, states=None)
        emb_before = mapper.dynamic_embeddings.clone()
        for _ in range(3):
            mapper.do_mapping(share, f0, selection_method="current")
        assert f0.sample_mask is not None and f0.sample_mask.is_cuda and int(f0.sample_mask.sum()) == 1024
        assert not torch.equal(emb_before, mapper.dynamic_embeddings)
        tracker = Tracking(args)
        P4 = np.eye(4); P4[:3, 3] = [0.06, -0.05, 0.02]
        f1 = LidarFrame(1, torch.from_numpy(pts), torch.from_numpy(cos), P4)
        err0 = float((f1.pose.translation().detach() - f0.pose.translation().detach()).norm())
        tracker.last_frame = f1
        f2 = LidarFrame(2, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
        out = tracker.do_tracking(share, f2)
        err1 = float((out.pose.translation().detach() - f0.pose.translation().detach()).norm())
        assert err1 < 0.8 * err0, (err0, err1)                            # 10 refinement steps on freshly drawn rays pull the pose back
    finally:
        RH.RAY_SELECTION = old


def test_incremental_map_update_equals_a_full_rebuild():
    """frame-by-frame map growth through svo.export_delta + in-place device scatter keeps map_states equal to
    a full export, keeps the optimised embedding rows of earlier frames, and the traversal built from it intersects the
    same voxels as one built from scratch"""
    from nerf_loam_amd import pipeline as P
    from nerf_loam_amd.lidar_frame import LidarFrame
    from nerf_loam_amd.mapping import Mapping
    torch.manual_seed(3)
    args = make_args()
    mapper = Mapping(args)
    share = Namespace(decoder=None, states=None)
    frames = []
    for i, seed in enumerate((11, 12, 13)):
        pts, cos = H.scene_points(64, 48, seed)
        pose = np.eye(4); pose[0, 3] = 1.5 * i
        fr = LidarFrame(i, torch.from_numpy(pts), torch.from_numpy(cos), pose)
        frames.append(fr)
        rows_before = 0 if mapper.dynamic_embeddings is None else mapper.dynamic_embeddings.shape[0]
        emb_before = None if rows_before == 0 else mapper.dynamic_embeddings.clone()
        mapper.create_voxels(fr)
        ms = mapper.map_states
        c, s_, f = mapper.svo.export_device_layout()
        assert np.array_equal(ms["voxel_center_xyz"].cpu().numpy(), c)
        assert np.array_equal(ms["voxel_structure"].cpu().numpy(), s_)
        assert np.array_equal(ms["voxel_vertex_idx"].cpu().numpy(), f)
        used = np.unique(f[f >= 0])
        table = ms["voxel_id2embedding_id"].cpu().numpy()
        assert (table[used] >= 0).all() and len(np.unique(table[used])) == len(used) == mapper.dynamic_embeddings.shape[0]
        if emb_before is not None:                                           # old rows keep their (optimised) values, new rows are zero
            assert torch.equal(mapper.dynamic_embeddings[:rows_before], emb_before)
            assert mapper.dynamic_embeddings.shape[0] > rows_before and not mapper.dynamic_embeddings[rows_before:].any()
        md = ms["_device"]
        fresh = P.MapDevice.from_tensors(torch.from_numpy(c), torch.from_numpy(s_), torch.from_numpy(f), torch.from_numpy(table),
                                         mapper.dynamic_embeddings, mapper.voxel_size)
        assert torch.equal(md.blk_ids, fresh.blk_ids) and torch.equal(md.blk_hdr, fresh.blk_hdr) and torch.equal(md.vertex_rows, fresh.vertex_rows)
        mapper.do_mapping(share, fr, selection_method="current")            # optimise on the grown map (rows change in place)
        assert mapper.map_states["voxel_vertex_emb"].data_ptr() == mapper.dynamic_embeddings.data_ptr()
    assert mapper.svo.count_nodes() == mapper.map_states["voxel_center_xyz"].shape[0]


def test_device_resident_share_data_hand_off():
    """nerf_loam_amd.share.ShareData - the reference's ShareData attributes, snapshots kept on the device in two
    alternating buffers: publication isolates the tracker from later mapper updates, no host round trip"""
    from nerf_loam_amd.lidar_frame import LidarFrame
    from nerf_loam_amd.mapping import Mapping
    from nerf_loam_amd.share import ShareData
    from nerf_loam_amd.tracking import Tracking
    torch.manual_seed(777)
    _reseed()                                                 # the device-side seed stream from its start: the outcome does not depend on test order
    pts, cos = H.scene_points(64, 64, 11)
    args = make_args()
    mapper = Mapping(args)
    share = ShareData()
    assert share.states is None and share.decoder is None
    f0 = LidarFrame(0, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
    mapper.create_voxels(f0)
    for _ in range(3):
        mapper.do_mapping(share, f0, selection_method="current")
    assert share.version == 3
    st = share.states
    assert st["voxel_vertex_emb"].is_cuda and st["voxel_vertex_emb"].data_ptr() != mapper.dynamic_embeddings.data_ptr()
    for k in ("voxel_center_xyz", "voxel_structure", "voxel_vertex_idx"):
        assert torch.equal(st[k], mapper.map_states[k])
    assert torch.equal(st["voxel_vertex_emb"], mapper.dynamic_embeddings)
    w_pub = share.decoder.pts_linears[1].weight.detach().clone()
    assert torch.equal(w_pub, mapper.decoder.pts_linears[1].weight.detach()) and share.decoder is not mapper.decoder
    # the mapper keeps optimising: the published snapshot does not move until the next publication
    snap = st["voxel_vertex_emb"].clone(); ptr3 = st["voxel_vertex_emb"].data_ptr()
    emb_clean = mapper.dynamic_embeddings.detach().clone()
    mapper.dynamic_embeddings.add_(1.0)
    assert torch.equal(share.states["voxel_vertex_emb"], snap) and share.states is st
    mapper.update_share_data(share)
    st4 = share.states
    assert share.version == 4 and st4 is not st and st4["voxel_vertex_emb"].data_ptr() != ptr3       # the other buffer
    assert torch.equal(st4["voxel_vertex_emb"], mapper.dynamic_embeddings)
    mapper.dynamic_embeddings.copy_(emb_clean)                                              # the optimised map again, bit for bit
    mapper.update_share_data(share)
    assert share.version == 5 and share.states["voxel_vertex_emb"].data_ptr() == ptr3              # buffers alternate
    # the tracker consumes the snapshot exactly like the reference's share_data
    tracker = Tracking(args)
    P4 = np.eye(4); P4[:3, 3] = [0.06, -0.05, 0.02]
    f1 = LidarFrame(1, torch.from_numpy(pts), torch.from_numpy(cos), P4)
    err0 = float((f1.pose.translation().detach() - f0.pose.translation().detach()).norm())
    tracker.last_frame = f1
    out = tracker.do_tracking(share, LidarFrame(2, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4)))
    err1 = float((out.pose.translation().detach() - f0.pose.translation().detach()).norm())
    # (the hand-off is under test - the numeric quality of track_frame is tests/test_gpu_api_parity.py's; the margin by which ten steps on
    #  a map of three mapping calls pull the pose back moves from run to run)
    print("share hand-off tracking", err0, err1)
    assert err1 < 0.8 * err0, (err0, err1)                                                  # measured 0.57


def test_get_scores_matches_oracle():
    """mesh-time dense SDF grid (reference render_helpers.get_scores): HIP gather + decoder forward vs the oracle"""
    from nerf_loam_amd.decoder import Decoder
    from nerf_loam_amd.render_helpers import get_scores
    from oracle import oracle as O
    sc = H.build_oracle_scene(64, 16, 21)
    ms = sc["ms"]
    surf = np.nonzero(ms.vertex_idx[:, 0] >= 0)[0][:300]                       # SURFACE voxels only, like the mesher passes
    d0 = O.decoder_init(21)
    dec = Decoder().cuda()
    dec.load_flat(torch.from_numpy(np.concatenate([a.reshape(-1) for a in (d0.W1, d0.b1, d0.W2, d0.b2, d0.W3, d0.b3)])).cuda())
    emb = torch.from_numpy(O.bf16_to_f32(ms.emb)).to(torch.bfloat16).cuda()
    states = {"voxel_vertex_idx": torch.from_numpy(ms.vertex_idx[surf]), "voxel_center_xyz": torch.from_numpy(ms.centres[surf]),
              "voxel_structure": torch.from_numpy(ms.structure[surf]), "voxel_vertex_emb": emb,
              "voxel_id2embedding_id": torch.from_numpy(ms.id2row)}
    res = 4
    got = get_scores(dec, states, 0.2, bits=res).numpy()
    assert got.shape == (len(surf), res, res, res, 1)
    lin = np.linspace(-0.5, 0.5, res, dtype=np.float32)
    offs = (np.stack(np.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3) * np.float32(0.2)).astype(np.float32)
    xyz = (ms.centres[surf][:, None, :] + offs[None]).reshape(-1, 3).astype(np.float32)
    vox = np.repeat(surf, res ** 3)
    feats, _ = O.trilinear_forward(xyz, vox, ms.centres, ms.vertex_rows(), ms.emb, 0.2)
    ref, _ = O.decoder_forward(feats, d0)
    assert np.abs(got.reshape(-1) - ref).max() < 1e-5


def test_get_scores_matches_the_reference(monkeypatch):
    """the same call against the REFERENCE's own get_scores (tests/golden/scores_res4.npz: render_helpers.py:96-153 imported and run by
    tests/golden/make_golden.py): grid points generated on the device (nl_gather_grid) from torch's linspace, one gather + one forward launch per chunk -
    also with a chunk of 7 voxels (43 launch pairs) and through the torch tensors of a whole map at res 8"""
    from nerf_loam_amd.decoder import Decoder
    from nerf_loam_amd import render_helpers as RH
    from oracle import oracle as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scores_res4.npz"))
    sc = H.build_oracle_scene(int(g["n_beams"]), int(g["n_azimuth"]), int(g["seed"]))
    ms = sc["ms"]
    ms.id2row = g["id_table"].copy()
    surf, res = g["surf"], int(g["res"])
    assert np.array_equal(ms.centres[surf], g["centres"])                         # the same map, the same voxels
    d0 = O.decoder_init(int(g["seed"]))
    dec = Decoder().cuda()
    dec.load_flat(torch.from_numpy(np.concatenate([a.reshape(-1) for a in (d0.W1, d0.b1, d0.W2, d0.b2, d0.W3, d0.b3)])).cuda())
    emb = torch.from_numpy(H.init_embeddings(len(ms.emb), int(g["seed"]))).to(torch.bfloat16).cuda()
    states = {"voxel_vertex_idx": torch.from_numpy(ms.vertex_idx[surf]), "voxel_center_xyz": torch.from_numpy(ms.centres[surf]),
              "voxel_structure": torch.from_numpy(ms.structure[surf]), "voxel_vertex_emb": emb, "voxel_id2embedding_id": torch.from_numpy(ms.id2row)}
    got = RH.get_scores(dec, states, float(g["voxel_size"]), bits=res).numpy()
    assert got.shape == g["sdf"].shape
    err = float(np.abs(got - g["sdf"]).max())
    assert err < 5e-6, err                                                          # (the bar of every reference-golden sdf comparison)
    monkeypatch.setattr(RH, "SCORES_CHUNK_POINTS", 7 * res ** 3)
    assert np.array_equal(RH.get_scores(dec, states, float(g["voxel_size"]), bits=res).numpy(), got)
    monkeypatch.undo()
    dev = RH.get_scores(dec, states, float(g["voxel_size"]), bits=8, device_out=True)
    assert dev.is_cuda and dev.shape == (len(surf), 8, 8, 8, 1) and torch.isfinite(dev).all()
    # the corner points of the res-8 grid are the corner points of the res-4 grid (linspace end points)
    assert np.abs(dev[:, ::7, ::7, ::7, 0].cpu().numpy() - got[:, ::3, ::3, ::3, 0]).max() < 1e-6


def test_predraw_with_a_varying_iteration_count_meets_no_stale_candidate_counter():
    """calls of 20, 5, 5 + 1 odd one, then 20 iterations on one engine (the tracker's first call runs 5x the iterations of the later ones,
    tracking.py:42): the (iteration, frame) slots the short calls skip keep the long call's candidate counters at ITS parity - the engine
    clears them before a later call uses them again.  Every subset of the last call against a fresh engine's draw, bitwise."""
    from nerf_loam_amd import pipeline as P
    rng = np.random.default_rng(11)
    M, n = 131072, 2048
    scans = [dict(dirs=torch.from_numpy(rng.normal(size=(M, 3)).astype(np.float32)).cuda(),
                  points=torch.from_numpy(rng.normal(size=(M, 3)).astype(np.float32)).cuda(), cos=torch.from_numpy(rng.random(M).astype(np.float32)).cuda())]
    eng = P.SdfEngine(max_rays=n, samples_per_ray_cap=4, max_frames=2)
    ref = P.SdfEngine(max_rays=n, samples_per_ray_cap=4, max_frames=2)
    for k, iters in enumerate((20, 5, 5, 20, 7, 20)):
        seeds = [50 * k + it for it in range(iters)]
        assert eng.predraw(scans, n, seeds)
        torch.cuda.synchronize()
        assert not eng.adam_state[3].item()
        pre = eng._pre
        for it in range(iters):
            masks = ref.select_rays(scans, n, seeds[it], want_masks=True)
            sl = slice(it * n, (it + 1) * n)
            assert torch.equal(pre["d"][sl], ref.rays_d_sensor[:n]) and torch.equal(pre["c"][sl], ref.cos_gt[:n]), (k, iters, it)
            assert torch.equal(pre["masks"][0][it], masks[0]), (k, iters, it)


@pytest.mark.parametrize("shapes,iters", [([(131072, 2048)], 20), ([(131072, 4096)] * 4, 15), ([(90001, 2048), (131072, 2048), (65536, 1024)], 5)])
def test_predrawn_subsets_equal_the_per_iteration_selection(shapes, iters):
    """SdfEngine.predraw (the ray subsets of ALL iterations of a call drawn up front, eight (iteration, frame) pairs per two launches)
    gives every iteration exactly the rays, order, frame ids and masks that select_rays draws for that iteration's seed"""
    from nerf_loam_amd import pipeline as P
    rng = np.random.default_rng(7)
    scans = []
    for M, _ in shapes:
        scans.append(dict(dirs=torch.from_numpy(rng.normal(size=(M, 3)).astype(np.float32)).cuda(),
                          points=torch.from_numpy(rng.normal(size=(M, 3)).astype(np.float32)).cuda(),
                          cos=torch.from_numpy(rng.random(M).astype(np.float32)).cuda()))
    n = max(nn for _, nn in shapes)
    tot = sum(min(n, M) for M, _ in shapes)
    eng = P.SdfEngine(max_rays=tot, samples_per_ray_cap=4, max_frames=8)
    ref = P.SdfEngine(max_rays=tot, samples_per_ray_cap=4, max_frames=8)
    seeds = [1000 + 17 * it for it in range(iters)]
    for call in range(2):                                          # twice: the workspace parity alternates between calls
        assert eng.predraw(scans, n, seeds)
        torch.cuda.synchronize()
        assert not eng.adam_state[3].item()
        pre = eng._pre
        for it in (0, 1, iters // 2, iters - 1):
            masks = ref.select_rays(scans, n, seeds[it], want_masks=True)
            sl = slice(it * tot, (it + 1) * tot)
            assert torch.equal(pre["d"][sl], ref.rays_d_sensor[:tot]) and torch.equal(pre["p"][sl], ref.points_gt[:tot])
            assert torch.equal(pre["c"][sl], ref.cos_gt[:tot]) and torch.equal(pre["f"][sl], ref.frame_id[:tot])
            for f in range(len(shapes)):
                assert torch.equal(pre["masks"][f][it], masks[f])
            eng.use_predrawn(it)
            d = eng._desc
            assert d.rays_d_sensor == pre["d"].data_ptr() + 12 * it * tot and d.frame_id == pre["f"].data_ptr() + 4 * it * tot and eng.N == tot
