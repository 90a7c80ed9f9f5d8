"""cProfile of Tracking.do_tracking on the frame-loop workload: what a tracked frame costs on the host beside its 20 iterations (GPU only)."""
import os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8"); os.environ.setdefault("OMP_NUM_THREADS", "8")
import cProfile, pstats, queue, sys, time
from argparse import Namespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loam_amd import synthetic as S, hostenv
from nerf_loam_amd.lidar_frame import LidarFrame
from nerf_loam_amd.mapping import Mapping
from nerf_loam_amd.tracking import Tracking
hostenv.cap_host_thread_pools(8)
args = Namespace(criteria=dict(sdf_weight=10000.0, fs_weight=1, eiko_weight=0.1, sdf_truncation=0.30), data_specs=dict(max_depth=50.0, min_depth=1.5),
                 decoder_specs=dict(depth=2, width=256, in_dim=16, skips=[], embedder="none", multires=0),
                 tracker_specs=dict(N_rays=2048, learning_rate=0.005, step_size=0.2, max_voxel_hit=20, num_iterations=20),
                 mapper_specs=dict(N_rays_each=2048, use_local_coord=False, voxel_size=0.2, step_size=0.5, window_size=4, num_iterations=20, max_voxel_hit=20,
                                   final_iter=True, mesh_res=8, learning_rate_emb=0.03, learning_rate_decorder=0.005, learning_rate_pose=0.001, freeze_frame=20,
                                   keyframe_gap=8, remove_back=False, key_distance=12), debug_args=dict(verbose=False, mesh_freq=100))
torch.manual_seed(777)
mapper, tracker = Mapping(args), Tracking(args)
share = Namespace(decoder=None, states=None)
kf = queue.Queue()
frames = []
for i in range(8):
    pts, cos = S.synthetic_scan(seed=777 + i, range_noise=0.01)
    frames.append(LidarFrame(i, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4)))
mapper.create_voxels(frames[0]); mapper.do_mapping(share, frames[0], selection_method="current"); tracker.last_frame = frames[0]
for fr in frames[1:4]:
    tracker.do_tracking(share, fr, kf); mapper.create_voxels(fr); mapper.do_mapping(share, fr, selection_method="current")
torch.cuda.synchronize()
pr = cProfile.Profile()
ts = []
for fr in frames[4:]:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable(); tracker.do_tracking(share, fr, kf); torch.cuda.synchronize(); pr.disable()
    ts.append((time.perf_counter() - t0) * 1e3)
    mapper.create_voxels(fr); mapper.do_mapping(share, fr, selection_method="current")
print("do_tracking ms per frame:", [round(t, 2) for t in ts])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
