"""child-process side of tests/test_gpu_share_ipc.py: the tracker's end of ShareData in ANOTHER process.  Imported by the spawned
process (torch.multiprocessing, spawn); the ShareData handles arrive as arguments (device buffers = HIP IPC handles)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def tracker_process(handles, pts, cos, q_in, q_out):
    import numpy as np
    import torch
    try:
        torch.cuda.set_device(0)
        from nerf_loam_amd.share import ShareData
        share = ShareData.attach(handles)
        held = None
        while True:
            msg = q_in.get()
            if msg == "stop":
                break
            if msg == "read":                                   # take (a lease on) the newest snapshot
                st = share.states
                held = st
                dec = share.decoder
                q_out.put(dict(version=share.version, n=int(st["voxel_center_xyz"].shape[0]), rows=int(st["voxel_vertex_emb"].shape[0]),
                               emb_sum=float(st["voxel_vertex_emb"].float().abs().sum()), centres_sum=float(st["voxel_center_xyz"].double().sum()),
                               dec_sum=float(dec.pts_linears[1].weight.double().abs().sum()), ptr=int(st["voxel_vertex_emb"].data_ptr())))
            elif msg == "recheck":                              # the snapshot taken earlier, NOT re-read: still the same bytes?
                q_out.put(dict(emb_sum=float(held["voxel_vertex_emb"].float().abs().sum()), version_seen=share.version))
            elif msg == "track":                                # the tracker's real work on the shared snapshot
                from test_gpu_api_mirror import make_args
                from nerf_loam_amd.lidar_frame import LidarFrame
                from nerf_loam_amd.tracking import Tracking
                tracker = Tracking(make_args())
                P4 = np.eye(4); P4[:3, 3] = [0.06, -0.05, 0.02]
                f1 = LidarFrame(1, torch.from_numpy(pts), torch.from_numpy(cos), P4)
                err0 = float(f1.pose.translation().detach().norm() - 0.0)
                tracker.last_frame = f1
                out = tracker.do_tracking(share, LidarFrame(2, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4)))
                t = out.pose.translation().detach().cpu().numpy() - 2000.0
                q_out.put(dict(err0=float(np.linalg.norm([0.06, -0.05, 0.02])), err1=float(np.linalg.norm(t)), hit_ratio=float(out.hit_ratio)))
    except Exception as e:                                      # noqa: BLE001
        import traceback
        q_out.put(dict(error=traceback.format_exc()))
