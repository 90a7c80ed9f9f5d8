"""where does Mapping.create_voxels spend a frame's map update?  (standing sensor, fresh noise per scan; wall clock per statement group, device synchronised)"""
import os, sys, time
from argparse import Namespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loam_amd import synthetic as S, pipeline as P
from nerf_loam_amd.lidar_frame import LidarFrame
from nerf_loam_amd.mapping import Mapping
args = Namespace(criteria=dict(sdf_weight=10000.0, fs_weight=1, eiko_weight=0.1, sdf_truncation=0.30), data_specs=dict(max_depth=50.0, min_depth=1.5),
                 decoder_specs=dict(depth=2, width=256, in_dim=16, skips=[], embedder="none", multires=0),
                 mapper_specs=dict(N_rays_each=2048, use_local_coord=False, voxel_size=0.2, step_size=0.5, window_size=4, num_iterations=20, max_voxel_hit=20,
                                   final_iter=True, mesh_res=8, learning_rate_emb=0.03, learning_rate_decorder=0.005, learning_rate_pose=0.001, freeze_frame=20,
                                   keyframe_gap=8, remove_back=False, key_distance=12), debug_args=dict(verbose=False, mesh_freq=100))
m = Mapping(args)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for i in range(6):
    pts, cos = S.synthetic_scan(seed=777 + i, range_noise=0.01)
    fr = LidarFrame(i, torch.from_numpy(pts), torch.from_numpy(cos), np.eye(4))
    t0 = sync()
    pose = fr.get_pose().detach()
    p = fr.get_points().float() @ pose[:3, :3].transpose(-1, -2) + pose[:3, 3]
    vox = torch.div(p, m.voxel_size, rounding_mode="floor").cpu().int()
    t1 = sync()
    m.svo.insert(vox)
    t2 = sync()
    ids, centres, structure, vertex_idx = m.svo.export_delta()
    t3 = sync()
    n = m.svo.count_nodes(); m._grow_nodes(n)
    new_ids = m.get_embeddings(torch.from_numpy(vertex_idx))
    t4 = sync()
    nb = m._node_buf
    if len(ids):
        di = torch.from_numpy(ids).to(m.device).long()
        nb["centres"].index_copy_(0, di, torch.from_numpy(centres).to(m.device)); nb["structure"].index_copy_(0, di, torch.from_numpy(structure).to(m.device))
        nb["vertex_idx"].index_copy_(0, di, torch.from_numpy(vertex_idx).to(m.device))
    if new_ids.numel():
        nb["id2row"].index_copy_(0, new_ids.to(m.device), m.voxel_id2embedding_id[new_ids].to(m.device))
    t5 = sync()
    md = P.MapDevice.from_tensors(nb["centres"][:n], nb["structure"][:n], nb["vertex_idx"][:n], nb["id2row"][:n], m.dynamic_embeddings, m.voxel_size, m.device)
    t6 = sync()
    print(f"frame {i}: transform+floor {1e3*(t1-t0):5.2f}  insert {1e3*(t2-t1):5.2f}  export_delta {1e3*(t3-t2):5.2f} ({len(ids)} rows)  embeddings {1e3*(t4-t3):5.2f}  uploads {1e3*(t5-t4):5.2f}  "
          f"MapDevice.from_tensors {1e3*(t6-t5):5.2f}  | total {1e3*(t6-t0):5.2f} ms   nodes {n}", flush=True)
