#!/usr/bin/env python3
"""Build-container helper of tests/test_reference_mapping_dropin.py (needs /root/reference): runs the REFERENCE's own
src/mapping.py - Mapping.__init__, create_voxels, get_embeddings, update_grid_features, unmodified - on CPU on top of a chosen
`svo` TorchScript library and dumps the resulting map_states.

    python tests/ref_mapping_probe.py <ours|ref> <out.npz>

`ours` = nerf_loam_amd/libnl_svo_torch.so (TORCH_LIBRARY(svo) over the C ABI), `ref` = oracle/_ref/svo_ref.so (the reference's
C++ built unmodified).  One library per process: both register the same TorchScript names.  What is neutralised, all outside
the octree: the modules the reference imports but this image lacks (open3d, cv2, skimage) are stubbed, `.cuda()` is the identity,
the hard-coded absolute path of torch.classes.load_library (src/mapping.py:19-20) is redirected - exactly the one line a
maintainer changes - and the 2e9-row id table (mapping.py:76, 8 GB) is capped at 4 M rows."""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NL_REFERENCE_ROOT", "/root/reference")
which, out_path = sys.argv[1], sys.argv[2]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf_loam_amd", "dropin"))          # `import grid` -> the drop-in alias module
sys.path.insert(0, os.path.join(REF, "src"))

for name in ("open3d", "cv2", "skimage", "skimage.measure", "trimesh"):
    m = types.ModuleType(name)
    m.__dict__.setdefault("marching_cubes", None)
    sys.modules[name] = m
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.cuda.empty_cache = lambda: None
torch.cuda.synchronize = lambda *a, **k: None

if which == "ours":
    from nerf_loam_amd import build
    build.build(); lib_path = build.build_torch_ext()
else:
    lib_path = os.path.join(ROOT, "oracle", "_ref", "svo_ref.so")
_load = torch.classes.load_library
torch.classes.load_library = lambda path: _load(lib_path)                    # the one line of src/mapping.py a maintainer edits
_ones = torch.ones
torch.ones = lambda *a, **k: _ones((1 << 22, 1), **k) if a and tuple(a[0]) == (int(2e9), 1) else _ones(*a, **k)

import mapping as RM                                                       # noqa: E402  the reference's src/mapping.py
from lidarFrame import LidarFrame                                          # noqa: E402
from nerf_loam_amd import synthetic as S                                   # noqa: E402

args = Namespace(
    decoder="lidar", criteria=dict(sdf_weight=10000.0, fs_weight=1, eiko_weight=0.1, sdf_truncation=0.30),
    data_specs=dict(max_depth=50.0, min_depth=1.5),
    decoder_specs=dict(depth=2, width=256, in_dim=16, skips=[], embedder="none", multires=0),
    mapper_specs=dict(N_rays_each=2048, use_local_coord=False, voxel_size=0.2, step_size=0.5, window_size=4, num_iterations=10,
                      max_voxel_hit=20, final_iter=True, mesh_res=2, learning_rate_emb=0.03, learning_rate_decorder=0.005,
                      learning_rate_pose=0.001, freeze_frame=5, keyframe_gap=8, remove_back=False, key_distance=12),
    debug_args=dict(verbose=False, mesh_freq=100))
mapper = RM.Mapping(args, None)
res = {}
for i, seed in enumerate((11, 12)):                                         # two frames: the second one grows the map
    pts, cos = S.synthetic_scan(64, 48, seed, range_noise=0.01, sector=(0.1, 0.1 + 48 / 2048.0))
    P4 = np.eye(4); P4[0, 3] = 1.5 * i
    fr = LidarFrame(i, torch.from_numpy(pts), torch.from_numpy(cos), P4)
    mapper.create_voxels(fr)                                               # svo.insert + update_grid_features + get_embeddings
    ms = mapper.map_states
    n = ms["voxel_center_xyz"].shape[0]
    res[f"f{i}_vertex_idx"] = ms["voxel_vertex_idx"].numpy()
    res[f"f{i}_centres"] = ms["voxel_center_xyz"].detach().numpy()
    res[f"f{i}_structure"] = ms["voxel_structure"].numpy()
    res[f"f{i}_id_table"] = ms["voxel_id2embedding_id"][:n, 0].numpy()
    res[f"f{i}_emb_shape"] = np.array(ms["voxel_vertex_emb"].shape)
    res[f"f{i}_counts"] = np.array([mapper.svo.count_nodes(), mapper.svo.count_leaf_nodes()])
res["encode"] = torch.ops.svo.encode(torch.tensor([[1, 2, 3], [70000, 5, 123456]], dtype=torch.int64)).numpy()
res["has_voxel"] = np.array([bool(mapper.svo.has_voxel(torch.tensor([10000, 10000, 9991], dtype=torch.int32))),
                             bool(mapper.svo.has_voxel(torch.tensor([1, 1, 1], dtype=torch.int32)))])
res["leaf_voxels"] = mapper.svo.get_leaf_voxels().numpy()
res["voxels_dfs"] = mapper.svo.get_voxels().numpy()
np.savez(out_path, **res)
print("ok", which, lib_path)
