"""Shared scene construction for the parity tests (oracle side).  Rebuilds, from the few scalars a
tests/golden/*.npz fixture stores, exactly the inputs tests/golden/make_golden.py fed the reference."""
import numpy as np

from nerf_loam_amd import synthetic as S
from oracle import oracle as O

VOXEL = 0.2


def scene_points(n_beams, n_azimuth, seed):
    return S.synthetic_scan(n_beams, n_azimuth, seed, range_noise=0.01, sector=(0.1, 0.1 + n_azimuth / 2048.0))


def init_embeddings(E, seed):
    return np.random.default_rng(seed + 1000).normal(0, 0.01, (E, 16)).astype(np.float32)


def build_oracle_scene(n_beams, n_azimuth, seed, voxel=VOXEL):
    pts, cos = scene_points(n_beams, n_azimuth, seed)
    pose6 = S.scan_pose()
    vox = S.voxel_coords(pts, O.rodrigues(pose6[3:]), pose6[:3], voxel)
    oc = O.Octree()
    oc.init(256 * 256 * 4, 16, voxel)
    oc.insert(vox)
    voxels, children, features = oc.get_centres_and_children()
    centres, structure = O.grid_features(voxels, children, voxel)
    id2row = -np.ones(len(voxels), np.int32)
    E = O.assign_embedding_rows(features, id2row, 0)
    emb = O.bf16_bits(init_embeddings(E, seed))
    ms = O.MapState(centres, structure, features, id2row, emb, voxel)
    return dict(points=pts, cos=cos, vox=vox, ms=ms, E=E, voxels=voxels, children=children)


def unpack_masks(packed, n_points):
    return np.unpackbits(packed, axis=-1)[..., :n_points].astype(bool)


def scatter_rows(n_rows, rows, vals, base=None):
    out = np.zeros((n_rows, vals.shape[1]), np.uint16) if base is None else base.copy()
    out[rows] = vals
    return out


def record_gpu_metric(key, **kw):
    """measured numbers of the -m gpu tests -> gpurun_out/gpu_metrics.json (merged back from the GPU box; the bars in the tests were set
    from these files)"""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "gpu_metrics.json")
    try:
        data = json.load(open(path))
    except Exception:                                           # noqa: BLE001
        data = {}
    data.setdefault(key, {}).update({k: float(v) for k, v in kw.items()})
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
