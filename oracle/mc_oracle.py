"""TEST INFRASTRUCTURE - CPU restatement of the per-voxel marching cubes of mesh extraction; never imported by the product (nerf_loam_amd/).

Reference call site: MeshExtractor.marching_cubes, /root/reference/src/utils/mesh_util.py:145-169 - for every surface voxel whose res^3 SDF grid changes
sign, `skimage.measure.marching_cubes(sdf_volume, 0, spacing=[1/(res-1)]*3)`, then `verts = (verts - 0.5) * voxel_size + centre`, faces offset by the running
vertex count, everything concatenated in voxel order.

PARITY UNPINNED for the triangulation: scikit-image (the reference's requirements name no version) is not installed in this image and is not vendored under
/root/reference, so neither its Lewiner case tables nor its vertex / face ORDER can be reproduced or checked.  What IS table-independent and is pinned
analytically by the tests (tests/test_mc_oracle.py): the VERTEX SET of any marching-cubes variant with linear interpolation - one vertex per lattice edge
whose end values change sign, at v0 / (v0 - v1) along the edge - which voxels are skipped (min > 0 or max < 0, mesh_util.py:158-159), the affine map to world
coordinates, a closed consistently-oriented surface, and the area / enclosed volume of analytic shapes.  The case table is derived from the cube's geometry by
scripts/gen_mc_table.py (oracle/mc_table.json); within a voxel, vertices are numbered in lattice-edge order (axis, ix, iy, iz) and faces in cell order.
"""
import json
import os

import numpy as np

_T = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_table.json")))
TRIS = _T["tris"]                      # [256] lists of (e0, e1, e2), edge e = 4 * axis + bit(lower other axis) + 2 * bit(higher other axis)


def _edge_base(e):
    """(axis, offset of the edge's lower end point inside the cell)"""
    a, idx = e // 4, e % 4
    others = [i for i in range(3) if i != a]
    off = [0, 0, 0]
    off[others[0]] = idx & 1
    off[others[1]] = idx >> 1
    return a, off


def marching_cubes_voxel(vol):
    """vol [res, res, res] float32 -> (verts [n, 3] float32 in lattice-index coordinates, faces [m, 3] int32), inside = value < 0"""
    res = vol.shape[0]
    vol = np.asarray(vol, np.float32)
    vid = -np.ones((3, res, res, res), np.int64)
    verts = []
    for a in range(3):
        for i in range(res):
            for j in range(res):
                for k in range(res):
                    p = [i, j, k]
                    if p[a] + 1 >= res:
                        continue
                    q = list(p)
                    q[a] += 1
                    v0, v1 = vol[i, j, k], vol[q[0], q[1], q[2]]
                    if (v0 < 0) != (v1 < 0):
                        t = np.float32(v0) / (np.float32(v0) - np.float32(v1))
                        pos = np.array(p, np.float32)
                        pos[a] = np.float32(p[a]) + t
                        vid[a, i, j, k] = len(verts)
                        verts.append(pos)
    faces = []
    for i in range(res - 1):
        for j in range(res - 1):
            for k in range(res - 1):
                cfg = 0
                for c in range(8):
                    if vol[i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1)] < 0:
                        cfg |= 1 << c
                for tri in TRIS[cfg]:
                    f = []
                    for e in tri:
                        a, off = _edge_base(e)
                        f.append(vid[a, i + off[0], j + off[1], k + off[2]])
                    assert min(f) >= 0
                    faces.append(f)
    return (np.array(verts, np.float32).reshape(-1, 3), np.array(faces, np.int32).reshape(-1, 3))


def marching_cubes(voxels, sdf, voxel_size):
    """the reference's MeshExtractor.marching_cubes (mesh_util.py:145-169): voxels [n, >= 3] centres, sdf [n, res, res, res(, 1)] -> (verts [N, 3] f32, faces [M, 3] i32)"""
    voxels = np.asarray(voxels, np.float32)[:, :3]
    sdf = np.asarray(sdf, np.float32)
    if sdf.ndim == 5:
        sdf = sdf[..., 0]
    res = sdf.shape[1]
    spacing = np.float32(1.0 / (res - 1))
    tv, tf, nv = [], [], 0
    for i in range(len(voxels)):
        vol = sdf[i]
        if vol.min() > 0 or vol.max() < 0:
            continue
        v, f = marching_cubes_voxel(vol)
        v = (v * spacing - np.float32(0.5)) * np.float32(voxel_size) + voxels[i]
        tv.append(v.astype(np.float32))
        tf.append(f + nv)
        nv += len(v)
    if not tv:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    return np.concatenate(tv), np.concatenate(tf).astype(np.int32)
