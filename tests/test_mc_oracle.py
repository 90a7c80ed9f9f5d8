"""The marching-cubes oracle (oracle/mc_oracle.py) and its generated case table: table-independent properties of the reference's per-voxel extraction
(src/utils/mesh_util.py:145-169 - skimage is not available here, see the oracle's header: the triangulation itself is unpinned)."""
import os
import subprocess
import sys
from collections import Counter

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mc_oracle as MC  # noqa: E402


def _grid(res):
    g = np.arange(res, dtype=np.float32)
    return np.meshgrid(g, g, g, indexing="ij")


def _edge_census(faces):
    """directed-edge counts of a triangle list"""
    c = Counter()
    for a, b, d in faces:
        c[(a, b)] += 1; c[(b, d)] += 1; c[(d, a)] += 1
    return c


def assert_closed_and_oriented(faces):
    c = _edge_census(faces.tolist())
    for (a, b), n in c.items():
        assert n == 1, f"directed edge {(a, b)} used {n} times"
        assert c.get((b, a), 0) == 1, f"edge {(a, b)} has no opposite partner: the surface has a hole or a flipped triangle"


def signed_volume_and_area(v, f):
    p0, p1, p2 = v[f[:, 0]].astype(np.float64), v[f[:, 1]].astype(np.float64), v[f[:, 2]].astype(np.float64)
    n = np.cross(p1 - p0, p2 - p0)
    return float(np.einsum("ij,ij->i", p0, n).sum() / 6.0), float(np.linalg.norm(n, axis=1).sum() / 2.0)


def test_the_committed_tables_are_what_the_generator_derives():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_mc_table.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_case_uses_each_crossed_edge_and_complements_mirror():
    for cfg in range(256):
        inside = [(cfg >> c) & 1 for c in range(8)]
        crossed = set()
        for e in range(12):
            a, off = MC._edge_base(e)
            c0 = off[0] | (off[1] << 1) | (off[2] << 2)
            c1 = c0 | (1 << a)
            if inside[c0] != inside[c1]:
                crossed.add(e)
        used = {e for tri in MC.TRIS[cfg] for e in tri}
        assert used == crossed, cfg
        assert len(MC.TRIS[cfg]) <= 5
    assert MC.TRIS[0] == [] and MC.TRIS[255] == []


@pytest.mark.parametrize("res,r,tol", [(8, 2.6, 0.12), (12, 4.3, 0.05), (16, 6.1, 0.03)])
def test_sphere_is_closed_outward_and_has_the_analytic_area_and_volume(res, r, tol):
    x, y, z = _grid(res)
    c = (res - 1) / 2.0 + 0.13
    vol = (np.sqrt((x - c) ** 2 + (y - c + 0.2) ** 2 + (z - c - 0.1) ** 2) - r).astype(np.float32)
    v, f = MC.marching_cubes_voxel(vol)
    assert len(f) > 50
    assert_closed_and_oriented(f)
    vol_, area = signed_volume_and_area(v, f)
    assert vol_ > 0, "normals point inward"
    # (an inscribed polyhedron: both fall short of the sphere's by O((h / r)^2))
    assert 0 < 4 / 3 * np.pi * r ** 3 - vol_ < tol * 4 / 3 * np.pi * r ** 3
    assert 0 < 4 * np.pi * r ** 2 - area < tol * 4 * np.pi * r ** 2
    # every vertex lies on a lattice edge (two integer coordinates) and, for this nearly linear field between neighbours, close to the sphere
    frac = np.abs(v - np.round(v))
    assert np.all(np.sort(frac, axis=1)[:, 1] == 0)
    d = np.sqrt((v[:, 0] - c) ** 2 + (v[:, 1] - c + 0.2) ** 2 + (v[:, 2] - c - 0.1) ** 2)
    assert np.max(np.abs(d - r)) < 0.08


def test_vertex_set_is_the_set_of_sign_changing_lattice_edges():
    rng = np.random.default_rng(5)
    res = 8
    vol = rng.normal(size=(res, res, res)).astype(np.float32)
    v, f = MC.marching_cubes_voxel(vol)
    n_cross = 0
    for a in range(3):
        s = [slice(None)] * 3; t = [slice(None)] * 3
        s[a] = slice(0, res - 1); t[a] = slice(1, res)
        n_cross += int(np.count_nonzero((vol[tuple(s)] < 0) != (vol[tuple(t)] < 0)))
    assert len(v) == n_cross
    assert len(np.unique(f)) == len(v), "a vertex no face uses"
    # linear interpolation: the trilinear field evaluated along the edge vanishes at the vertex
    for p in v[:200]:
        a = int(np.argmax(np.abs(p - np.round(p)) > 0)) if np.any(p != np.round(p)) else 0
        lo = np.floor(p).astype(int); hi = lo.copy(); hi[a] = min(lo[a] + 1, res - 1)
        t = p[a] - lo[a]
        val = (1 - t) * vol[tuple(lo)] + t * vol[tuple(hi)]
        assert abs(val) < 1e-5 * max(1.0, abs(vol[tuple(lo)]), abs(vol[tuple(hi)]))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_signs_with_ambiguous_faces_still_close_the_surface(seed):
    rng = np.random.default_rng(seed)
    res = 10
    vol = np.ones((res, res, res), np.float32)
    vol[1:-1, 1:-1, 1:-1] = rng.choice([-1.0, 1.0], size=(res - 2,) * 3, p=[0.45, 0.55]).astype(np.float32) * rng.uniform(0.2, 1.0, size=(res - 2,) * 3).astype(np.float32)
    v, f = MC.marching_cubes_voxel(vol)
    assert len(f) > 200
    assert_closed_and_oriented(f)
    vol_, _ = signed_volume_and_area(v, f)
    assert vol_ > 0


def test_plane_and_the_world_map_and_the_skip_rule():
    res, vs = 8, 0.2
    x, y, z = _grid(res)
    plane = (z - 3.25).astype(np.float32)                       # inside below z = 3.25
    v, f = MC.marching_cubes_voxel(plane)
    assert np.allclose(v[:, 2], 3.25) and len(f) == 2 * (res - 1) ** 2
    p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(p1 - p0, p2 - p0)
    assert np.all(n[:, 2] > 0) and np.allclose(n[:, :2], 0)     # towards the positive values
    assert abs(np.linalg.norm(n, axis=1).sum() / 2 - (res - 1) ** 2) < 1e-4
    centres = np.array([[1.0, 2.0, 3.0], [5.0, 5.0, 5.0], [-1.0, 0.5, 2.0], [9.0, 9.0, 9.0]], np.float32)
    sdf = np.stack([plane, np.abs(plane) + 0.1, -plane, -np.abs(plane) - 0.1])[..., None]      # voxels 1 and 3 do not change sign
    V, F = MC.marching_cubes(centres, sdf, vs)
    assert len(V) == 2 * len(v) and len(F) == 2 * len(f)
    assert np.allclose(V[:len(v)], (v / (res - 1) - 0.5) * vs + centres[0], atol=1e-6)
    assert np.allclose(V[len(v):, 2], (3.25 / (res - 1) - 0.5) * vs + centres[2][2], atol=1e-6)
    assert F[len(f):].min() == len(v) and F.max() == len(V) - 1
    E, G = MC.marching_cubes(centres[1:2], sdf[1:2], vs)
    assert E.shape == (0, 3) and G.shape == (0, 3)
