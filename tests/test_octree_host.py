"""CPU: the product's native host octree (csrc/nl_octree.cpp via nerf_loam_amd.svo.Octree) against
the oracle restatement and - when oracle/_ref/svo_ref.so has been built (build container) - against
the reference C++ itself.  Integer outputs: bit-exact."""
import os
import pickle
import subprocess
import sys

import numpy as np
import torch
import pytest

import helpers as H
from nerf_loam_amd import synthetic as S
from nerf_loam_amd.svo import Octree
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vox(seed=777, nb=64, na=48):
    pts, _ = H.scene_points(nb, na, seed)
    pose = S.scan_pose()
    return S.voxel_coords(pts, O.rodrigues(pose[3:]), pose[:3], 0.2)


def _both(vox_batches):
    a = Octree(); a.init(256 * 256 * 4, 16, 0.2)
    b = O.Octree(); b.init(256 * 256 * 4, 16, 0.2)
    for v in vox_batches:
        a.insert(v); b.insert(v)
    return a, b


def test_matches_oracle_single_and_incremental():
    v = _vox()
    for batches in ([v], [v[:1000], v[500:], v[::7] + np.array([3, -2, 1], np.int32)]):
        a, b = _both(batches)
        assert a.count_nodes() == b.count_nodes() and a.count_leaf_nodes() == b.count_leaf_nodes()
        va, ca, fa = [t.numpy() for t in a.get_centres_and_children()]
        vb, cb, fb = b.get_centres_and_children()
        assert np.array_equal(va, vb) and np.array_equal(ca, cb) and np.array_equal(fa, fb)
        c, s, f = a.export_device_layout()
        c2, s2 = O.grid_features(vb, cb, 0.2)
        assert np.array_equal(c, c2) and np.array_equal(s, s2) and np.array_equal(f, fb)


def test_invariants_and_edge_cases():
    a = Octree(); a.init(256 * 256 * 4, 16, 0.2)
    assert a.count_nodes() == 1 and a.count_leaf_nodes() == 0
    a.insert(np.zeros((0, 3), np.int32))                       # empty insert
    assert a.count_nodes() == 1
    a.insert(np.array([[10000, 10000, 10000], [10001, 10000, 10000]], np.int32))
    v, c, f = [t.numpy() for t in a.get_centres_and_children()]
    assert a.count_nodes() == 31 and a.count_leaf_nodes() == 2   # SURVEY Appendix C smoke values
    surf = f[:, 0] >= 0
    assert surf.sum() == 2 and (f[surf] >= 0).all()              # every SURFACE leaf has 8 vertex ids
    assert np.array_equal(f[surf][0], np.arange(18, 26))
    assert np.array_equal(f[surf][1], [22, 23, 24, 25, 27, 28, 29, 30])
    assert a.has_voxel([10000, 10000, 10000]) and not a.has_voxel([5, 5, 5])
    a.insert(np.array([[10000, 10000, 10000]], np.int32))        # duplicate insert creates nothing
    assert a.count_nodes() == 31
    # children side halves per level along any root->leaf path
    node, side = 0, v[0, 3]
    while True:
        kids = c[node][c[node] >= 0]
        if len(kids) == 0:
            break
        node = int(kids[0]); assert v[node, 3] == side / 2; side = v[node, 3]
    with pytest.raises(ValueError):
        a.insert(np.zeros((4, 2), np.int32))
    with pytest.raises(RuntimeError):
        Octree().count_nodes()


def test_pickle_replays_inserts():
    v = _vox(nb=16, na=32)
    a = Octree(); a.init(256 * 256 * 4, 16, 0.2); a.insert(v[:200]); a.insert(v[200:])
    b = pickle.loads(pickle.dumps(a))
    for x, y in zip(a.get_centres_and_children(), b.get_centres_and_children()):
        assert np.array_equal(x.numpy(), y.numpy())


def test_two_instances_are_independent():
    v = _vox(nb=16, na=32)
    a = Octree(); a.init(256 * 256 * 4, 16, 0.2); a.insert(v)
    b = Octree(); b.init(256 * 256 * 4, 16, 0.2); b.insert(v)          # the reference corrupts here
    for x, y in zip(a.get_centres_and_children(), b.get_centres_and_children()):
        assert np.array_equal(x.numpy(), y.numpy())


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "svo_ref.so")),
                    reason="reference svo not built (oracle/build_ref.sh needs /root/reference)")
def test_matches_reference_cpp():
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import helpers as H
from nerf_loam_amd import synthetic as S
from nerf_loam_amd.svo import Octree
from oracle import oracle as O
torch.classes.load_library({os.path.join(ROOT, 'oracle', '_ref', 'svo_ref.so')!r})
pts, _ = S.synthetic_scan(64, 256, 5, range_noise=0.02, sector=(0.3, 0.3 + 256 / 2048))
pose = S.scan_pose()
v = S.voxel_coords(pts, O.rodrigues(pose[3:]), pose[:3], 0.2)
r = torch.classes.svo.Octree(); r.init(256*256*4, 16, 0.2); r.insert(torch.from_numpy(v[:9000])); r.insert(torch.from_numpy(v[7000:]))
a = Octree(); a.init(256*256*4, 16, 0.2); a.insert(v[:9000]); a.insert(v[7000:])
o = O.Octree(); o.init(256*256*4, 16, 0.2); o.insert(v[:9000]); o.insert(v[7000:])
ref = [t.numpy() for t in r.get_centres_and_children()]
for got in ([t.numpy() for t in a.get_centres_and_children()], list(o.get_centres_and_children())):
    assert all(np.array_equal(x, y) for x, y in zip(ref, got))
assert r.count_nodes() == a.count_nodes() and r.count_leaf_nodes() == a.count_leaf_nodes()
print('OK', r.count_nodes())
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)   # own process: reference's global node counter
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


def test_incremental_export_reproduces_the_full_export():
    """deltas (changed rows only) applied in order == full export, after every insert batch, including
    FEATURE -> SURFACE upgrades and parents gaining children"""
    from nerf_loam_amd.svo import Octree
    rng = np.random.default_rng(5)
    oc = Octree(); oc.init(256 * 256 * 4, 16, 0.2)
    C = np.zeros((0, 3), np.float32); S = np.zeros((0, 9), np.int32); F = np.zeros((0, 8), np.int32)
    total_delta = 0
    base = np.array([10000, 10000, 10000])
    for batch in range(6):
        pts = base + rng.integers(-12, 12, size=(300, 3)) + np.array([batch * 5, 0, 0])
        if batch == 3:                                           # re-insert vertex-only leaves of earlier voxels: upgrades
            pts = np.concatenate([pts, prev + 1])
        prev = pts
        oc.insert(torch.from_numpy(pts.astype(np.int32)))
        ids, c, s, f = oc.export_delta()
        n = oc.count_nodes()
        grow = n - len(C)
        C = np.concatenate([C, np.zeros((grow, 3), np.float32)]); S = np.concatenate([S, np.zeros((grow, 9), np.int32)])
        F = np.concatenate([F, np.zeros((grow, 8), np.int32)])
        assert len(np.unique(ids)) == len(ids) and set(range(n - grow, n)) <= set(ids.tolist())      # every new node is in the delta
        C[ids] = c; S[ids] = s; F[ids] = f
        fc, fs, ff = oc.export_device_layout()
        assert np.array_equal(C, fc) and np.array_equal(S, fs) and np.array_equal(F, ff)
        total_delta += len(ids)
        if batch > 0:
            assert len(ids) < n                                  # later deltas are partial
    assert len(oc.export_delta()[0]) == 0                        # nothing changed since
