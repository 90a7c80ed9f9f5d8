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


@pytest.mark.parametrize("pattern", ["shuffled", "far_apart", "duplicates", "vertex_then_voxel", "non_pow2_grid", "wraps_around"])
def test_insert_patterns_match_oracle(pattern):
    """the insert resumes a descent from the path of the previous one and skips points whose voxel already is a SURFACE leaf:
    orders and repeats that stress exactly that (node ids are creation-ordered, so any slip changes the export)"""
    rng = np.random.default_rng(11)
    v = _vox()
    grid = 256 * 256 * 4
    if pattern == "shuffled":
        batches = [v[rng.permutation(len(v))], v[rng.permutation(len(v))][:500] + np.array([1, 0, 0], np.int32)]
    elif pattern == "far_apart":
        batches = [rng.integers(0, grid, size=(1500, 3)).astype(np.int32), np.array([[0, 0, 0], [grid - 2, grid - 2, grid - 2], [0, 0, 0]], np.int32)]
    elif pattern == "duplicates":
        base = v[:40]
        batches = [np.repeat(base, 5, axis=0), base[::-1].copy(), np.concatenate([base, base + np.array([0, 1, 0], np.int32), base])]
    elif pattern == "vertex_then_voxel":
        # a coordinate first created as a corner vertex (FEATURE) of its -x neighbour, then inserted as a voxel right after:
        # same key twice in a row, the leaf level must still be revisited for the upgrade
        a = v[:200]
        batches = [np.stack([a, a + np.array([1, 0, 0], np.int32)], axis=1).reshape(-1, 3), a + np.array([1, 1, 1], np.int32)]
    elif pattern == "wraps_around":
        grid = 64                                                  # coordinates beyond the grid alias modulo its size
        c = rng.integers(0, 200, size=(600, 3)).astype(np.int32)
        batches = [c, c[:100] + 64, c[::-1].copy()]
    else:
        grid = 1000                                                # not a power of two: every descent starts at the root
        batches = [rng.integers(0, 512, size=(800, 3)).astype(np.int32)]
    a = Octree(); a.init(grid, 16, 0.2)
    b = O.Octree(); b.init(grid, 16, 0.2)
    for bt in batches:
        a.insert(bt); b.insert(bt)
        assert a.count_nodes() == b.count_nodes() and a.count_leaf_nodes() == b.count_leaf_nodes()
    va, ca, fa = [t.numpy() for t in a.get_centres_and_children()]
    vb, cb, fb = b.get_centres_and_children()
    assert np.array_equal(va, vb) and np.array_equal(ca, cb) and np.array_equal(fa, fb)


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
w = v[np.random.default_rng(3).permutation(len(v))[:6000]] + np.array([2, -1, 0], np.int32)    # incoherent order, partly new
r.insert(torch.from_numpy(w)); a.insert(w); o.insert(w)
ref = [t.numpy() for t in r.get_centres_and_children()]
for got in ([t.numpy() for t in a.get_centres_and_children()], list(o.get_centres_and_children())):
    assert all(np.array_equal(x, y) for x, y in zip(ref, got))
assert r.count_nodes() == a.count_nodes() and r.count_leaf_nodes() == a.count_leaf_nodes()
assert np.array_equal(r.get_leaf_voxels().numpy(), a.get_leaf_voxels().numpy()) and a.get_leaf_voxels().shape == (a.count_leaf_nodes(), 3)
assert np.array_equal(r.get_voxels().numpy(), a.get_voxels().numpy())
rng = np.random.default_rng(5)
for cand in (v[:3000], v[:3000] + np.array([1, 0, 0], np.int32), v[::9] + rng.integers(-3, 4, size=(len(v[::9]), 3)).astype(np.int32),
             rng.integers(0, 262144, size=(500, 3)).astype(np.int32), np.repeat(v[:5], 4, axis=0)):
    assert r.try_insert(torch.from_numpy(cand)) == a.try_insert(cand), (r.try_insert(torch.from_numpy(cand)), a.try_insert(cand))
far = v[:50] + np.array([2048, 0, 0], np.int32)           # keys equal to those of v[:50] in their low 32 bits: the reference counts
r.insert(torch.from_numpy(far)); a.insert(far)            # matches through a std::set<int>, so such pairs count once
both = np.concatenate([v[:50], far])
assert r.try_insert(torch.from_numpy(both)) == a.try_insert(both) and 0.4 < a.try_insert(both) < 0.6
assert r.count_nodes() == a.count_nodes()                                   # try_insert inserts nothing
from nerf_loam_amd.svo import encode
c = torch.from_numpy(rng.integers(0, 1 << 21, size=(200, 3)))
assert torch.equal(torch.ops.svo.encode(c), encode(c)) and encode(c).shape == (200, 1)
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


def test_children_block_layout_describes_the_tree_and_reproduces_the_oracle_hits():
    """pack_children_blocks (host logic of the traversal layout; CPU tensors): (1) walking the blocks from the pseudo root
    recovers voxel_structure exactly, children blocks are consecutive in octant order; (2) a plain DFS over the packed layout
    with centres recomputed from the lattice path finds the same leaves with the same t_min / t_max as the oracle's
    traversal of the reference layout (bit for bit)"""
    from nerf_loam_amd.pipeline import pack_children_blocks
    sc = H.build_oracle_scene(32, 24, 5)
    ms = sc["ms"]
    c, st = ms.centres, ms.structure
    ids, hdr = pack_children_blocks(torch.from_numpy(c), torch.from_numpy(st))
    ids, hdr = ids.numpy(), hdr.numpy()
    interior = (st[:, :8] > -1).any(1)
    assert ids[0, 0] == 0 and hdr[0, 0] == 1
    # the pseudo block's other id slots describe the single-child chain under the root (the traversal kernel runs its slab tests in
    # registers): [1] length, [2] / [3] octants (3 bits each), [4] the block the work-list starts with, [5..7] its lattice position
    n_chain, octs, end_blk, end_pos = int(ids[0, 1]), int(ids[0, 2]) | (int(ids[0, 3]) << 30), int(ids[0, 4]), ids[0, 5:8].tolist()
    b, pos, cs = 1, [0, 0, 0], int(st[0, 8]) // 2
    for lvl in range(n_chain):
        has = (hdr[b, 1] >> 8) & 255
        u = (octs >> (3 * lvl)) & 7
        assert has == 1 << u and cs > 1                                 # exactly one child owns a block: a chain level
        pos = [pos[0] + (cs if u & 1 else 0), pos[1] + (cs if u & 2 else 0), pos[2] + (cs if u & 4 else 0)]
        b, cs = int(hdr[b, 0]), cs // 2
    has_end = (hdr[b, 1] >> 8) & 255
    assert (b, pos) == (end_blk, end_pos) and (cs == 1 or bin(has_end).count("1") != 1) and n_chain >= 5     # the chain ends where the tree branches
    seen_nodes, stack = 0, [(1, 0)]                                   # (block, the node whose children it lists)
    while stack:
        b, node = stack.pop()
        seen_nodes += 1
        assert np.array_equal(ids[b], st[node, :8])
        exist, has = hdr[b, 1] & 255, (hdr[b, 1] >> 8) & 255
        assert exist == sum(1 << u for u in range(8) if st[node, u] > -1)
        assert has == sum(1 << u for u in range(8) if st[node, u] > -1 and interior[st[node, u]])
        assert (hdr[b, 0] == -1) == (has == 0)
        for u in range(8):
            if (has >> u) & 1:
                stack.append((hdr[b, 0] + bin(has & ((1 << u) - 1)).count("1"), st[node, u]))
    assert seen_nodes == int(interior.sum()) == len(ids) - 1

    # (2) DFS over the packed layout, fp32 arithmetic of the kernels (nl_slab), against the oracle
    f32 = np.float32
    vs, root_side = f32(0.2), int(st[0, 8])
    pts, cos = sc["points"], sc["cos"]
    pose = np.array([2000.0, 2000.0, 2000.0, 0, 0, 0], np.float32)
    o, d = O.ray_setup(S.unit_dirs(pts), O.rodrigues(pose[3:]), pose[:3])
    sel = np.arange(0, len(o), 37)
    oi, o0, o1 = O.svo_intersect(o[sel], d[sel], c, st, 0.2, 20)

    def slab(o_, inv, ctr, half):
        lo, hi = f32(0), f32(100000)
        for a in range(3):
            t0 = f32(f32(f32(ctr[a] - half) - o_[a]) * inv[a]); t1 = f32(f32(f32(ctr[a] + half) - o_[a]) * inv[a])
            if t1 < t0: t0, t1 = t1, t0
            if t1 < lo or t0 > hi: return None
            lo = max(lo, t0); hi = min(hi, t1)
            if lo > hi: return None
        return lo, hi

    for q, r in enumerate(sel[:60]):
        with np.errstate(divide="ignore"):
            inv = (f32(1) / d[r]).astype(f32)
        hits = []
        def centre(x, y, z, side):
            hs = f32(side) * f32(0.5)
            return [f32(f32(f32(x) + hs) * vs), f32(f32(f32(y) + hs) * vs), f32(f32(f32(z) + hs) * vs)], f32(f32(vs * f32(0.5)) * f32(side))
        def walk(b, x, y, z, cs):                                       # children of the node at (x,y,z), child side cs, DESCENDING octants
            exist, has = hdr[b, 1] & 255, (hdr[b, 1] >> 8) & 255
            for u in range(7, -1, -1):
                if not (exist >> u) & 1: continue
                cx, cy, cz = x + (cs if u & 1 else 0), y + (cs if u & 2 else 0), z + (cs if u & 4 else 0)
                ctr, half = centre(cx, cy, cz, cs)
                t = slab(o[r], inv, ctr, half)
                if t is None: continue
                if cs == 1:
                    hits.append((ids[b, u], t[0], t[1]))
                elif (has >> u) & 1:
                    walk(hdr[b, 0] + bin(has & ((1 << u) - 1)).count("1"), cx, cy, cz, cs // 2)
        ctr, half = centre(0, 0, 0, root_side)
        if slab(o[r], inv, ctr, half) is not None:
            walk(1, 0, 0, 0, root_side // 2)
        hits = hits[:20]
        n = int((oi[q] != -1).sum())
        assert n == len(hits)
        assert [h[0] for h in hits] == oi[q, :n].tolist()
        assert np.array_equal(np.array([h[1] for h in hits], np.float32), o0[q, :n]) and np.array_equal(np.array([h[2] for h in hits], np.float32), o1[q, :n])


def test_blocks_packed_by_the_octree_equal_the_host_layer_packing():
    """nl_octree_pack_blocks (C++, straight from the tree) == pipeline.pack_children_blocks on the exported structure rows, word for word: an empty tree, one
    voxel, a tree that branches at the root, a scan, and a map grown over six batches (FEATURE -> SURFACE upgrades, parents gaining children)"""
    from nerf_loam_amd.pipeline import pack_children_blocks

    def check(oc):
        c, s, f = oc.export_device_layout()
        want_ids, want_hdr = pack_children_blocks(torch.from_numpy(c), torch.from_numpy(s))
        ids, hdr = oc.pack_blocks()
        assert ids.dtype == np.int32 and hdr.dtype == np.int32
        assert np.array_equal(ids, want_ids.numpy()), "blk_ids differ"
        assert np.array_equal(hdr, want_hdr.numpy()), "blk_hdr differ"
        return len(ids)

    oc = Octree(); oc.init(256 * 256 * 4, 16, 0.2)
    assert check(oc) == 1                                           # the pseudo block alone
    oc.insert(torch.tensor([[10000, 10001, 10002]], dtype=torch.int32))
    assert check(oc) > 10                                           # a single-child chain down to the voxel's parents
    oc.insert(torch.tensor([[3, 200000, 7], [250000, 5, 9]], dtype=torch.int32))      # far corners of the lattice: the root branches, no chain
    check(oc)
    sc = H.build_oracle_scene(32, 24, 5)
    oc2 = Octree(); oc2.init(256 * 256 * 4, 16, 0.2)
    oc2.insert(torch.from_numpy(S.voxel_coords(sc["points"], np.eye(3, dtype=np.float32), S.scan_pose()[:3], 0.2)))
    check(oc2)
    rng = np.random.default_rng(5)
    oc3 = Octree(); oc3.init(256 * 256 * 4, 16, 0.2)
    base = np.array([10000, 10000, 10000])
    for batch in range(6):
        pts = base + rng.integers(-12, 12, size=(300, 3)) + np.array([batch * 5, 0, 0])
        if batch == 3:
            pts = np.concatenate([pts, prev + 1])
        prev = pts
        oc3.insert(torch.from_numpy(pts.astype(np.int32)))
        check(oc3)
    # a small grid (non power-of-two sizes take the root descent every time; 64 is the smallest the reference's tests use)
    oc4 = Octree(); oc4.init(64, 16, 0.2)
    oc4.insert(torch.from_numpy(rng.integers(0, 62, size=(200, 3)).astype(np.int32)))
    check(oc4)
