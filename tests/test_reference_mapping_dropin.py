"""CPU, build container only (needs /root/reference; skipped elsewhere): the reference's OWN src/mapping.py - Mapping.__init__,
create_voxels, get_embeddings, update_grid_features, unmodified - runs on `torch.classes.svo.Octree` / `torch.ops.svo.encode` as
registered by nerf_loam_amd/libnl_svo_torch.so (TORCH_LIBRARY(svo) over the C ABI) with ONLY the path of its
torch.classes.load_library line changed, and produces exactly the map_states it produces on the reference's C++ octree
(oracle/_ref/svo_ref.so).  SURVEY 8 b2: bindings.cpp:4-31, call sites mapping.py:19-20, 81-82, 283-339."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NL_REFERENCE_ROOT", "/root/reference")


def _run(which, out):
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_mapping_probe.py"), which, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present (GPU box)")
def test_reference_mapping_runs_on_the_registered_svo_names(tmp_path):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "svo_ref.so")):
        pytest.skip("oracle/_ref/svo_ref.so not built")
    a, b = str(tmp_path / "ours.npz"), str(tmp_path / "ref.npz")
    _run("ours", a)
    _run("ref", b)
    A, B = np.load(a), np.load(b)
    assert sorted(A.files) == sorted(B.files)
    for k in A.files:
        if k.endswith("_id_table"):
            # mapping.py:317 writes arange(...) through an index tensor that lists a vertex id once per OCCURRENCE: which of the
            # duplicates' rows an id ends up with is torch's index_put order (it differs between two runs of the reference itself).
            # Same ids assigned, rows inside the table the reference allocated - the rows of the losers are never read
            E = int(A[k.replace("_id_table", "_emb_shape")][0])
            assert np.array_equal(A[k] >= 0, B[k] >= 0) and A[k].max() < E and B[k].max() < E, k
            continue
        assert np.array_equal(A[k], B[k]), k                      # bit for bit: node ids, centres, structure, encode, DFS orders
    assert A["f1_counts"][0] > A["f0_counts"][0] and A["f1_emb_shape"][0] > A["f0_emb_shape"][0]     # the second frame grew the map


def test_torchscript_octree_pickles_like_the_reference_binding(tmp_path):
    """def_pickle state of bindings.cpp:23-31: (size, feat_dim, voxel_size, inserted tensors), rebuilt by replay"""
    import io
    import torch
    from nerf_loam_amd import svo
    svo.load_torch_library()
    t = torch.classes.svo.Octree()
    t.init(256 * 256 * 4, 16, 0.2)
    t.insert(torch.tensor([[10000, 10000, 10000], [10001, 10000, 10000]], dtype=torch.int32))
    n, leaves = t.count_nodes(), t.count_leaf_nodes()
    buf = io.BytesIO()
    torch.save(t, buf)
    buf.seek(0)
    u = torch.load(buf, weights_only=False)
    assert (u.count_nodes(), u.count_leaf_nodes()) == (n, leaves) and leaves == 2
    va, ca, fa = t.get_centres_and_children()
    vb, cb, fb = u.get_centres_and_children()
    assert torch.equal(va, vb) and torch.equal(ca, cb) and torch.equal(fa, fb)
    assert torch.classes.svo.Octant() is not None
    with pytest.raises(RuntimeError):
        torch.classes.svo.Octree().count_nodes()                   # not initialised
