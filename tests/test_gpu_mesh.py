"""HIP marching cubes (csrc/nl_mesh.hip, nerf_loam_amd/mesh_util.py) against the oracle (oracle/mc_oracle.py) - the replacement of the reference's
per-voxel skimage call, src/utils/mesh_util.py:145-169.  Integer / index work (faces, vertex order) bit-exact; vertex coordinates bit-exact too (the same
fp32 operations in the same order).  What the oracle itself is pinned to: tests/test_mc_oracle.py."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _volumes(rng, n, res):
    """a mix: smooth blobs, planes, noise with ambiguous faces, and grids that do not change sign (skipped)"""
    g = np.arange(res, dtype=np.float32)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    out = []
    for i in range(n):
        kind = i % 5
        if kind == 0:
            c = rng.uniform(0.0, res - 1.0, 3)
            out.append(np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - rng.uniform(0.4, max(0.6, res / 2.5)))
        elif kind == 1:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            out.append((x * nrm[0] + y * nrm[1] + z * nrm[2]) - rng.uniform(0.2, res - 1.2) * nrm.sum())
        elif kind == 2:
            out.append(rng.normal(size=(res, res, res)))
        elif kind == 3:
            out.append(np.abs(rng.normal(size=(res, res, res))) + 0.01)              # all positive: skipped
        else:
            out.append(-np.abs(rng.normal(size=(res, res, res))) - 0.01)             # all negative: skipped
    return np.stack(out).astype(np.float32)


@pytest.mark.parametrize("res,n", [(8, 40), (4, 33), (5, 17), (16, 6), (2, 9)])
def test_marching_cubes_equals_the_oracle(res, n):
    from nerf_loam_amd import ops
    from oracle import mc_oracle as MC
    rng = np.random.default_rng(100 + res)
    sdf = _volumes(rng, n, res)
    centres = rng.uniform(-50, 50, size=(n, 3)).astype(np.float32)
    vs = 0.2
    want_v, want_f = MC.marching_cubes(centres, sdf, vs)
    v, f = ops.marching_cubes(torch.from_numpy(sdf).cuda(), torch.from_numpy(centres).cuda(), vs)
    torch.cuda.synchronize()
    assert v.shape == want_v.shape and f.shape == want_f.shape and len(want_f) > 0
    assert np.array_equal(f.cpu().numpy(), want_f), "faces differ"
    assert np.array_equal(v.cpu().numpy().view(np.uint32), want_v.view(np.uint32)), "vertex coordinates differ"
    # twice the same launch: the same arrays (offsets from scans, no atomics)
    v2, f2 = ops.marching_cubes(torch.from_numpy(sdf).cuda(), torch.from_numpy(centres).cuda(), vs)
    assert torch.equal(v, v2) and torch.equal(f, f2)


def test_empty_inputs_and_grids_without_a_surface():
    from nerf_loam_amd import ops
    v, f = ops.marching_cubes(torch.zeros(0, 8, 8, 8, device="cuda"), torch.zeros(0, 3, device="cuda"), 0.2)
    assert v.shape == (0, 3) and f.shape == (0, 3) and f.dtype == torch.int32
    v, f = ops.marching_cubes(torch.ones(5, 8, 8, 8, device="cuda"), torch.zeros(5, 3, device="cuda"), 0.2)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    with pytest.raises(Exception):
        ops.marching_cubes(torch.ones(2, 17, 17, 17, device="cuda"), torch.zeros(2, 3, device="cuda"), 0.2)      # res > 16: refused, not truncated
    with pytest.raises(Exception):
        ops.marching_cubes(torch.ones(2, 8, 8, 8), torch.zeros(2, 3), 0.2)                                         # host tensors: no CPU path


def test_full_map_counts_match_a_torch_census():
    """20 000 voxels at res 8 (the size mesh extraction runs at): properties that need no oracle loop - the vertex count is the number of sign-changing
    lattice edges, the triangle count the table's sum over the cell configurations, every face indexes vertices of its own voxel"""
    from nerf_loam_amd import ops
    from oracle import mc_oracle as MC
    n, res = 20000, 8
    gen = torch.Generator(device="cuda").manual_seed(3)
    base = torch.randn(n, 1, 1, 1, device="cuda", generator=gen) * 2.0
    sdf = (torch.randn(n, res, res, res, device="cuda", generator=gen) * 0.6 + base).contiguous()
    centres = torch.rand(n, 3, device="cuda", generator=gen) * 100
    v, f = ops.marching_cubes(sdf, centres, 0.2)
    neg = sdf < 0
    live = ~((sdf.amin(dim=(1, 2, 3)) > 0) | (sdf.amax(dim=(1, 2, 3)) < 0))
    cross = ((neg[:, 1:] != neg[:, :-1]).sum(dim=(1, 2, 3)) + (neg[:, :, 1:] != neg[:, :, :-1]).sum(dim=(1, 2, 3)) + (neg[:, :, :, 1:] != neg[:, :, :, :-1]).sum(dim=(1, 2, 3)))
    cfg = torch.zeros(n, res - 1, res - 1, res - 1, dtype=torch.long, device="cuda")
    for c in range(8):
        cfg |= neg[:, (c & 1):res - 1 + (c & 1), ((c >> 1) & 1):res - 1 + ((c >> 1) & 1), ((c >> 2) & 1):res - 1 + ((c >> 2) & 1)].long() << c
    ntri = torch.tensor([len(t) for t in MC.TRIS], device="cuda")[cfg].sum(dim=(1, 2, 3))
    assert int(live.sum()) > 1000 and int((~live).sum()) > 1000
    assert v.shape[0] == int((cross * live).sum()) and f.shape[0] == int((ntri * live).sum())
    voff = torch.cumsum(cross * live, 0)
    tri_owner = torch.repeat_interleave(torch.arange(n, device="cuda"), (ntri * live))
    lo = (voff - cross * live)[tri_owner]
    assert bool(((f >= lo[:, None]) & (f < voff[tri_owner][:, None])).all())
    # world map: a voxel's vertices lie inside its cube
    vert_owner = torch.repeat_interleave(torch.arange(n, device="cuda"), (cross * live))
    assert float((v - centres[vert_owner]).abs().max()) <= 0.1 + 1e-5


def test_mesh_extractor_mirrors_the_reference_call(golden_dir):
    """MeshExtractor.create_mesh / Mapping.extract_mesh's call shape (mesh_util.py:80-142 with clean_mseh = require_color = False): get_scores on the device,
    marching cubes on the device, vertices + offset; equal to the oracle's extraction from the same grid"""
    import helpers as H
    from nerf_loam_amd.decoder import Decoder
    from nerf_loam_amd.mesh_util import MeshExtractor
    from nerf_loam_amd.render_helpers import get_scores
    from oracle import mc_oracle as MC
    from oracle import oracle as O
    sc = H.build_oracle_scene(64, 16, 21)
    ms = sc["ms"]
    surf = np.nonzero(ms.vertex_idx[:, 0] >= 0)[0][:400]
    d0 = O.decoder_init(21)
    dec = Decoder().cuda()
    dec.load_flat(torch.from_numpy(np.concatenate([a.reshape(-1) for a in (d0.W1, d0.b1, d0.W2, d0.b2, d0.W3, d0.b3)])).cuda())
    emb = torch.from_numpy(H.init_embeddings(len(ms.emb), 21)).to(torch.bfloat16).cuda()
    states = {"voxel_vertex_idx": torch.from_numpy(ms.vertex_idx[surf]), "voxel_center_xyz": torch.from_numpy(ms.centres[surf]),
              "voxel_structure": torch.from_numpy(ms.structure[surf]), "voxel_vertex_emb": emb, "voxel_id2embedding_id": torch.from_numpy(ms.id2row)}
    res = 8
    grid = get_scores(dec, states, 0.2, bits=res)
    med = float(grid.median())                                     # a decoder whose zero set crosses the voxels: the output bias moves to the field's median
    with torch.no_grad():
        dec.sdf_out.bias.sub_(med)
    grid = get_scores(dec, states, 0.2, bits=res)
    assert float(grid.min()) < 0 < float(grid.max())
    args = types.SimpleNamespace(mapper_specs={"voxel_size": 0.2})
    mesher = MeshExtractor(args)
    mesh = mesher.create_mesh(dec, states, 0.2, states["voxel_center_xyz"], clean_mseh=False, require_color=False, offset=-2000, res=res)
    want_v, want_f = MC.marching_cubes(ms.centres[surf], grid.numpy(), 0.2)
    verts, tris = np.asarray(mesh.vertices), np.asarray(mesh.triangles)
    assert len(want_f) > 100 and tris.shape == want_f.shape
    assert np.array_equal(tris, want_f)
    assert np.allclose(verts, want_v + np.float32(-2000), atol=0, rtol=0)
    v2, f2 = mesher.marching_cubes(states["voxel_center_xyz"], grid)             # host inputs, like the reference passes them
    assert np.array_equal(v2, want_v) and np.array_equal(f2, want_f)
    with pytest.raises(NotImplementedError):
        mesher.create_mesh(dec, states, 0.2, states["voxel_center_xyz"], clean_mseh=True)
