#!/usr/bin/env python3
"""Why tests/test_gpu_sparse_adam.py feeds both optimisers the SAME accumulators: two SEPARATE runs of the same calls - dense engine twice,
touched-rows engine twice, one against the other - and how many elements of the optimiser state differ, by how much.  The scatter adds one
global fp32 atomic per touched row and wave; the order the waves arrive in is not fixed, the sum changes in its last bit, the bf16 rounding of
the gradient flips for one element in ~10^5 and Adam turns that into one bf16 ulp (measured: 1 embedding element, 4 / 6 moment elements, the
decoder 8e-7 downstream; which pairs differ depends on the inputs and on what ran before)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                                     # noqa: E402
import test_gpu_sparse_adam as T                                 # noqa: E402
from oracle import oracle as O                                   # noqa: E402


def _run(P, sc, masks, emb_bits, dec_np, pose0, sparse, one_call, grow_after_call=0):
    """two calls of three iterations each (second call: the masks in reverse order + one unusable iteration in front); returns every
    piece of optimiser state"""
    ms = sc["ms"]
    n_rays = int(masks[0].sum())
    eng = P.SdfEngine(max_rays=n_rays, samples_per_ray_cap=64, max_frames=2, sparse_adam=sparse)
    dec = P.DecoderDevice(dec_np.W1, dec_np.b1, dec_np.W2, dec_np.b2, dec_np.W3, dec_np.b3)
    cfg = P.IterConfig(step_size=0.1)
    emb_t = torch.from_numpy(emb_bits.view(np.int16).copy()).cuda()
    out = {}
    for call in range(2):
        if call == 1 and grow_after_call:                                  # the map grew: rows appended (zeros), like a new frame's vertices
            emb_t = torch.cat([emb_t, torch.zeros(grow_after_call, 16, dtype=torch.int16, device="cuda")])
        m = P.MapDevice(ms.centres, ms.structure, ms.vertex_idx, ms.id2row, emb_t.cpu().numpy().view(np.uint16), ms.voxel_size)
        eng.set_poses(pose0[None], [1])
        eng.begin_call(m, dec)
        order = [0, 1, 2] if call == 0 else [None, 2, 1, 0]
        if one_call:
            eng.bind(m, dec, cfg, train_decoder=True, skip_mode=1)
        for it in order:
            if it is None:                                                 # rays looking away from the map: no hit -> the step is skipped on the device
                fr = O.select_rays(sc["points"], sc["cos"], pose0, masks[0])
                eng.set_rays(-fr.rays_d, fr.points, fr.cos)
            else:
                fr = O.select_rays(sc["points"], sc["cos"], pose0, masks[it])
                eng.set_rays(fr.rays_d, fr.points, fr.cos)
            if one_call:
                eng.run_bound()
            else:
                eng.forward_backward(m, dec, cfg, train_decoder=True)
                eng.optimiser_step(m, dec, cfg, skip_mode=1)
        torch.cuda.synchronize()
        steps, skipped, overflow = eng.call_status()
        assert (steps, skipped, overflow) == ((3, 0, False) if call == 0 else (3, 1, False))
        emb_t = m.emb.clone()
        out[call] = dict(emb=m.emb.cpu().numpy().copy(), m=eng.emb_m.cpu().numpy().copy(), v=eng.emb_v.cpu().numpy().copy(),
                         g=eng.g_emb.cpu().numpy().copy(), dec=dec.params.cpu().numpy().copy(), pose=eng.pose6[0].cpu().numpy().copy())
        if sparse:
            lst, cnt, flags = eng._touched
            n = int(cnt.item())
            rows = np.sort(lst[:n].cpu().numpy())
            assert len(np.unique(rows)) == n                                # every row listed once
            live = np.nonzero((out[call]["m"] != 0).any(1) | (out[call]["v"] != 0).any(1))[0]
            assert np.isin(live, rows).all()                                # every row that carries moments is listed ...
            bits = np.unpackbits(flags.cpu().numpy().view(np.uint8), bitorder="little")
            assert np.array_equal(np.nonzero(bits)[0], rows)                # ... and flagged; nothing else is
            out[call]["touched"] = n
    return out


g, sc, masks, emb, dec_np, P = T._scene(os.path.join(ROOT, "tests", "golden"), 0)
pose0 = g["poses0"][0].copy()
runs = {}
for name, sparse in (("dense_a", False), ("dense_b", False), ("sparse_a", True), ("sparse_b", True)):
    runs[name] = _run(P, sc, masks, emb, dec_np, pose0, sparse=sparse, one_call=False)
for a, b in (("dense_a", "dense_b"), ("sparse_a", "sparse_b"), ("dense_a", "sparse_a")):
    for call in (0, 1):
        for k in ("emb", "m", "v", "dec", "pose"):
            x, y = runs[a][call][k], runs[b][call][k]
            if x.dtype in (np.int16, np.uint16):
                xf, yf = O.bf16_to_f32(x.view(np.uint16)), O.bf16_to_f32(y.view(np.uint16))
            else:
                xf, yf = x, y
            nd = int((x != y).sum())
            print(f"{a} vs {b} call {call} {k:4s}: {nd} of {x.size} differ" + (f", max |d| {np.abs(xf - yf).max():.3e}, rows {np.unique(np.nonzero(x != y)[0])[:8]}" if nd else ""))
