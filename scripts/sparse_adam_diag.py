#!/usr/bin/env python3
"""diagnostic for tests/test_gpu_sparse_adam.py: dense vs touched-rows engine, each run twice - how many elements differ and by how much"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_sparse_adam as T                                 # noqa: E402
from oracle import oracle as O                                   # noqa: E402

g, sc, masks, emb, dec_np, P = T._scene(os.path.join(ROOT, "tests", "golden"), 0)
pose0 = g["poses0"][0].copy()
runs = {}
for name, sparse in (("dense_a", False), ("dense_b", False), ("sparse_a", True), ("sparse_b", True)):
    runs[name] = T._run(P, sc, masks, emb, dec_np, pose0, sparse=sparse, one_call=False)
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
