"""How many (workgroup, embedding row) pairs does the embedding-gradient scatter flush to global memory for different ways of
assigning samples to workgroups?  (bench workload, one iteration; GPU only; analysis aid for k_trilinear_bwd)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P
L.require_gpu()
w = bench.build_workload(torch.device("cuda"))
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
st = eng.stats(); Pn = st["P"]
vox = eng.s_vox[:Pn].long(); ray = eng.s_ray[:Pn].long()
rows = w["map"].vertex_rows[vox].long()                       # [P, 8]
E = int(w["map"].n_rows)
print("samples", Pn, "rows", E, "distinct rows touched", int(torch.unique(rows).numel()))
def pairs(group):
    key = (group[:, None] * E + rows).reshape(-1)
    return int(torch.unique(key).numel())
idx = torch.arange(Pn, device="cuda")
for chunk in (1024, 2048, 4096, 8192):
    print(f"1-D chunks of {chunk} samples: groups {Pn // chunk + 1}, (group,row) pairs {pairs(idx // chunk)}")
beam, az = ray // 2048, ray % 2048
for bb, ab in ((2, 64), (4, 32), (4, 64), (8, 32), (8, 64), (16, 32), (8, 128), (16, 64), (64, 8), (64, 16), (32, 16)):
    g = (beam // bb) * (2048 // ab) + az // ab
    ng = int(torch.unique(g).numel())
    cnt = torch.bincount(g)
    # rows per group (LDS table size needed)
    key = torch.unique(g[:, None] * E + rows)
    rpg = torch.bincount(key // E)
    print(f"2-D tiles {bb} beams x {ab} az: groups {ng}, samples/group max {int(cnt.max())}, (group,row) pairs {key.numel()}, rows/group max {int(rpg.max())} mean {float(rpg[rpg>0].float().mean()):.0f}")
