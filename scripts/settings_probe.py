"""the kitti / ncd settings legs and the 150-scan map legs of bench.py alone (A/B of library switches through the environment): ms per iteration"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L
L.require_gpu()
dev = torch.device("cuda", 0)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("NL_")) or "defaults"
for rep in range(2):
    s = bench.settings_bench(dev, with_parity=False)
    print(tag, "settings:", {k: round(v["ms_per_iter"], 3) for k, v in s.items()}, flush=True)
if "--large-map" in sys.argv:
    w = bench.build_workload(dev)
    lm = bench.large_map_bench(w, dev)
    print(tag, "large_map:", {k: round(v["ms_per_iter"], 4) for k, v in lm.items() if isinstance(v, dict) and "ms_per_iter" in v}, flush=True)
