"""the 150-scan map legs of bench.py alone (mapping 2048 x 1, BA 4096 x 4, full scan - with the replicated gradient accumulators and with a single array -,
the tracker steps): ms per iteration / step.  GPU only."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L
L.require_gpu()
dev = torch.device("cuda", 0)
w = bench.build_workload(dev)
lm = bench.large_map_bench(w, dev)
for k, v in lm.items():
    if isinstance(v, dict) and "ms_per_iter" in v:
        print(f"{k:48s} {v['ms_per_iter']:.4f} ms / iteration   optimiser {v['optimiser_ms']:.4f}  begin_call {v['begin_call_ms']:.4f}  copies {v.get('emb_grad_copies', '-')}")
for k, v in lm.get("track_2048", {}).items():
    print(f"track_2048 {k:37s} {v['ms_per_step']:.4f} ms / step   {v['samples_per_hit_ray']:.1f} samples per hit ray")
print("parity", lm["parity_vs_oracle"])
