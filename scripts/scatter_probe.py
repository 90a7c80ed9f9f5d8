"""k_trilinear_bwd on the bench workload: HIP-event time of the kernel alone (embedding scatter + pose partials, each alone,
both) and per-phase s_memtime stamps of every workgroup (GPU only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P, ops
L.require_gpu()
w = bench.build_workload(torch.device("cuda"))
eng = P.SdfEngine(max_rays=len(w["points"]), samples_per_ray_cap=48)
eng.set_rays(w["dirs"], w["points"], w["cos"]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(); eng.begin_call(w["map"], w["dec"])
for _ in range(2):
    eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=True)
m = w["map"]
def run(emb, pose, blocks):
    ops.trilinear_bwd(eng.loss_scalars, eng.s_vox, eng.s_depth, eng.s_ray, eng.rays_d_world, eng.rays_d_sensor, eng.frame_id,
                      eng.poses12, eng.F, m.centres, m.vertex_rows, m.emb, m.voxel_size, eng.dX,
                      eng.g_emb if emb else None, eng.g_pose if pose else None, blocks)
for blocks in (eng.field_blocks, 2 * eng.field_blocks):
    for emb, pose in ((1, 1), (1, 0), (0, 1)):
        run(emb, pose, blocks); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(emb, pose, blocks)
        e1.record(); torch.cuda.synchronize()
        print(f"blocks {blocks} emb_grad {emb} pose_grad {pose}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us")
nb = 2 * eng.field_blocks
dbg = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
L.lib().nl_field_set_debug_buffer(L.ptr(dbg))
run(1, 1, nb)
torch.cuda.synchronize()
L.lib().nl_field_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(nb, 8)
d = d[(d[:, [0, 1, 2, 3, 5]] > 0).all(1)]
names = ["init->barrier", "sample loop", "last run flush", "table flush + touched rows"]
ph = np.diff(d[:, [0, 1, 2, 3, 5]], axis=1)
print("workgroups", len(d))
for n, v, mx in zip(names, ph.mean(0), ph.max(0)):
    print(f"  {n:28s} mean {v:10.0f}  max {mx:10.0f}")
print("  per workgroup total mean", (d[:, 5] - d[:, 0]).mean(), "cycles; kernel span", (d[:, 5].max() - d[:, 0].min()), "cycles")
