"""The fused decoder kernel at the live shapes (2048 rays, step 0.04: ~32.7 k samples = two 64-sample tiles per CU): phase stamps of
workgroup 0's two tiles and the launch time back to back - how much of the launch is the tiles themselves.  GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_loam_amd import _lib as L, pipeline as P
L.require_gpu()
w = bench.build_workload(torch.device("cuda"))
rng = np.random.default_rng(3)
sel = np.sort(rng.choice(len(w["points"]), 2048, replace=False))
eng = P.SdfEngine(max_rays=2048, samples_per_ray_cap=96)
eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel]); eng.set_poses(w["pose"][None], [1])
cfg = P.IterConfig(step_size=0.04); eng.begin_call(w["map"], None)
dbg = torch.zeros(256 + 8 * 1024, dtype=torch.int64, device="cuda")     # (+ k_decoder2's per-workgroup records)
for _ in range(3):
    eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=False, want_emb_grad=False, want_pose_grad=True)
L.lib().nl_decoder_set_debug_buffer(L.ptr(dbg))
eng.forward_backward(w["map"], w["dec"], cfg, train_decoder=False, want_emb_grad=False, want_pose_grad=True)
torch.cuda.synchronize()
L.lib().nl_decoder_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(16, 16)[:, :11]
names = ["A:loadX", "B:H1", "C:loop", "C:epi", "D:loss", "E:dH2", "F:loop", "F:epi", "H:dH1", "I:L1bwd"]
print("P", eng.stats()["P"])
for t in range(3):
    if d[t, 0] == 0: break
    print("tile", t, "phases", np.diff(d[t]).tolist(), "total", int(d[t, 10] - d[t, 0]))
print("tile0 start -> tile1 end:", int(d[1, 10] - d[0, 0]), "cycles")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
from nerf_loam_amd import ops
st = eng
torch.cuda.synchronize()
# time the decoder launch alone
import time
args = (st.loss_scalars, st.X, w["dec"].params, w["dec"].W2T, st.s_ray, st.s_depth, st.cos_gt, st.gt_dist, st.sdf, st.dsdf, st.dX, st.partials, st.relu2_mask, st.n_slabs, 0, st.counters)
for _ in range(5): ops.decoder_fwd_bwd(*args)
torch.cuda.synchronize(); ev0.record()
for _ in range(50): ops.decoder_fwd_bwd(*args)
ev1.record(); torch.cuda.synchronize()
print("decoder launch (back to back, 50x): %.2f us each" % (ev0.elapsed_time(ev1) / 50 * 1e3))
