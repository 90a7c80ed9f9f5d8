"""Host side of SURVEY 8 (b2): the native octree (csrc/nl_octree.cpp) next to the REFERENCE's C++ octree (oracle/_ref/svo_ref.so,
built by oracle/build_ref.sh from /root/reference - build container only) on the per-frame work of Mapping.create_voxels +
update_grid_features: insert the frame's voxels, then export the tree.  Six synthetic 64x2048 scans, the sensor moving 1.5 m per
frame.  CPU only; prints ms per frame.  The reference re-exports the whole tree every frame (mapping.py:314-327); the product
exports the changed rows (nl_octree_export_delta)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_loam_amd import synthetic as S
from oracle import oracle as O

def frames(n=6):
    out = []
    for f in range(n):
        pts, _ = S.synthetic_scan(64, 2048, 100 + f, range_noise=0.02)
        pose = S.scan_pose(tx=1.5 * f, ty=0.2 * f)
        out.append(S.voxel_coords(pts, O.rodrigues(pose[3:]), pose[:3], 0.2))
    return out

def ms(t0):
    return (time.perf_counter() - t0) * 1e3

def run_product(fr):
    from nerf_loam_amd.svo import Octree
    t = Octree(); t.init(256 * 256 * 4, 16, 0.2)
    for i, v in enumerate(fr):
        t0 = time.perf_counter(); t.insert(v); ti = ms(t0)
        t0 = time.perf_counter(); ids, c, s, f = t.export_delta(); td = ms(t0)
        t0 = time.perf_counter(); t.export_device_layout(); tf = ms(t0)
        print(f"product   frame {i}: {len(v):6d} voxels in, {t.count_nodes():7d} nodes | insert {ti:7.2f} ms | delta export {td:6.2f} ms "
              f"({len(ids)} rows) | full export {tf:6.2f} ms")

def run_reference(fr):
    import torch
    torch.classes.load_library(os.path.join(ROOT, "oracle", "_ref", "svo_ref.so"))
    t = torch.classes.svo.Octree(); t.init(256 * 256 * 4, 16, 0.2)
    for i, v in enumerate(fr):
        tv = torch.from_numpy(v)
        t0 = time.perf_counter(); t.insert(tv); ti = ms(t0)
        t0 = time.perf_counter(); out = t.get_centres_and_children(); tf = ms(t0)
        print(f"reference frame {i}: {len(v):6d} voxels in, {t.count_nodes():7d} nodes | insert {ti:7.2f} ms | full export {tf:6.2f} ms")

if __name__ == "__main__":
    fr = frames()
    if len(sys.argv) > 1 and sys.argv[1] == "reference":
        run_reference(fr)
    else:
        run_product(fr)
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "svo_ref.so")):
            subprocess.run([sys.executable, __file__, "reference"], check=False)      # own process: the reference's global node counter
