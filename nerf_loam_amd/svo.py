"""`svo.Octree` operator surface on the native host octree (csrc/nl_octree.cpp).

Mirrors torch.classes.svo.Octree (/root/reference/third_party/sparse_octree/src/bindings.cpp:4-31):
same method names, argument meaning and returned tensors (CPU torch tensors, same dtypes/shapes),
same pickle state (size, feat_dim, voxel_size, list of inserted tensors).  Differences, on purpose:
the node counter is per instance (the reference's is process-global) and malformed
input raises instead of printing to stdout."""
import ctypes

import numpy as np
import torch

from . import _lib as L


class Octree:
    def __init__(self):
        self._h = None
        self.all_pts = []
        self.size_ = self.feat_dim_ = 0
        self.voxel_size_ = 0.0

    def init(self, grid_dim, feat_dim, voxel_size):
        if self._h:
            L.lib().nl_octree_destroy(self._h)
        self._h = ctypes.c_void_p(L.lib().nl_octree_create(int(grid_dim)))
        if not self._h:
            raise ValueError("Octree.init: grid_dim must be > 1")
        self.size_, self.feat_dim_, self.voxel_size_ = int(grid_dim), int(feat_dim), float(voxel_size)
        self.all_pts = []

    def _need(self):
        if not self._h:
            raise RuntimeError("Octree not initialized!")

    def insert(self, pts):
        self._need()
        t = torch.as_tensor(pts)
        if t.dim() != 2 or t.shape[1] != 3:
            raise ValueError(f"Point dimensions mismatch: inputs are {tuple(t.shape)} expect [M,3]")
        keep = np.array(t.cpu().numpy(), copy=True)          # private copy for the pickle state (numpy: no thread-pool wake-up per frame)
        self.all_pts.append(torch.from_numpy(keep))
        a = np.ascontiguousarray(keep, dtype=np.int32)
        rc = L.lib().nl_octree_insert(self._h, a.ctypes.data_as(ctypes.c_void_p), a.shape[0])
        if rc:
            raise RuntimeError("nl_octree_insert failed")

    def count_nodes(self):
        self._need()
        return int(L.lib().nl_octree_count_nodes(self._h))

    def count_leaf_nodes(self):
        self._need()
        return int(L.lib().nl_octree_count_leaf_nodes(self._h))

    def has_voxel(self, pt):
        self._need()
        p = [int(v) for v in torch.as_tensor(pt).reshape(-1).tolist()]
        return len(p) == 3 and bool(L.lib().nl_octree_has_voxel(self._h, *p))

    def get_centres_and_children(self):
        """-> (voxels f32[n,4], children f32[n,8], features i32[n,8]) like octree.cpp:293-342"""
        self._need()
        n = self.count_nodes()
        vox = np.empty((n, 4), np.float32)
        ch = np.empty((n, 8), np.float32)
        ft = np.empty((n, 8), np.int32)
        P = ctypes.c_void_p
        rc = L.lib().nl_octree_export(self._h, vox.ctypes.data_as(P), ch.ctypes.data_as(P), ft.ctypes.data_as(P))
        if rc:
            raise RuntimeError("nl_octree_export failed")
        return torch.from_numpy(vox), torch.from_numpy(ch), torch.from_numpy(ft)

    def get_voxels(self):
        v, _, _ = self.get_centres_and_children()
        return v

    def export_device_layout(self):
        """centres f32[n,3], structure i32[n,9], vertex_idx i32[n,8] (mapping.py:319-327 folded in)"""
        self._need()
        n = self.count_nodes()
        c = np.empty((n, 3), np.float32)
        s = np.empty((n, 9), np.int32)
        f = np.empty((n, 8), np.int32)
        P = ctypes.c_void_p
        rc = L.lib().nl_octree_export_device_layout(self._h, ctypes.c_float(self.voxel_size_), c.ctypes.data_as(P), s.ctypes.data_as(P),
                                                    f.ctypes.data_as(P))
        if rc:
            raise RuntimeError("nl_octree_export_device_layout failed")
        return c, s, f

    def export_delta(self):
        """Rows (export_device_layout format) of the nodes that changed since the previous call:
        ids i32[d], centres f32[d,3], structure i32[d,9], vertex_idx i32[d,8].  Applying the deltas in order to arrays of
        count_nodes() rows reproduces export_device_layout()."""
        self._need()
        d = int(L.lib().nl_octree_delta_count(self._h))
        ids = np.empty(d, np.int32); c = np.empty((d, 3), np.float32); s = np.empty((d, 9), np.int32); f = np.empty((d, 8), np.int32)
        P = ctypes.c_void_p
        if d:
            rc = L.lib().nl_octree_export_delta(self._h, ctypes.c_float(self.voxel_size_), ids.ctypes.data_as(P), c.ctypes.data_as(P),
                                                s.ctypes.data_as(P), f.ctypes.data_as(P))
            if rc:
                raise RuntimeError("nl_octree_export_delta failed")
        return ids, c, s, f

    # pickle protocol of bindings.cpp:23-31: (size, feat_dim, voxel_size, all_pts), rebuilt by replay
    def __getstate__(self):
        return (self.size_, self.feat_dim_, self.voxel_size_, self.all_pts)

    def __setstate__(self, state):
        self._h = None
        size, feat, vs, pts = state
        self.init(size, feat, vs)
        for p in pts:
            self.insert(p)

    def __del__(self):
        try:
            if self._h:
                L.lib().nl_octree_destroy(self._h)
                self._h = None
        except Exception:
            pass
