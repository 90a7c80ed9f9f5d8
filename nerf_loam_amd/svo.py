"""`svo.Octree` operator surface on the native host octree (csrc/nl_octree.cpp).

Mirrors torch.classes.svo.Octree (/root/reference/third_party/sparse_octree/src/bindings.cpp:4-31):
same method names (init, insert, try_insert, get_voxels, get_leaf_voxels, get_features, count_nodes, count_leaf_nodes, has_voxel,
get_centres_and_children; the free op `encode`), argument meaning and returned tensors (CPU torch tensors, same dtypes/shapes),
same pickle state (size, feat_dim, voxel_size, list of inserted tensors).  Differences, on purpose:
the node counter is per instance (the reference's is process-global) and malformed
input raises instead of printing to stdout."""
import ctypes

import numpy as np
import torch

from . import _lib as L


def encode(coords):
    """torch.ops.svo.encode (bindings.cpp:6, include/test.h:75-86): i64[K,3] -> i64[K,1] Morton keys of 21-bit coordinates.  As in
    the reference, the z slot of the key is filled from the x column (test.h:82 reads coords[3 i] twice)."""
    c = torch.as_tensor(coords)
    if c.dim() != 2 or c.shape[1] != 3 or c.dtype != torch.int64:
        raise ValueError("encode expects an int64 tensor [K,3]")
    a = c.cpu().numpy().astype(np.uint64)

    def spread(v):
        x = v & np.uint64(0x1fffff)
        for sh, mask in ((32, 0x1f00000000ffff), (16, 0x1f0000ff0000ff), (8, 0x100f00f00f00f00f), (4, 0x10c30c30c30c30c3), (2, 0x1249249249249249)):
            x = (x | (x << np.uint64(sh))) & np.uint64(mask)
        return x
    key = (spread(a[:, 0]) | (spread(a[:, 1]) << np.uint64(1)) | (spread(a[:, 0]) << np.uint64(2))) & np.uint64(0x7fffffffffffffff)
    return torch.from_numpy(key.astype(np.int64)).reshape(-1, 1)


def load_torch_library():
    """Register the reference's TorchScript names - torch.classes.svo.Octree, torch.classes.svo.Octant, torch.ops.svo.encode
    (bindings.cpp:4-31) - from nerf_loam_amd/libnl_svo_torch.so and return its path.  The reference loads its own build with
    `torch.classes.load_library(<absolute path>)` (src/mapping.py:19-20): pointing that line at the returned path is the whole
    change its Mapping needs.  (This module's `Octree` below is the same octree behind a plain Python class, with the extra
    incremental-export methods the device-resident Mapping uses.)"""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnl_svo_torch.so")
    if not os.path.exists(path):
        from . import build
        build.build()
        build.build_torch_ext()
    L.lib()                                                  # libnerfloam_hip.so first (torch's HIP runtime before ours, _lib.lib())
    torch.classes.load_library(path)
    return path


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
        """-> f32[n,4]: (x, y, z, side) of every node in depth-first octant order (octree.cpp:242-265) - NOT the node-id order of
        get_centres_and_children()"""
        self._need()
        out = np.empty((self.count_nodes(), 4), np.float32)
        if L.lib().nl_octree_voxels_dfs(self._h, out.ctypes.data_as(ctypes.c_void_p)):
            raise RuntimeError("nl_octree_voxels_dfs failed")
        return torch.from_numpy(out)

    def get_leaf_voxels(self):
        """-> f32[L,3]: integer coordinates of the SURFACE leaves, depth-first octant order (octree.cpp:212-240)"""
        self._need()
        n = int(L.lib().nl_octree_leaf_voxels(self._h, None))
        out = np.empty((n, 3), np.float32)
        if n:
            L.lib().nl_octree_leaf_voxels(self._h, out.ctypes.data_as(ctypes.c_void_p))
        return torch.from_numpy(out)

    def try_insert(self, pts):
        """fraction of the vertex keys of `pts` already in the tree, nothing inserted (octree.cpp:113-149)"""
        self._need()
        t = torch.as_tensor(pts)
        if t.dim() != 2 or t.shape[1] != 3:
            raise ValueError(f"Point dimensions mismatch: inputs are {tuple(t.shape)} expect [M,3]")
        a = np.ascontiguousarray(t.cpu().numpy(), dtype=np.int32)
        return float(L.lib().nl_octree_try_insert(self._h, a.ctypes.data_as(ctypes.c_void_p), a.shape[0]))

    def get_features(self, pts):
        raise NotImplementedError("Octree::get_features has an empty body in the reference (octree.cpp:208-210)")

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

    def pack_blocks(self):
        """the children-block traversal layout of the ray-intersect kernels, packed by the octree itself (nl_octree_pack_blocks): blk_ids i32[B,8],
        blk_hdr i32[B,2] - what pipeline.pack_children_blocks derives from the exported structure rows"""
        self._need()
        P = ctypes.c_void_p
        B = int(L.lib().nl_octree_pack_blocks(self._h, None, None, 0))
        ids = np.empty((B, 8), np.int32); hdr = np.empty((B, 2), np.int32)
        if int(L.lib().nl_octree_pack_blocks(self._h, ids.ctypes.data_as(P), hdr.ctypes.data_as(P), B)) != B:
            raise RuntimeError("nl_octree_pack_blocks failed")
        return ids, hdr

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
