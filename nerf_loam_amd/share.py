"""ShareData: the mapper -> tracker hand-off, device-resident.

Mirror of /root/reference/src/share.py: the same attributes (`decoder`, `states`, `stop_mapping`, `stop_tracking`,
`tracking_trajectory`) with the same meaning - a setter publishes a SNAPSHOT that later mutation of the mapper's tensors does
not disturb, a getter returns the latest published snapshot.  The reference obtains the snapshot with deepcopy + moving every
tensor to the host + pickling through a manager process on every do_mapping, and the tracker moves everything back to the
GPU on every frame (mapping.py:227-232, tracking.py:101-107).  Here the snapshot is a device-to-device copy into one of two
capacity-managed device buffers (decoder parameters, octree tensors, embedding table: a few MB, one async copy kernel each)
and a version flip; the tracker reads views of the published buffer and reuses the packed traversal layout until the version
changes.  Copies and the kernels that read the snapshot are ordered by the stream they are enqueued on, so a reader never
sees a half-written buffer.

Scope: one process (mapper and tracker as threads or called in turn, as tests/ and the mirrored Mapping/Tracking do).  For two
processes the buffers can be handed over once with torch.multiprocessing (HIP IPC handles) after `reserve()`; the process
orchestration itself stays with the reference (out of scope)."""
import threading

import torch

from . import _lib as L
from .pipeline import MapDevice

_NODE_KEYS = (("voxel_center_xyz", 3, torch.float32), ("voxel_structure", 9, torch.int32), ("voxel_vertex_idx", 8, torch.int32))


class _Snapshot:
    def __init__(self):
        self.params = None          # [NL_DEC_PARAMS] f32
        self.nodes = {}             # key -> [cap, k]
        self.id2row = None          # [cap] i32
        self.emb = None             # [cap_rows, C] bf16
        self.n = self.rows = 0
        self.voxel_size = None
        self.has_decoder = self.has_states = False


class ShareData:
    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self._lock = threading.RLock()
        self._bufs = [_Snapshot(), _Snapshot()]
        self._cur = -1                       # index of the published snapshot
        self.version = 0                     # bumped by every `states` publication
        self._decoder_template = None
        self._cache = {}                     # version -> (states dict, decoder module)
        self.stop_mapping = False
        self.stop_tracking = False
        self.tracking_trajectory = []
        self._voxels = self._octree = None

    # ------------------------------------------------------------------ the remaining attributes of src/share.py (off the hot path)
    @property
    def voxels(self):
        from copy import deepcopy
        with self._lock:
            return deepcopy(self._voxels)

    @voxels.setter
    def voxels(self, voxels):
        from copy import deepcopy
        with self._lock:
            self._voxels = deepcopy(voxels)

    @property
    def octree(self):
        from copy import deepcopy
        with self._lock:
            return deepcopy(self._octree)

    @octree.setter
    def octree(self, octree):
        from copy import deepcopy
        with self._lock:
            self._octree = deepcopy(octree)

    def push_pose(self, pose):
        from copy import deepcopy
        with self._lock:
            self.tracking_trajectory.append(deepcopy(pose))

    # ------------------------------------------------------------------ capacity
    def reserve(self, n_nodes, n_rows, channels=16):
        """pre-allocate both buffers (no reallocation - hence stable IPC handles - while the map stays below these sizes)"""
        with self._lock:
            for b in self._bufs:
                self._fit(b, n_nodes, n_rows, channels)

    def _fit(self, b, n, rows, channels):
        if b.id2row is None or b.id2row.shape[0] < n:
            cap = max(2 * n, 1 << 14)
            b.nodes = {k: torch.empty((cap, w), dtype=dt, device=self.device) for k, w, dt in _NODE_KEYS}
            b.id2row = torch.empty(cap, dtype=torch.int32, device=self.device)
        if b.emb is None or b.emb.shape[0] < rows or b.emb.shape[1] != channels:
            b.emb = torch.empty((max(2 * rows, 4096), channels), dtype=torch.bfloat16, device=self.device)
        if b.params is None:
            b.params = torch.empty(L.NL_DEC_PARAMS, dtype=torch.float32, device=self.device)

    def _pending(self):
        return self._bufs[1 - max(self._cur, 0)] if self._cur >= 0 else self._bufs[0]

    # ------------------------------------------------------------------ decoder
    @property
    def decoder(self):
        """a Decoder module holding the published parameters (built once per version), or None before the first publication"""
        with self._lock:
            if self._cur < 0 or not self._bufs[self._cur].has_decoder:
                return None
            ent = self._cache.setdefault(self.version, {})
            if "decoder" not in ent:
                from copy import deepcopy
                m = deepcopy(self._decoder_template)
                m.load_flat(self._bufs[self._cur].params)
                ent["decoder"] = m
            return ent["decoder"]

    @decoder.setter
    def decoder(self, module):
        with self._lock:
            b = self._pending()
            if b.params is None:
                b.params = torch.empty(L.NL_DEC_PARAMS, dtype=torch.float32, device=self.device)
            b.params.copy_(module.flat_params(self.device), non_blocking=True)
            b.has_decoder = True
            if self._decoder_template is None:
                from copy import deepcopy
                self._decoder_template = deepcopy(module)

    # ------------------------------------------------------------------ map states
    @property
    def states(self):
        """the reference's map_states dict over the published snapshot (views, no copies) + the packed traversal layout"""
        with self._lock:
            if self._cur < 0 or not self._bufs[self._cur].has_states:
                return None
            ent = self._cache.setdefault(self.version, {})
            if "states" not in ent:
                b = self._bufs[self._cur]
                st = {k: b.nodes[k][:b.n] for k, _, _ in _NODE_KEYS}
                st["voxel_id2embedding_id"] = b.id2row[:b.n]
                st["voxel_vertex_emb"] = b.emb[:b.rows]
                if b.voxel_size is not None:
                    st["_device"] = MapDevice.from_tensors(st["voxel_center_xyz"], st["voxel_structure"], st["voxel_vertex_idx"],
                                                           st["voxel_id2embedding_id"], st["voxel_vertex_emb"], b.voxel_size, self.device)
                ent["states"] = st
            return ent["states"]

    @states.setter
    def states(self, states):
        with self._lock:
            b = self._pending()
            n = int(states["voxel_center_xyz"].shape[0])
            emb = states["voxel_vertex_emb"]
            rows = int(emb.shape[0])
            self._fit(b, n, rows, int(emb.shape[1]))
            for k, _, dt in _NODE_KEYS:
                b.nodes[k][:n].copy_(states[k].to(self.device, dt), non_blocking=True)
            table = states["voxel_id2embedding_id"].reshape(-1)
            b.id2row[:n].copy_(table[:n].to(self.device, torch.int32), non_blocking=True)
            b.emb[:rows].copy_(emb.detach(), non_blocking=True)
            b.n, b.rows = n, rows
            md = states.get("_device")
            b.voxel_size = md.voxel_size if md is not None else b.voxel_size
            b.has_states = True
            if not b.has_decoder and self._cur >= 0 and self._bufs[self._cur].has_decoder:      # carry the decoder over
                b.params.copy_(self._bufs[self._cur].params, non_blocking=True)
                b.has_decoder = True
            self._cur = self._bufs.index(b)
            self.version += 1
            old = self._bufs[1 - self._cur]
            old.has_decoder = old.has_states = False                                            # becomes the pending buffer
            for v in [v for v in self._cache if v < self.version]:
                del self._cache[v]
