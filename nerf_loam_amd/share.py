"""ShareData: the mapper -> tracker hand-off, device-resident, in one process or across two.

Mirror of /root/reference/src/share.py: the same attributes (`decoder`, `states`, `stop_mapping`, `stop_tracking`,
`tracking_trajectory`, `voxels`, `octree`, `push_pose`) with the same meaning - a setter publishes a SNAPSHOT that later mutation
of the mapper's tensors does not disturb, a getter returns the latest published snapshot.  The reference obtains the snapshot
with deepcopy + moving every tensor to the host + pickling through a manager process on every do_mapping, and the tracker moves
everything back to the GPU on every frame (mapping.py:227-232, tracking.py:101-107, nerfloam.py:23-49).

Here a snapshot is a device-to-device copy into one of THREE capacity-managed device buffers plus a flip of a small control block:

  * the control block (version, index of the published buffer, the reader's lease, per-buffer sizes, stop flags) lives in shared
    host memory (`torch.Tensor.share_memory_`); the device buffers are plain torch CUDA tensors;
  * two processes: `handles()` returns a picklable description (control block + the device buffers); sent to a process started
    with torch.multiprocessing (spawn) the buffers travel as HIP IPC handles (dmabuf; HSA_ENABLE_IPC_MODE_LEGACY=0), and
    `ShareData.attach(handles)` in the child gives the tracker a ShareData over THE SAME device memory: a publication is a few
    device-to-device copies in the mapper process and a version flip, the tracker reads views - no host copy, no pickling of tensors
    per frame.  `reserve()` first: the buffers must not be reallocated after they were handed over;
  * a reader holds a LEASE on the buffer it reads (`states` / `decoder` getters take it, re-reading moves it): the publisher never
    writes the published buffer nor the leased one - with three buffers there is always a free one, so a tracker that keeps
    working on a snapshot for a whole track_frame call while the mapper publishes twice is never overwritten in place;
  * the publisher synchronises its stream before it flips the version (another process cannot order itself against this
    process's stream), once per published frame.

The process loop itself (spawning mapper / tracker, queues of frames) stays with the reference (out of scope)."""
import threading

import torch

from . import _lib as L
from .pipeline import MapDevice

_NODE_KEYS = (("voxel_center_xyz", 3, torch.float32), ("voxel_structure", 9, torch.int32), ("voxel_vertex_idx", 8, torch.int32))
N_BUF = 3
# control block (int64): [0] version, [1] published buffer (-1 none), [2] leased buffer (-1 none), [3] stop_mapping, [4] stop_tracking,
# then per buffer b at 8 + 8 b: n_nodes, n_rows, has_states, has_decoder, voxel_size (float64 bits), channels
_CTL = 8 + 8 * N_BUF
_VER, _CUR, _LEASE, _STOPM, _STOPT = range(5)


class _Buffers:
    def __init__(self):
        self.params = None          # [NL_DEC_PARAMS] f32
        self.nodes = {}             # key -> [cap, k]
        self.id2row = None          # [cap] i32
        self.emb = None             # [cap_rows, C] bf16


class ShareData:
    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self._lock = threading.RLock()
        self._bufs = [_Buffers() for _ in range(N_BUF)]
        self._ctl = torch.zeros(_CTL, dtype=torch.int64).share_memory_()
        self._ctl[_CUR] = -1
        self._ctl[_LEASE] = -1
        self._pending = None                 # buffer a publication in progress (decoder set, states not yet) writes into
        self._decoder_template = None
        self._cache = {}                     # version -> {"states": dict, "decoder": module}
        self._attached = False
        self.tracking_trajectory = []
        self._voxels = self._octree = None

    # ------------------------------------------------------------------ control block
    def _meta(self, b):
        o = 8 + 8 * b
        c = self._ctl
        return dict(n=int(c[o]), rows=int(c[o + 1]), has_states=bool(c[o + 2]), has_decoder=bool(c[o + 3]),
                    voxel_size=float(c[o + 4:o + 5].view(torch.float64)[0]), channels=int(c[o + 5]))

    def _set_meta(self, b, **kw):
        o = 8 + 8 * b
        c = self._ctl
        for k, i in (("n", 0), ("rows", 1), ("has_states", 2), ("has_decoder", 3), ("channels", 5)):
            if k in kw:
                c[o + i] = int(kw[k])
        if "voxel_size" in kw:
            c[o + 4:o + 5].view(torch.float64)[0] = float(kw["voxel_size"])

    @property
    def version(self):
        return int(self._ctl[_VER])

    @property
    def stop_mapping(self):
        return bool(self._ctl[_STOPM])

    @stop_mapping.setter
    def stop_mapping(self, v):
        self._ctl[_STOPM] = int(bool(v))

    @property
    def stop_tracking(self):
        return bool(self._ctl[_STOPT])

    @stop_tracking.setter
    def stop_tracking(self, v):
        self._ctl[_STOPT] = int(bool(v))

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

    # ------------------------------------------------------------------ capacity / hand-over to another process
    def reserve(self, n_nodes, n_rows, channels=16):
        """pre-allocate all buffers: no reallocation - hence stable IPC handles - while the map stays below these sizes"""
        with self._lock:
            for b in self._bufs:
                self._fit(b, n_nodes, n_rows, channels, exact=True)

    def _fit(self, b, n, rows, channels, exact=False):
        grow_nodes = b.id2row is None or b.id2row.shape[0] < n
        grow_rows = b.emb is None or b.emb.shape[0] < rows or b.emb.shape[1] != channels
        if (grow_nodes or grow_rows) and self._attached:
            raise L.NerfLoamHipError("ShareData: the map outgrew the buffers that were handed to the other process - reserve() more before "
                                     "handles()")
        if grow_nodes:
            cap = n if exact else max(2 * n, 1 << 14)
            b.nodes = {k: torch.empty((cap, w), dtype=dt, device=self.device) for k, w, dt in _NODE_KEYS}
            b.id2row = torch.empty(cap, dtype=torch.int32, device=self.device)
        if grow_rows:
            b.emb = torch.empty((rows if exact else max(2 * rows, 4096), channels), dtype=torch.bfloat16, device=self.device)
        if b.params is None:
            b.params = torch.empty(L.NL_DEC_PARAMS, dtype=torch.float32, device=self.device)

    def handles(self, decoder_template=None):
        """picklable description of this ShareData for `ShareData.attach` in a process started with torch.multiprocessing (spawn):
        the control block (shared host memory) and the device buffers (they travel as HIP IPC handles).  Call after reserve();
        keep this object alive while the other process uses the buffers."""
        with self._lock:
            if any(b.emb is None for b in self._bufs):
                raise L.NerfLoamHipError("ShareData.handles(): reserve(n_nodes, n_rows) first")
            self._attached = True                # from now on the buffers must not be reallocated
            tmpl = decoder_template if decoder_template is not None else self._decoder_template
            return dict(ctl=self._ctl, device=str(self.device),
                        bufs=[dict(params=b.params, nodes=dict(b.nodes), id2row=b.id2row, emb=b.emb) for b in self._bufs],
                        decoder_state=None if tmpl is None else {k: v.detach().cpu() for k, v in tmpl.state_dict().items()})

    @classmethod
    def attach(cls, handles):
        """the other process's end: a ShareData over the SAME control block and device buffers (no copies)"""
        self = cls.__new__(cls)
        self.device = torch.device(handles["device"])
        self._lock = threading.RLock()
        self._ctl = handles["ctl"]
        self._bufs = []
        for h in handles["bufs"]:
            b = _Buffers()
            b.params, b.nodes, b.id2row, b.emb = h["params"], h["nodes"], h["id2row"], h["emb"]
            self._bufs.append(b)
        self._pending = None
        self._cache = {}
        self._attached = True
        self.tracking_trajectory = []
        self._voxels = self._octree = None
        self._decoder_template = None
        if handles.get("decoder_state") is not None:
            from .decoder import Decoder
            m = Decoder().to(self.device)
            m.load_state_dict(handles["decoder_state"])
            self._decoder_template = m
        return self

    # ------------------------------------------------------------------ publication (mapper side)
    def _pick_pending(self):
        """a buffer that is neither published nor leased by the reader (three buffers: one always exists)"""
        if self._pending is None:
            cur, lease = int(self._ctl[_CUR]), int(self._ctl[_LEASE])
            self._pending = next(b for b in range(N_BUF) if b != cur and b != lease)
            self._set_meta(self._pending, has_states=0, has_decoder=0)
        return self._pending

    def _publish(self, b):
        """make buffer b the published snapshot: the copies above must have completed before another process (or thread on another
        stream) may read them"""
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        self._ctl[_CUR] = b
        self._ctl[_VER] += 1
        self._pending = None
        for v in [v for v in self._cache if v < self.version]:
            del self._cache[v]

    # ------------------------------------------------------------------ reader side: lease + views
    def _lease_current(self):
        """take the lease on the published buffer; re-checked so that a publication racing with it cannot hand out a buffer that
        is being rewritten"""
        for _ in range(8):
            cur = int(self._ctl[_CUR])
            if cur < 0:
                return -1, 0
            self._ctl[_LEASE] = cur
            ver = int(self._ctl[_VER])
            if int(self._ctl[_CUR]) == cur:
                return cur, ver
        return cur, ver

    def _cache_entry(self, ver):
        """per-version views / modules; older versions are dropped HERE as well as in _publish: a ShareData obtained through
        attach() (the tracker process) never publishes, and every cached version holds a MapDevice (packed traversal blocks,
        vertex rows: MBs of device memory) and a Decoder copy.  Single reader per process: there is one lease slot, and
        whoever reads `states` / `decoder` last holds it."""
        for v in [v for v in self._cache if v != ver]:
            del self._cache[v]
        return self._cache.setdefault(ver, {})

    def release(self):
        """give the lease back (optional: re-reading `states` moves it anyway)"""
        self._ctl[_LEASE] = -1

    # ------------------------------------------------------------------ decoder
    @property
    def decoder(self):
        """a Decoder module holding the published parameters (built once per version), or None before the first publication"""
        with self._lock:
            cur, ver = self._lease_current()
            if cur < 0 or not self._meta(cur)["has_decoder"] or self._decoder_template is None:
                return None
            ent = self._cache_entry(ver)
            if "decoder" not in ent:
                from copy import deepcopy
                m = deepcopy(self._decoder_template)
                m.load_flat(self._bufs[cur].params)
                ent["decoder"] = m
            return ent["decoder"]

    @decoder.setter
    def decoder(self, module):
        with self._lock:
            i = self._pick_pending()
            b = self._bufs[i]
            if b.params is None:
                b.params = torch.empty(L.NL_DEC_PARAMS, dtype=torch.float32, device=self.device)
            b.params.copy_(module.flat_params(self.device), non_blocking=True)
            self._set_meta(i, has_decoder=1)
            if self._decoder_template is None:
                from copy import deepcopy
                self._decoder_template = deepcopy(module)

    # ------------------------------------------------------------------ map states
    @property
    def states(self):
        """the reference's map_states dict over the published snapshot (views, no copies) + the packed traversal layout"""
        with self._lock:
            cur, ver = self._lease_current()
            if cur < 0:
                return None
            meta = self._meta(cur)
            if not meta["has_states"]:
                return None
            ent = self._cache_entry(ver)
            if "states" not in ent:
                b = self._bufs[cur]
                n, rows = meta["n"], meta["rows"]
                st = {k: b.nodes[k][:n] for k, _, _ in _NODE_KEYS}
                st["voxel_id2embedding_id"] = b.id2row[:n]
                st["voxel_vertex_emb"] = b.emb[:rows]
                if meta["voxel_size"] > 0:
                    st["_device"] = MapDevice.from_tensors(st["voxel_center_xyz"], st["voxel_structure"], st["voxel_vertex_idx"],
                                                           st["voxel_id2embedding_id"], st["voxel_vertex_emb"], meta["voxel_size"], self.device)
                ent["states"] = st
            return ent["states"]

    @states.setter
    def states(self, states):
        with self._lock:
            i = self._pick_pending()
            b = self._bufs[i]
            n = int(states["voxel_center_xyz"].shape[0])
            emb = states["voxel_vertex_emb"]
            rows = int(emb.shape[0])
            self._fit(b, n, rows, int(emb.shape[1]))
            for k, _, dt in _NODE_KEYS:
                b.nodes[k][:n].copy_(states[k].to(self.device, dt), non_blocking=True)
            table = states["voxel_id2embedding_id"].reshape(-1)
            b.id2row[:n].copy_(table[:n].to(self.device, torch.int32), non_blocking=True)
            b.emb[:rows].copy_(emb.detach(), non_blocking=True)
            md = states.get("_device")
            cur = int(self._ctl[_CUR])
            vs = md.voxel_size if md is not None else (self._meta(cur)["voxel_size"] if cur >= 0 else 0.0)
            self._set_meta(i, n=n, rows=rows, has_states=1, voxel_size=vs, channels=int(emb.shape[1]))
            if not self._meta(i)["has_decoder"] and cur >= 0 and self._meta(cur)["has_decoder"]:        # carry the decoder over
                b.params.copy_(self._bufs[cur].params, non_blocking=True)
                self._set_meta(i, has_decoder=1)
            self._publish(i)
