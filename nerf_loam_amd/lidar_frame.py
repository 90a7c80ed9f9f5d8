"""LidarFrame: one scan as a DEVICE-RESIDENT container (points + cos uploaded once, nothing else).

Keeps the constructor and accessor names of /root/reference/src/lidarFrame.py:9-57 (the Mapping / Tracking call sites use them),
but what the reference computes with host torch ops comes from HIP kernels here:

  * unit directions `rays_d = points / (||points|| + 1e-8)` (lidarFrame.py:47-52): never materialised on the hot path - the ray
    selection kernels compute the direction of a selected return in flight (nl_select.hip `sel_emit`); the `rays_d` / `rays_norm`
    attributes are lazy views of `nl_unit_dirs`' output for callers that want the whole array (bit-identical to the host ops);
  * the per-iteration ray subset (lidarFrame.py:55-57, sample_util.py:4-19): drawn on the device by nl_select_rays_batch
    (render_helpers.RAY_SELECTION = "device"); `sample_rays` below is only the seeded host fallback that consumes torch's global
    CPU generator exactly like the reference does.

Any object with `.index`, `.pose.data`, `.points`, `.pointsCos` works in bundle_adjust_frames / track_frame (render_helpers.scan_of),
the reference's own LidarFrame included; this class is what `Mapping.insert_keyframe` builds."""
import numpy as np
import torch
import torch.nn as nn

from .se3pose import OptimizablePose

WORLD_OFFSET = 2000.0           # lidarFrame.py:18 (B1): added to the translation of every frame that arrives with a pose matrix


def scan_of(frame, device):
    """The frame's returns resident on `device`, cached on the frame object: dict(points [M,3] f32, cos [M] f32, dirs, mask_u8 [M]).
    `dirs` is None when the points are fp32 - the selection kernels then derive directions from the points (a1 on the device);
    a frame that holds points in another dtype has its own `rays_d` uploaded (the reference divides in that dtype and rounds after)."""
    dev = torch.device(device)
    pts, pc = frame.points, frame.pointsCos
    key = (pts.data_ptr(), pts._version, int(pts.shape[0]), dev, pc.data_ptr(), pc._version)
    sc = frame.__dict__.get("_nl_scan")
    if sc is None or sc["key"] != key:
        p = pts.detach().reshape(-1, 3)
        dirs = None
        if p.dtype != torch.float32:
            own = frame.__dict__.get("rays_d", frame.__dict__.get("_nl_own_rays_d"))     # (the reference's LidarFrame keeps the array it built in that dtype)
            own = p / (torch.norm(p, 2, -1, keepdim=True) + 1e-8) if own is None else own
            dirs = own.detach().reshape(-1, 3).to(dev, torch.float32).contiguous()
        sc = dict(key=key, points=p.to(dev, torch.float32).contiguous(),
                  cos=frame.pointsCos.detach().reshape(-1).to(dev, torch.float32).contiguous(), dirs=dirs,
                  mask_u8=torch.zeros(int(pts.shape[0]), dtype=torch.uint8, device=dev))
        frame.__dict__["_nl_scan"] = sc
    return sc


class LidarFrame(nn.Module):
    def __init__(self, index, points, pointsCos, pose=None, new_keyframe=False):
        super().__init__()
        self.index = index
        self.num_point = len(points)
        self.points = points
        self.pointsCos = pointsCos
        if new_keyframe:
            self.pose = pose                                     # a key-scan shares the pose module of the frame it was cut from
        elif pose is not None:
            T = np.array(pose, dtype=np.float64, copy=True)
            T[:3, 3] += WORLD_OFFSET
            self.pose = OptimizablePose.from_matrix(torch.tensor(T, dtype=torch.float32))
        self.rel_pose = None
        self.sample_mask = None

    # ---- accessors of the reference surface
    def get_pose(self):
        return self.pose.matrix()

    def get_translation(self):
        return self.pose.translation()

    def get_rotation(self):
        return self.pose.rotation()

    def get_points(self):
        return self.points

    def get_pointsCos(self):
        return self.pointsCos

    def set_rel_pose(self, rel_pose):
        self.rel_pose = rel_pose

    def get_rel_pose(self):
        return self.rel_pose

    # ---- a1 on the device
    def device_scan(self, device="cuda"):
        return scan_of(self, device)

    @torch.no_grad()
    def get_rays(self, device="cuda"):
        """[M,1,3] unit directions, computed by nl_unit_dirs on the resident points (device tensor); sets `rays_norm` like the reference"""
        from . import ops
        sc = scan_of(self, device)
        if "rays_d" not in sc:
            M = sc["points"].shape[0]
            d = torch.empty(M, 3, dtype=torch.float32, device=sc["points"].device)
            n = torch.empty(M, dtype=torch.float32, device=sc["points"].device)
            ops.unit_dirs(sc["points"], d, n)
            sc["rays_d"], sc["rays_norm"] = d.view(M, 1, 3), n.view(M, 1)
        return sc["rays_d"]

    # `rays_d` [M,1,3] / `rays_norm` [M,1] as the reference's attributes (lidarFrame.py:47-52): on the device of `points`, so that
    # reference-style code mixing them with `points` (mapping.py:260-262 `frame.points[frame.rays_norm.reshape(-1) <= d]`) works on host
    # points too; computed once by nl_unit_dirs (the engine itself never reads them: device_scan / get_rays are its accessors).
    # Assignable like plain attributes (`frame.rays_d = ...` keeps the caller's tensor).
    def _host_side(self, name):
        own = self.__dict__.get("_nl_own_" + name)
        if own is not None:
            return own
        self.get_rays()
        sc = self.__dict__["_nl_scan"]
        t = sc[name]
        if t.device != self.points.device:
            k = name + "_on_points_device"
            if k not in sc:                                  # (dict.setdefault evaluates its default - a device-to-host copy and a sync - on EVERY access)
                sc[k] = t.to(self.points.device)
            t = sc[k]
        return t

    @property
    def rays_d(self):
        return self._host_side("rays_d")

    @rays_d.setter
    def rays_d(self, value):
        self.__dict__["_nl_own_rays_d"] = value
        self.__dict__.pop("_nl_scan", None)                  # (a non-fp32 frame's resident `dirs` are built from this tensor: rebuild the scan)

    @property
    def rays_norm(self):
        return self._host_side("rays_norm")

    @rays_norm.setter
    def rays_norm(self, value):
        self.__dict__["_nl_own_rays_norm"] = value

    # ---- a2, seeded host fallback (RAY_SELECTION = "host")
    @torch.no_grad()
    def sample_rays(self, N_rays, track=False):
        """Uniform N-subset by Gumbel top-k on torch's global CPU generator: the same draw, score arithmetic and top-k as
        sample_util.py:4-19 applied to an all-ones mask, so a seeded run selects the rays the reference selects (tests/test_reference_surface.py).
        Every return has the log-probability log(1/M + 1e-9); the Gumbel variate of a uniform u is -log(1e-7 - log(u + 1e-7))."""
        M = self.num_point
        ones = torch.ones(1, M)
        logp = torch.log(ones / (ones.sum() + 1e-9) + 1e-9)
        u = torch.rand_like(logp)
        keep = (logp + -torch.log(1e-7 - torch.log(u + 1e-7))).topk(N_rays, dim=-1).indices
        mask = torch.zeros(M, dtype=torch.bool)
        mask[keep.reshape(-1)] = True
        self.sample_mask = mask.view(M, 1)
