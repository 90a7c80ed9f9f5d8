"""Seeded synthetic LiDAR scans for benchmarks and parity tests (no dataset is reachable offline).

Geometry follows SURVEY.md 8(d): sensor at the origin, ground plane z = -1.8 m, street-canyon
walls at x = +-30 m and y = +-10 m, `n_beams` elevations linspace(+2 deg, -24.8 deg) x `n_azimuth`
azimuths; analytic ranges.  `pointsCos` mirrors the reference datasets' convention
(/root/reference/src/dataset/maicity.py:55-70): |n.p|/||p|| on ground returns, 1 elsewhere.
The world pose carries the reference's +2000 m offset (/root/reference/src/lidarFrame.py:18).
"""
import numpy as np

WORLD_OFFSET = 2000.0


def synthetic_scan(n_beams=64, n_azimuth=2048, seed=777, range_noise=0.0, sector=(0.0, 1.0)):
    """Returns points [M,3] f32 (sensor frame) and pointsCos [M] f32, beam-major order.
    `sector` = (start, end) fraction of the full revolution covered by the n_azimuth columns (small
    test scenes keep the real scan's angular density by shrinking the sector)."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(2.0, -24.8, n_beams))
    span = 2.0 * np.pi * (sector[1] - sector[0])
    azim = 2.0 * np.pi * sector[0] + np.linspace(0.0, span, n_azimuth, endpoint=False) + 0.5 * span / (2 * n_azimuth)
    e, a = np.meshgrid(elev, azim, indexing="ij")
    u = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], -1).reshape(-1, 3)
    with np.errstate(divide="ignore"):
        t_ground = np.where(u[:, 2] < 0, -1.8 / u[:, 2], np.inf)
        t_wx = 30.0 / np.abs(u[:, 0])
        t_wy = 10.0 / np.abs(u[:, 1])
    t = np.minimum(np.minimum(t_ground, t_wx), t_wy)
    if range_noise > 0:
        t = t + rng.normal(0.0, range_noise, t.shape)
    pts = (u * t[:, None]).astype(np.float32)
    is_ground = t_ground <= np.minimum(t_wx, t_wy)
    cos = np.where(is_ground, np.abs(pts[:, 2]) / np.linalg.norm(pts, axis=1), 1.0).astype(np.float32)
    keep = (t > 1.5) & (t < 50.0)
    return pts[keep], cos[keep]


def scan_pose(tx=0.0, ty=0.0, tz=0.0, w=(0.0, 0.0, 0.0)):
    """6-vector [t, w] like se3pose.OptimizablePose.data, with the +2000 m world offset."""
    return np.array([tx + WORLD_OFFSET, ty + WORLD_OFFSET, tz + WORLD_OFFSET, *w], np.float32)


def unit_dirs(points):
    """lidarFrame.py:47-52: rays_d = points / (||points|| + 1e-8), in the arithmetic of the reference's host torch ops: the norm kernel
    accumulates the squares with fused multiply-adds (s = x x; s = fma(y, y, s); s = fma(z, z, s)), emulated here through fp64 (a
    24 x 24-bit product is exact in fp64).  Host-side restatement of nl_unit_dir (csrc/nl_device_math.h) for the synthetic-data
    generator and the oracle; pinned against torch and the device function in tests/test_device_math_host.py."""
    p = np.asarray(points, np.float32)
    q = p.astype(np.float64)
    s = (p[..., 0] * p[..., 0]).astype(np.float32)
    s = (q[..., 1] * q[..., 1] + s.astype(np.float64)).astype(np.float32)
    s = (q[..., 2] * q[..., 2] + s.astype(np.float64)).astype(np.float32)
    nrm = np.sqrt(s) + np.float32(1e-8)
    return (p / nrm[..., None]).astype(np.float32)


def voxel_coords(points, pose_R, pose_t, voxel_size):
    """mapping.py:283-289: floor((points @ R^T + t) / voxel_size) as int32."""
    pw = (points.astype(np.float32) @ pose_R.T.astype(np.float32) + pose_t.astype(np.float32)).astype(np.float32)
    return np.floor(pw / np.float32(voxel_size)).astype(np.int32)
