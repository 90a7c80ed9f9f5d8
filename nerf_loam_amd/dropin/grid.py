"""`import grid` drop-in: put THIS directory on sys.path / PYTHONPATH and the reference's `import grid as _ext`
(/root/reference/src/variations/voxel_helpers.py:22) resolves to the MI355X implementation - same two live entry points, same
argument order, tensor layouts and dtypes as third_party/sparse_voxels/src/binding.cpp:10-21.  The five other names that module
exports are dead code in NeRF-LOAM (NSVF leftovers): they exist here so attribute access does not fail at import
time, and raise if anybody calls them."""
from nerf_loam_amd.grid import inverse_cdf_sampling, svo_intersect  # noqa: F401


def _dead(name):
    def f(*a, **k):
        raise NotImplementedError(f"grid.{name} is not on NeRF-LOAM's path (no reference call site) and is not implemented")
    f.__name__ = name
    return f


ball_intersect = _dead("ball_intersect")
aabb_intersect = _dead("aabb_intersect")
triangle_intersect = _dead("triangle_intersect")
uniform_ray_sampling = _dead("uniform_ray_sampling")
build_octree = _dead("build_octree")
