"""MeshExtractor: the mesh-extraction call site of the reference's mapper.

Mirror of /root/reference/src/utils/mesh_util.py for the path the LiDAR pipeline runs (`Mapping.extract_mesh`, mapping.py:354-378 ->
`create_mesh(..., clean_mseh=False, require_color=False)`, :80-142 -> `get_scores` -> `marching_cubes`, :145-169).  The reference evaluates the
SDF grid in chunks of 10 000 voxels with a `.cpu()` each, then runs scikit-image's marching cubes on the CPU one voxel at a time; here the grid stays on the
device (`get_scores(device_out=True)`: nl_gather_grid + the matrix-core decoder forward) and the extraction is two launches around two prefix scans
(csrc/nl_mesh.hip), with one read-back of the totals and one of the mesh.

Same surface, not the same arrays: the vertex set is the one any linear-interpolation marching cubes produces and the world map is the reference's
(`(v / (res - 1) - 0.5) * voxel_size + centre`, then `+ offset`), but vertex / face order and the triangulation of a cell are this library's (the case table
is derived in scripts/gen_mc_table.py; scikit-image is not installed here and its tables are not part of the reference's sources).

`clean_mseh` / `require_color` are dead for LiDAR in the reference (`self.rays_d` is never set: get_valid_points would fail; mapping.py passes False for
both) and raise here."""
import numpy as np
import torch

from . import ops
from .render_helpers import get_scores


class TriangleMesh:
    """what create_mesh returns when open3d is not importable: the two arrays open3d's TriangleMesh would hold"""

    def __init__(self, vertices, triangles):
        self.vertices, self.triangles = vertices, triangles

    def __repr__(self):
        return f"TriangleMesh with {len(self.vertices)} points and {len(self.triangles)} triangles."


class MeshExtractor:
    def __init__(self, args):
        self.voxel_size = args.mapper_specs["voxel_size"]
        self.rays_d = None
        self.depth_points = None

    @torch.no_grad()
    def marching_cubes(self, voxels, sdf, device_out=False):
        """voxels [n, >= 3] centres, sdf [n, res, res, res, 1] (host or device) -> (verts [N, 3] float32, faces [M, 3] int32) as numpy arrays like the
        reference's (device_out=True: device tensors)"""
        dev = sdf.device if sdf.is_cuda else torch.device("cuda")
        s = sdf.to(dev, torch.float32)
        if s.dim() == 5:
            s = s[..., 0]
        c = voxels[:, :3].detach().to(dev, torch.float32).contiguous()
        verts, faces = ops.marching_cubes(s.contiguous(), c, self.voxel_size)
        if device_out:
            return verts, faces
        return verts.cpu().numpy(), faces.cpu().numpy()

    @torch.no_grad()
    def create_mesh(self, decoder, map_states, voxel_size, voxels, frame_poses=None, depth_maps=None, clean_mseh=False, require_color=False,
                    offset=-80, res=8):
        if clean_mseh or require_color:
            raise NotImplementedError("clean_mseh / require_color: dead code for LiDAR in the reference (mesh_util.py:91-135 needs the RGB-D fields rays_d / depth maps); "
                                      "Mapping.extract_mesh passes False for both")
        sdf_grid = get_scores(decoder, map_states, voxel_size, bits=res, device_out=True)
        verts, faces = self.marching_cubes(map_states["voxel_center_xyz"], sdf_grid.reshape(-1, res, res, res, 1))
        verts = verts + np.float32(offset)
        try:
            import open3d as o3d
        except ImportError:
            return TriangleMesh(verts, faces)
        mesh = o3d.geometry.TriangleMesh()
        mesh.vertices = o3d.utility.Vector3dVector(verts)
        mesh.triangles = o3d.utility.Vector3iVector(faces)
        mesh.compute_vertex_normals()
        return mesh
