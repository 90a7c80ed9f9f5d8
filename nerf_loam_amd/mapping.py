"""Mapping: the hot-path call sites of the reference's mapper process.

Mirror of /root/reference/src/mapping.py for what SURVEY.md 8 puts in scope: `create_voxels`
(:283-291), `get_embeddings` (:293-317), `update_grid_features` (:319-339), `do_mapping` ->
`bundle_adjust_frames` (:172-202), `select_optimize_targets` (:205-225), `insert_keyframe` (:262-281).
`extract_mesh` (:354-378, round 6).  The process loop (spin), logging and the share-data plumbing stay with the reference.

MI355X-first differences (same results per vertex id):
  * the sparse octree is the native host octree (svo.Octree, flat arrays, per-instance counter);
  * the node-id -> embedding-row table is an [n_nodes] int32 tensor instead of a 2e9-row (8 GB)
    host tensor (mapping.py:76) and rows are allocated once per NEW VERTEX, not once per occurrence
    (the reference over-allocates ~3x and never reads the duplicates);
  * the bf16 embedding table grows ON THE DEVICE (the reference copies it D->H, cats, copies H->D
    every frame, mapping.py:312-314); map_states holds device tensors, so nothing is re-uploaded per
    iteration (render_helpers.py:76-77,205-206 re-upload the whole octree each iteration)."""
import random

import numpy as np
import torch

from . import hostenv
from .criterion import Criterion
from .decoder import Decoder
from .lidar_frame import LidarFrame
from .pipeline import MapDevice
from .render_helpers import bundle_adjust_frames
from .svo import Octree


def _get(d, name, default):
    return d.get(name, default) if isinstance(d, dict) else getattr(d, name, default)


class Mapping:
    def __init__(self, args, logger=None, device="cuda"):
        self.args, self.logger, self.device = args, logger, torch.device(device)
        hostenv.warn_once_if_pools_exceed_quota()
        self.decoder = Decoder(**args.decoder_specs).to(self.device)
        self.loss_criteria = Criterion(args)
        self.keyframe_graph = []
        self.initialized = False
        ms = args.mapper_specs
        self.voxel_size = ms["voxel_size"]
        self.window_size = ms["window_size"]
        self.num_iterations = ms["num_iterations"]
        self.n_rays = ms["N_rays_each"]
        self.sdf_truncation = args.criteria["sdf_truncation"]
        self.max_voxel_hit = ms["max_voxel_hit"]
        self.step_size = ms["step_size"] * self.voxel_size
        self.learning_rate_emb = ms["learning_rate_emb"]
        self.learning_rate_decorder = ms["learning_rate_decorder"]
        self.learning_rate_pose = ms["learning_rate_pose"]
        self.max_distance = args.data_specs["max_depth"]
        self.freeze_frame = ms["freeze_frame"]
        self.keyframe_gap = ms["keyframe_gap"]
        self.remove_back = ms["remove_back"]
        self.key_distance = ms["key_distance"]
        embed_dim = args.decoder_specs["in_dim"]
        self.embed_dim = embed_dim - 3 if ms["use_local_coord"] else embed_dim
        self.voxel_id2embedding_id = torch.full((0,), -1, dtype=torch.int32)        # [n_nodes] host table, grown with the tree
        self.current_num_embeds = 0
        self.dynamic_embeddings = None
        self._emb_buf = None                     # embedding table with spare capacity; dynamic_embeddings is a view of it
        self._node_buf = None                    # device copies of the octree tensors with spare capacity
        self.svo = Octree()
        self.svo.init(256 * 256 * 4, embed_dim, self.voxel_size)
        self.first_frame_id = 0
        self.current_keyframe = None
        self.map_states = None

    # ------------------------------------------------------------------ map growth (once per frame)
    def create_voxels(self, frame):
        pose = frame.get_pose().detach()
        pts = frame.get_points().float() @ pose[:3, :3].transpose(-1, -2) + pose[:3, 3]
        voxels = torch.div(pts, self.voxel_size, rounding_mode="floor")
        self.svo.insert(voxels.cpu().int())
        self.update_grid_features()

    @torch.no_grad()
    def get_embeddings(self, points_idx):
        """assign an embedding row to every vertex id seen for the first time; new rows are zero.  `points_idx` may be the
        vertex rows of the CHANGED nodes only (incremental update).  Returns the ids that received a row."""
        n = self.svo.count_nodes()
        if self.voxel_id2embedding_id.shape[0] < n:
            grown = torch.full((max(n, 2 * self.voxel_id2embedding_id.shape[0]),), -1, dtype=torch.int32)
            grown[:self.voxel_id2embedding_id.shape[0]] = self.voxel_id2embedding_id
            self.voxel_id2embedding_id = grown
        flat = points_idx.reshape(-1).long()
        flat = flat[flat.ne(-1)]
        new_ids = torch.unique(flat[self.voxel_id2embedding_id[flat].eq(-1)])
        if new_ids.numel() == 0:
            return new_ids
        start, end = self.current_num_embeds, self.current_num_embeds + new_ids.numel()
        self.voxel_id2embedding_id[new_ids] = torch.arange(start, end, dtype=torch.int32)
        # embedding table with spare capacity: growth appends zero rows in place (the reference round-trips the whole
        # table through the host every frame, mapping.py:312-314); a reallocation happens only when capacity doubles
        if self._emb_buf is None or end > self._emb_buf.shape[0]:
            cap = max(2 * end, 4096)
            buf = torch.zeros((cap, self.embed_dim), dtype=torch.bfloat16, device=self.device)
            if self._emb_buf is not None:
                buf[:start] = self._emb_buf[:start]
            self._emb_buf = buf
        self.dynamic_embeddings = self._emb_buf[:end]
        self.current_num_embeds = end
        return new_ids

    def _grow_nodes(self, n):
        cap = 0 if self._node_buf is None else self._node_buf["centres"].shape[0]
        if n <= cap:
            return
        new_cap = max(2 * n, 1 << 14)
        shapes = dict(centres=((new_cap, 3), torch.float32), structure=((new_cap, 9), torch.int32),
                      vertex_idx=((new_cap, 8), torch.int32), id2row=((new_cap,), torch.int32))
        buf = {k: torch.full(sh, -1 if k != "centres" else 0, dtype=dt, device=self.device) for k, (sh, dt) in shapes.items()}
        if self._node_buf is not None:
            for k in buf:
                buf[k][:cap] = self._node_buf[k]
        self._node_buf = buf

    @torch.no_grad()
    def update_grid_features(self):
        """Incremental map update: only the rows of nodes that changed since the last frame leave the host
        octree (svo.export_delta), are uploaded and scattered into capacity-managed device tensors; new vertex ids get
        zero embedding rows appended in place.  The reference re-exports the whole tree, rebuilds a 2e9-row id table entry
        by entry and re-uploads everything every frame (mapping.py:283-339)."""
        ids, centres, structure, vertex_idx = self.svo.export_delta()
        n = self.svo.count_nodes()
        self._grow_nodes(n)
        new_ids = self.get_embeddings(torch.from_numpy(vertex_idx))
        nb = self._node_buf
        if len(ids):
            di = torch.from_numpy(ids).to(self.device).long()
            nb["centres"].index_copy_(0, di, torch.from_numpy(centres).to(self.device))
            nb["structure"].index_copy_(0, di, torch.from_numpy(structure).to(self.device))
            nb["vertex_idx"].index_copy_(0, di, torch.from_numpy(vertex_idx).to(self.device))
        if new_ids.numel():
            nb["id2row"].index_copy_(0, new_ids.to(self.device), self.voxel_id2embedding_id[new_ids].to(self.device))
        self.map_states = {
            "voxel_vertex_idx": nb["vertex_idx"][:n],
            "voxel_center_xyz": nb["centres"][:n],
            "voxel_structure": nb["structure"][:n],
            "voxel_vertex_emb": self.dynamic_embeddings,
            "voxel_id2embedding_id": nb["id2row"][:n],
        }
        self.map_states["_device"] = MapDevice.from_tensors(nb["centres"][:n], nb["structure"][:n], nb["vertex_idx"][:n], nb["id2row"][:n],
                                                            self.dynamic_embeddings, self.voxel_size, self.device, blocks=self.svo.pack_blocks())

    # ------------------------------------------------------------------ mesh extraction (cold path)
    @torch.no_grad()
    def extract_mesh(self, res=8, clean_mesh=False):
        """reference mapping.py:354-378: the surface voxels (nodes whose eight vertex ids are all set), their dense SDF grids and the per-voxel marching
        cubes - grid and extraction on the device (mesh_util.MeshExtractor); mesh vertices carry the reference's offset of -2000"""
        from .mesh_util import MeshExtractor
        if getattr(self, "mesher", None) is None:
            self.mesher = MeshExtractor(self.args)
        self.decoder.eval()
        ms = self.map_states
        surface = ~ms["voxel_vertex_idx"].eq(-1).any(-1)
        # mapping.py:361: centres = (voxels[:, :3] + voxels[:, -1:] / 2) * voxel_size - what update_grid_features keeps in voxel_center_xyz
        states = {"voxel_vertex_idx": ms["voxel_vertex_idx"][surface].contiguous(), "voxel_center_xyz": ms["voxel_center_xyz"][surface].contiguous(),
                  "voxel_structure": ms["voxel_structure"][surface].contiguous(), "voxel_vertex_emb": self.dynamic_embeddings,
                  "voxel_id2embedding_id": ms["voxel_id2embedding_id"]}
        return self.mesher.create_mesh(self.decoder, states, self.voxel_size, states["voxel_center_xyz"], frame_poses=None, depth_maps=None,
                                       clean_mseh=clean_mesh, require_color=False, offset=-2000, res=res)

    # ------------------------------------------------------------------ optimisation call site
    def do_mapping(self, share_data=None, tracked_frame=None, update_pose=True, update_decoder=True, selection_method="current"):
        self.decoder.train()
        targets = self.select_optimize_targets(tracked_frame, selection_method=selection_method)
        bundle_adjust_frames(
            targets, self.dynamic_embeddings, self.map_states, self.decoder, self.loss_criteria, self.voxel_size, self.step_size,
            self.n_rays * 2 if selection_method == "random" else self.n_rays, self.num_iterations, self.sdf_truncation,
            self.max_voxel_hit, self.max_distance,
            learning_rate=[self.learning_rate_emb, self.learning_rate_decorder, self.learning_rate_pose],
            update_pose=update_pose,
            update_decoder=update_decoder if tracked_frame is None or (tracked_frame.index - self.first_frame_id) < self.freeze_frame else False)
        if share_data is not None:
            self.update_share_data(share_data)

    def select_optimize_targets(self, tracked_frame=None, selection_method="previous"):
        if selection_method == "current":
            if tracked_frame is None:
                raise ValueError("select one track frame")
            return [tracked_frame]
        if len(self.keyframe_graph) <= self.window_size:
            targets = self.keyframe_graph[:]
        elif selection_method == "random":
            targets = random.sample(self.keyframe_graph, self.window_size)
        elif selection_method == "previous":
            targets = self.keyframe_graph[-self.window_size:]
        else:
            raise NotImplementedError(f"seletion method {selection_method} unknown")
        if tracked_frame is not None and tracked_frame is not self.current_keyframe:
            targets = targets + [tracked_frame]
        return targets

    def insert_keyframe(self, frame, valid_distance=-1):
        """mapping.py:266-280 (`valid_distance` is accepted and overridden there too)"""
        lim = self.key_distance + 0.01
        mask = (frame.points.abs() < lim).all(-1)
        if int(mask.sum()) < 2 * self.n_rays:
            raise ValueError("valid_distance too small")
        kf = LidarFrame(frame.index, frame.points[mask], frame.get_pointsCos()[mask], frame.pose, new_keyframe=True)
        self.current_keyframe = kf
        self.keyframe_graph += [kf]

    def update_share_data(self, share_data, frameid=None):
        """decoder + map tensors for the tracker (mapping.py:227-232); tensors stay on the device"""
        share_data.decoder = self.decoder
        share_data.states = dict(self.map_states)
