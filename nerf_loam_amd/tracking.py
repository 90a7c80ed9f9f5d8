"""Tracking: the pose-refine call site of the reference's tracker process.

Mirror of /root/reference/src/tracking.py:98-148 (`do_tracking`): constant-velocity initialisation of
the translation, `track_frame`, relative-pose bookkeeping, the 5x iteration count on the first
tracked frame (B17), and the hand-over of every tracked frame to the mapper's queue (`check_keyframe`, :144-153: the reference's
`spin()` relies on `do_tracking` doing it).  The data loader and the process loop stay with the reference."""
from copy import deepcopy

import torch

from . import hostenv
from .criterion import Criterion
from .render_helpers import track_frame
from .se3pose import OptimizablePose


class Tracking:
    def __init__(self, args, data_stream=None, logger=None):
        self.args = args
        hostenv.warn_once_if_pools_exceed_quota()
        ts = args.tracker_specs
        self.loss_criteria = Criterion(args)
        self.voxel_size = args.mapper_specs["voxel_size"]
        self.N_rays = ts["N_rays"]
        self.num_iterations = ts["num_iterations"]
        self.sdf_truncation = args.criteria["sdf_truncation"]
        self.learning_rate = ts["learning_rate"]
        self.max_voxel_hit = ts["max_voxel_hit"]
        self.step_size = ts["step_size"] * self.voxel_size
        self.max_distance = args.data_specs["max_depth"]
        self.last_frame = None
        self.rel_pose = None

    def do_tracking(self, share_data, current_frame, kf_buffer=None):
        decoder = share_data.decoder
        map_states = share_data.states
        constant_move_pose = self.last_frame.get_pose().detach()
        input_pose = deepcopy(self.last_frame.pose)
        input_pose.requires_grad_(False)
        if self.rel_pose is not None:
            constant_move_pose[:3, 3] = (constant_move_pose @ self.rel_pose)[:3, 3]
            input_pose.data[:3] = constant_move_pose[:3, 3].T
        frame_pose, hit_mask = track_frame(
            input_pose, current_frame, map_states, decoder, self.loss_criteria, self.voxel_size, self.N_rays, self.step_size,
            self.num_iterations if self.rel_pose is not None else self.num_iterations * 5, self.sdf_truncation, self.learning_rate,
            self.max_voxel_hit, self.max_distance, profiler=None, depth_variance=True)
        if hit_mask is None:
            current_frame.pose = OptimizablePose.from_matrix(constant_move_pose)
        else:
            current_frame.pose = frame_pose
            current_frame.hit_ratio = hit_mask.sum() / self.N_rays
        self.rel_pose = torch.linalg.inv(self.last_frame.get_pose().detach()) @ current_frame.get_pose().detach()
        current_frame.set_rel_pose(self.rel_pose)
        self.last_frame = current_frame
        self.check_keyframe(current_frame, kf_buffer)             # tracking.py:144-147: the mapper process receives every tracked frame
        return current_frame

    def check_keyframe(self, check_frame, kf_buffer):
        """tracking.py:150-154: a blocking put on the mapper's queue; a missing / closed queue is ignored like the reference ignores it"""
        if kf_buffer is None:
            return
        try:
            kf_buffer.put(check_frame, block=True)
        except Exception:                                         # noqa: BLE001 - the reference's bare `except: pass`
            pass
