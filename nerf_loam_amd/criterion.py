"""Criterion: free-space + truncated-surface SDF losses.

Mirror of /root/reference/src/criterion.py: same constructor (reads args.criteria / args.data_specs)
and call signature, returns (loss, loss_dict).  On MI355X the masks, the data-dependent
re-weighting (criterion.py:84-88), the squared-residual sums and dL/dsdf are produced inside the
fused kernels (nl_geometry.hip k_sample / k_loss_finalize, nl_decoder.hip phase D); this object
carries the hyper-parameters into them and turns the device sums into the scalar the reference's
callers log.  `compute_eikonal_loss` is never enabled by any reference call site and is rejected."""
import torch
import torch.nn as nn


class Criterion(nn.Module):
    def __init__(self, args) -> None:
        super().__init__()
        self.args = args
        self.eiko_weight = args.criteria["eiko_weight"]
        self.sdf_weight = args.criteria["sdf_weight"]
        self.fs_weight = args.criteria["fs_weight"]
        self.truncation = args.criteria["sdf_truncation"]
        self.max_dpeth = args.data_specs["max_depth"]

    def forward(self, outputs, obs=None, pointsCos=None, use_color_loss=True, use_depth_loss=True, compute_sdf_loss=True,
                weight_depth_loss=False, compute_eikonal_loss=False):
        if compute_eikonal_loss:
            raise NotImplementedError("eikonal loss is not on the reference's hot path (criterion.py:18 default False)")
        lv = outputs["_engine"].loss_value(outputs["_cfg"])
        loss_dict = {"fs_loss": lv["fs_loss"], "sdf_loss": lv["sdf_loss"], "loss": lv["loss"]}
        return torch.tensor(lv["loss"], dtype=torch.float32), loss_dict
