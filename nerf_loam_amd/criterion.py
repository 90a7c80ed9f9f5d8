"""Criterion: free-space + truncated-surface SDF losses.

Mirror of /root/reference/src/criterion.py: same constructor (reads args.criteria / args.data_specs)
and call signature, returns (loss, loss_dict).  Two ways in:

* `outputs` from this package's `render_rays` (the optimisation loops): the masks, the data-dependent
  re-weighting (criterion.py:84-88), the squared-residual sums and dL/dsdf are produced inside the
  fused kernels (nl_geometry.hip sampler / k_loss_finalize, nl_decoder.hip phase D); this object carries
  the hyper-parameters into them and turns the device sums into the scalar the reference's callers log.
* `outputs` holding the caller's own tensors (sdf, z_vals, ray_mask, valid_mask, as the reference's
  render_rays returns them, criterion.py:24-36): the loss is computed by nl_criterion_forward and is
  differentiable with respect to outputs["sdf"] (nl_criterion_backward) - HIP kernels, no torch arithmetic.

`compute_eikonal_loss` is never enabled by any reference call site and is rejected."""
import torch
import torch.nn as nn

from . import ops


class _SdfLoss(torch.autograd.Function):
    """loss(sdf) of criterion.py:38-47 on caller tensors; the gradient flows to sdf only (z_vals come out of the sampler, which has
    none; the weights of :84-88 are counts)."""

    @staticmethod
    def forward(ctx, sdf, z_vals, valid_u8, points, cos, ray_idx, hyper):
        out = torch.empty(8, dtype=torch.float32, device=sdf.device)
        ws = torch.empty(8, dtype=torch.int32, device=sdf.device)
        sdf_c = sdf.detach().contiguous()
        ops.criterion_forward(sdf_c, z_vals, valid_u8, points, cos, ray_idx, *hyper, ws, out)
        ctx.save_for_backward(sdf_c, z_vals, valid_u8, points, cos, ray_idx, out)
        ctx.hyper = hyper
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, grad_loss, _grad_out):
        sdf_c, z_vals, valid_u8, points, cos, ray_idx, out = ctx.saved_tensors
        dsdf = torch.empty_like(sdf_c)
        g = grad_loss.detach().to(torch.float32).reshape(1).contiguous()
        ops.criterion_backward(sdf_c, z_vals, valid_u8, points, cos, ray_idx, *ctx.hyper, out, g, dsdf)
        return dsdf, None, None, None, None, None, None


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
        if "_engine" in outputs:
            lv = outputs["_engine"].loss_value(outputs["_cfg"])
            loss_dict = {"fs_loss": lv["fs_loss"], "sdf_loss": lv["sdf_loss"], "loss": lv["loss"]}
            return torch.tensor(lv["loss"], dtype=torch.float32), loss_dict
        return self._forward_tensors(outputs, obs, pointsCos, compute_sdf_loss)

    def _forward_tensors(self, outputs, obs, pointsCos, compute_sdf_loss):
        """criterion.py:24-57 on the caller's tensors.  torch does the plumbing (the row list of the hit rays, contiguity); every number is
        computed by the HIP kernels of csrc/nl_criterion.hip."""
        if obs is None or pointsCos is None:
            raise ValueError("Criterion.forward on caller tensors needs obs (points [N,3]) and pointsCos [N]")
        sdf = outputs["sdf"]
        if not sdf.is_cuda:
            raise RuntimeError("Criterion.forward: tensors must live on the GPU - this package has no CPU path")
        loss_dict = {}
        if not compute_sdf_loss:                                  # criterion.py:38: nothing else contributes (no colour / depth term exists)
            loss_dict["loss"] = 0
            return 0, loss_dict
        dev = sdf.device
        z_vals = outputs["z_vals"].detach().to(dev, torch.float32).contiguous()
        valid = outputs["valid_mask"].detach().to(dev).contiguous()
        valid_u8 = valid.view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8)
        ray_idx = torch.nonzero(outputs["ray_mask"].to(dev).reshape(-1), as_tuple=False).reshape(-1).to(torch.int32).contiguous()
        points = obs.detach().to(dev, torch.float32).reshape(-1, 3).contiguous()
        cos = pointsCos.detach().to(dev, torch.float32).reshape(-1).contiguous()
        n_rays = outputs["ray_mask"].numel()
        if points.shape[0] != n_rays or cos.numel() != n_rays:    # (the reference's boolean indexing `points[ray_mask]` raises on a mismatch)
            raise ValueError(f"Criterion.forward: ray_mask has {n_rays} entries, obs {points.shape[0]} points, pointsCos {cos.numel()}")
        if sdf.dim() != 2 or z_vals.shape != sdf.shape or valid_u8.shape != sdf.shape or ray_idx.numel() != sdf.shape[0]:
            raise ValueError("Criterion.forward: sdf, z_vals, valid_mask must be [R,S] with R = ray_mask.sum()")
        if sdf.shape[0] == 0:
            raise ValueError("Criterion.forward: no ray hit the map (the reference's render_rays returns None in that case)")
        hyper = (float(self.truncation), float(self.max_dpeth), float(self.fs_weight), float(self.sdf_weight))
        loss, out = _SdfLoss.apply(sdf.to(torch.float32), z_vals, valid_u8, points, cos, ray_idx, hyper)
        o = out[:3].tolist()                                      # the reference's three .item() calls: one read-back
        loss_dict["fs_loss"] = o[1]
        loss_dict["sdf_loss"] = o[2]
        loss_dict["loss"] = o[0]
        return loss, loss_dict
