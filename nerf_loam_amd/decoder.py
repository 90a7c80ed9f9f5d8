"""Decoder: the SDF MLP container.

Mirror of /root/reference/src/variations/lidar.py:80-131 for the configuration every shipped config
uses (depth 2, width 256, in_dim 16, skips [], embedder none => Linear(16,256)-ReLU-Linear(256,256)-
ReLU-Linear(256,1)); parameter names match (`pts_linears.{0,1}`, `sdf_out`), so reference
checkpoints / state_dicts load.  forward()/get_values() run the hand-written MFMA forward kernel
(nl_decoder.hip k_decoder_fwd) - no PyTorch GEMM path; other configurations raise."""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops


class Decoder(nn.Module):
    def __init__(self, depth=2, width=256, in_dim=16, sdf_dim=128, skips=(), multires=0, embedder="none", point_dim=3,
                 local_coord=False, **kwargs):
        super().__init__()
        if depth != 2 or width != 256 or in_dim != 16 or len(skips) != 0 or embedder != "none":
            raise NotImplementedError("the MI355X decoder kernels implement the reference's shipped configuration only: "
                                      "depth 2, width 256, in_dim 16, skips [], embedder none")
        self.D, self.W, self.skips, self.point_dim = depth, width, list(skips), point_dim
        self.pts_linears = nn.ModuleList([nn.Linear(in_dim, width), nn.Linear(width, width)])
        self.sdf_out = nn.Linear(width, 1)

    def param_list(self):
        return [self.pts_linears[0].weight, self.pts_linears[0].bias, self.pts_linears[1].weight, self.pts_linears[1].bias,
                self.sdf_out.weight, self.sdf_out.bias]

    def flat_params(self, device):
        """decoder parameter block of include/nerfloam_hip.h (W1 b1 W2 b2 W3 b3)"""
        return torch.cat([p.detach().reshape(-1).float() for p in self.param_list()]).to(device).contiguous()

    @torch.no_grad()
    def load_flat(self, flat):
        off = 0
        for p in self.param_list():
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n

    @torch.no_grad()
    def get_values(self, x):
        if not x.is_cuda:
            raise L.NerfLoamHipError("Decoder.get_values needs a CUDA (HIP) tensor - no CPU path")
        x = x.detach().float().contiguous()
        # the device parameter block + W2 operand planes, kept until somebody writes the parameters (tensor version counters): the same
        # cache bundle_adjust_frames / track_frame use (render_helpers._decoder_device) - no re-flattening, no plane rebuild per call
        from .render_helpers import _decoder_device
        dec = _decoder_device(self, x.device)
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        ops.decoder_forward(x, dec.params, dec.W2T, x.shape[0], out, L.lib().nl_decoder_grid_hint())
        return out.unsqueeze(-1)

    def forward(self, inputs):
        return {"sdf": self.get_values(inputs)}
