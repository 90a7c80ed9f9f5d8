"""GPU (-m gpu): Criterion.forward on caller tensors (nl_criterion_forward / nl_criterion_backward through the class the reference's callers
hold) against the oracle's restatement of /root/reference/src/criterion.py:16-115 and against the formula itself written in torch with
autograd (the reference's own arithmetic: masks :66-82, weights :84-88, the two means :96-99, the weighted sum :46-47)."""
import types

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _args(tau=0.3, sdf_w=10000.0, fs_w=1.0, max_depth=50.0):
    return types.SimpleNamespace(criteria=dict(eiko_weight=0.0, sdf_weight=sdf_w, fs_weight=fs_w, sdf_truncation=tau),
                                 data_specs=dict(max_depth=max_depth))


def _case(seed, N, R, S, far=False):
    rng = np.random.default_rng(seed)
    pts = rng.normal(0, 8, (N, 3)).astype(np.float32)
    if far:
        pts[: N // 8] *= 20                                          # beyond max_depth: depth mask off
    cos = rng.uniform(0.3, 1.0, N).astype(np.float32)
    ray_mask = np.zeros(N, bool); ray_mask[rng.choice(N, R, replace=False)] = True
    d = np.linalg.norm(pts[ray_mask], axis=1)
    z = (d[:, None] + rng.normal(0, 0.4, (R, S))).astype(np.float32)   # samples around the surface: all three regions populated
    valid = rng.random((R, S)) < 0.8
    z[~valid] = 80.0                                                  # the padded slots of render_rays
    sdf = rng.normal(0, 0.5, (R, S)).astype(np.float32)
    return pts, cos, ray_mask, z, valid, sdf


def _torch_reference(sdf, z_vals, valid, pts, cos, a):
    """criterion.py:33-47, 59-100 retyped (l2, no eikonal)"""
    gt = torch.norm(pts, 2, -1) * cos.view(-1)
    z = z_vals * cos.view(-1, 1)
    depth = gt.unsqueeze(-1).expand(*z.shape)
    tau = a.criteria["sdf_truncation"]
    front = torch.where(z < depth - tau, torch.ones_like(z), torch.zeros_like(z))
    back = torch.where(z > depth + tau, torch.ones_like(z), torch.zeros_like(z))
    dm = torch.where((depth > 0.0) & (depth < a.data_specs["max_depth"]), torch.ones_like(depth), torch.zeros_like(depth))
    sm = (1.0 - front) * (1.0 - back) * dm
    nf, ns = torch.count_nonzero(front).float(), torch.count_nonzero(sm).float()
    wf, ws = 1.0 - nf / (nf + ns), 1.0 - ns / (nf + ns)
    fs = torch.mean(torch.square(sdf * front * valid - front)) * wf
    sd = torch.mean(torch.square((z + sdf * tau) * sm * valid - depth * sm)) * ws
    return a.criteria["fs_weight"] * fs + a.criteria["sdf_weight"] * sd, fs, sd


@pytest.mark.parametrize("seed,N,R,S,far", [(1, 400, 300, 40, False), (2, 5000, 4096, 96, True), (3, 64, 64, 7, False)])
def test_criterion_on_caller_tensors_matches_the_reference_formula(seed, N, R, S, far):
    from nerf_loam_amd.criterion import Criterion
    a = _args()
    pts, cos, ray_mask, z, valid, sdf = _case(seed, N, R, S, far)
    dev = "cuda"
    t = lambda x: torch.from_numpy(x).to(dev)
    sdf_t = t(sdf).requires_grad_(True)
    outputs = dict(sdf=sdf_t, z_vals=t(z), ray_mask=t(ray_mask), valid_mask=t(valid), sampled_xyz=None)
    crit = Criterion(a)
    loss, ld = crit(outputs, t(pts), t(cos).view(-1, 1))
    (3.0 * loss).backward()                                           # a non-unit upstream gradient
    g = sdf_t.grad.cpu().numpy() / 3.0
    # the oracle (numpy restatement, fp32)
    o_loss, o_dsdf, st = O.sdf_loss(z, sdf, valid, pts[ray_mask], cos[ray_mask], O.LossCfg(truncation=0.3, sdf_weight=10000.0, fs_weight=1.0, max_depth=50.0))
    assert abs(ld["loss"] - float(o_loss)) <= 2e-5 * abs(float(o_loss))
    assert abs(ld["fs_loss"] - float(st["fs_loss"])) <= 2e-5 * abs(float(st["fs_loss"])) + 1e-12
    assert abs(ld["sdf_loss"] - float(st["sdf_loss"])) <= 2e-5 * abs(float(st["sdf_loss"])) + 1e-12
    assert float(loss) == pytest.approx(ld["loss"], rel=1e-7)
    scale = np.abs(o_dsdf).max()
    assert np.abs(g - o_dsdf).max() <= 2e-5 * scale
    assert np.array_equal(g != 0, o_dsdf != 0)                         # the same samples carry gradient
    # the formula in torch with autograd, on the same device tensors
    sdf_r = t(sdf).requires_grad_(True)
    r_loss, r_fs, r_sd = _torch_reference(sdf_r, t(z), t(valid).float(), t(pts)[t(ray_mask)], t(cos)[t(ray_mask)], a)
    r_loss.backward()
    assert float(loss) == pytest.approx(float(r_loss), rel=2e-5)
    assert ld["fs_loss"] == pytest.approx(float(r_fs), rel=2e-5) and ld["sdf_loss"] == pytest.approx(float(r_sd), rel=2e-5)
    assert np.abs(g - sdf_r.grad.cpu().numpy()).max() <= 2e-5 * scale


def test_criterion_rejects_what_it_cannot_do():
    from nerf_loam_amd.criterion import Criterion
    crit = Criterion(_args())
    pts, cos, ray_mask, z, valid, sdf = _case(5, 50, 40, 8)
    t = lambda x: torch.from_numpy(x)
    cpu_out = dict(sdf=t(sdf), z_vals=t(z), ray_mask=t(ray_mask), valid_mask=t(valid))
    with pytest.raises(RuntimeError, match="no CPU path"):
        crit(cpu_out, t(pts), t(cos))
    with pytest.raises(NotImplementedError):
        crit(cpu_out, t(pts), t(cos), compute_eikonal_loss=True)
    dev_out = {k: v.cuda() for k, v in cpu_out.items()}
    loss, ld = crit(dev_out, t(pts).cuda(), t(cos).cuda(), compute_sdf_loss=False)
    assert loss == 0 and ld == {"loss": 0}
    with pytest.raises(ValueError):
        crit(dict(dev_out, z_vals=dev_out["z_vals"][:, :-1]), t(pts).cuda(), t(cos).cuda())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_criterion_on_caller_tensors_matches_the_reference_class(golden_dir, tag):
    """tests/golden/criterion.npz: the reference's OWN Criterion.forward + autograd on these tensors (make_golden.py run_criterion_case)"""
    import os
    from nerf_loam_amd.criterion import Criterion
    g = np.load(os.path.join(golden_dir, "criterion.npz"))
    N = int(g[f"{tag}_n"])
    ray_mask = np.unpackbits(g[f"{tag}_ray_mask"])[:N].astype(bool)
    z = g[f"{tag}_z_vals"]; R, S = z.shape
    valid = np.unpackbits(g[f"{tag}_valid"], axis=-1)[:, :S].astype(bool)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    sdf_t = t(g[f"{tag}_sdf"]).requires_grad_(True)
    loss, ld = Criterion(_args())(dict(sdf=sdf_t, z_vals=t(z), ray_mask=t(ray_mask), valid_mask=t(valid), sampled_xyz=None),
                                  t(g[f"{tag}_points"]), t(g[f"{tag}_cos"]).view(-1, 1))
    loss.backward()
    assert ld["loss"] == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
    assert ld["fs_loss"] == pytest.approx(float(g[f"{tag}_fs_loss"]), rel=2e-5) and ld["sdf_loss"] == pytest.approx(float(g[f"{tag}_sdf_loss"]), rel=2e-5)
    ref = g[f"{tag}_dsdf"]
    got = sdf_t.grad.cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max() and np.array_equal(got != 0, ref != 0)
