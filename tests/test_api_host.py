"""CPU: host-side behaviour of the API mirrors.  The Criterion mirror has no CPU arithmetic - tensors that do not live on the GPU are refused, loudly (the loss of caller tensors
is computed by csrc/nl_criterion.hip; tests/test_gpu_criterion.py holds the numerics)."""
import types

import pytest
import torch


def test_criterion_refuses_cpu_tensors_and_the_eikonal_term():
    from nerf_loam_amd.criterion import Criterion
    a = types.SimpleNamespace(criteria=dict(eiko_weight=0.0, sdf_weight=10000.0, fs_weight=1.0, sdf_truncation=0.3), data_specs=dict(max_depth=50.0))
    crit = Criterion(a)
    assert (crit.sdf_weight, crit.fs_weight, crit.truncation, crit.max_dpeth) == (10000.0, 1.0, 0.3, 50.0)
    out = dict(sdf=torch.zeros(4, 3), z_vals=torch.ones(4, 3), ray_mask=torch.ones(4, dtype=torch.bool), valid_mask=torch.ones(4, 3, dtype=torch.bool))
    with pytest.raises(RuntimeError, match="no CPU path"):
        crit(out, torch.ones(4, 3), torch.ones(4))
    with pytest.raises(NotImplementedError):
        crit(out, torch.ones(4, 3), torch.ones(4), compute_eikonal_loss=True)
    with pytest.raises(ValueError):
        crit(out)                                                  # caller tensors need the observations


def test_private_seed_stream_follows_torch_manual_seed():
    """the seeds of the device-side random streams (ray subsets, sampler jitter: render_helpers._draw_seed) come from a private generator
    derived from torch's global seed: a new manual_seed value restarts it, reseed() restarts it explicitly, and drawing from it leaves the
    global CPU stream (the one the reference's LidarFrame.sample_rays consumes) untouched"""
    from nerf_loam_amd import render_helpers as RH
    torch.manual_seed(123)
    RH.reseed()
    a = [RH._draw_seed() for _ in range(4)]
    g0 = torch.random.get_rng_state().clone()
    b = [RH._draw_seed() for _ in range(4)]
    assert torch.equal(torch.random.get_rng_state(), g0) and a != b           # the global stream did not move; the private one did
    torch.manual_seed(124)
    c = [RH._draw_seed() for _ in range(4)]
    assert c != a
    torch.manual_seed(123)                                                     # a NEW value (124 -> 123): restarted by itself
    assert [RH._draw_seed() for _ in range(4)] == a
    torch.manual_seed(123)                                                     # the same value again: cannot be told from not seeding ...
    assert [RH._draw_seed() for _ in range(4)] == b
    RH.reseed()                                                                # ... hence the explicit call
    assert [RH._draw_seed() for _ in range(4)] == a


def test_host_thread_pools_can_be_capped_below_the_cpu_quota():
    """nerf_loam_amd.hostenv: the container's CPU quota is readable, and the BLAS / torch pools follow cap_host_thread_pools (a process whose
    pools exceed the quota gets its launching thread throttled: profiles/r05_sync_probe.txt)"""
    import numpy as np
    import torch
    from nerf_loam_amd import hostenv as H
    q = H.host_cpu_quota()
    assert q > 0
    before = torch.get_num_threads()
    try:
        n = H.cap_host_thread_pools(2)
        assert n == 2 and torch.get_num_threads() == 2
        (np.ones((64, 64), np.float32) @ np.ones((64, 64), np.float32)).sum()      # (a BLAS pool exists now)
        H.cap_host_thread_pools(2)
        import threadpoolctl
        assert all(i["num_threads"] <= 2 for i in threadpoolctl.threadpool_info() if i.get("user_api") == "blas")
        assert H.pools_exceed_quota() is None or q < 2
        assert H.cap_host_thread_pools() == max(1, min(16, int(q) - 2))
    finally:
        torch.set_num_threads(before)
