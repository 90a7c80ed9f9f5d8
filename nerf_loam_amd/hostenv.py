"""Host-side environment checks of the launch path (no reference counterpart: the reference leaves thread pools at their defaults).

Why it matters here and not in the reference: an iteration of this package is ~7-16 kernel launches issued by ONE host thread that must stay
ahead of a device finishing an iteration in 0.1-1.6 ms.  In a container with a CPU quota (cgroup `cpu.max`) far below the visible core count -
the MI355X boxes this was developed on show 256 cores and grant 16 - numpy's OpenBLAS pool starts 64 threads and torch's OpenMP pool 128; they
spin after every parallel region, the cgroup's quota runs out and the kernel throttles the WHOLE process, the launching thread included, for up to
~80 ms of a 100 ms period (`scripts/sync_probe.py`, `profiles/r05_sync_probe.txt`: 8-17 of 40 twenty-millisecond blocks of pose-refinement steps
came back up to 76 ms late after a numpy GEMM with the default pools, none with <= 16 threads)."""
import os


def host_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), else the visible core count"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return float(os.cpu_count() or 1)


def cap_host_thread_pools(threads=None):
    """Limit numpy's BLAS pool and torch's intra-op pool to `threads` (default: the CPU quota minus two, at most 16) in a running process.
    Call it once before a Mapping / Tracking loop when the process also runs multi-threaded numpy / torch-CPU work under a CPU quota.
    Returns the limit applied.  (A fresh process can set OPENBLAS_NUM_THREADS / OMP_NUM_THREADS instead - bench.py does both.)"""
    n = int(threads) if threads else max(1, min(16, int(host_cpu_quota()) - 2))
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(limits=n, user_api="blas")
    except Exception:                                              # noqa: BLE001 - no threadpoolctl / no BLAS loaded: nothing to limit
        pass
    import torch
    torch.set_num_threads(n)
    return n


def pools_exceed_quota():
    """(blas_threads, torch_threads, quota) when a host pool is larger than the container's CPU quota, else None - a cheap check for a start-up log"""
    quota = host_cpu_quota()
    blas = 0
    try:
        import threadpoolctl
        blas = max([int(i.get("num_threads", 0)) for i in threadpoolctl.threadpool_info() if i.get("user_api") == "blas"] or [0])
    except Exception:                                              # noqa: BLE001
        pass
    import torch
    tt = int(torch.get_num_threads())
    return (blas, tt, quota) if max(blas, tt) > quota else None


_warned = False


def warn_once_if_pools_exceed_quota():
    """Mapping / Tracking call this when they are constructed: one warning per process if a host thread pool is larger than the CPU quota"""
    global _warned
    if _warned:
        return
    _warned = True
    try:
        over = pools_exceed_quota()
    except Exception:                                              # noqa: BLE001 - a diagnostic must not break construction
        return
    if over:
        import warnings
        warnings.warn(f"nerf_loam_amd: host thread pools (BLAS {over[0]}, torch {over[1]} threads) exceed the container's CPU quota ({over[2]:g}): multi-threaded host work "
                      "between iterations can get the launching thread throttled for tens of ms - see nerf_loam_amd.hostenv.cap_host_thread_pools", RuntimeWarning, stacklevel=3)
