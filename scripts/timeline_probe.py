"""GPU timeline of the latency-bound regimes: per kernel of one iteration, the median duration and the median idle gap in front
of it, from a rocprofv3 --kernel-trace of two loops -
   section 1: the pose-refinement step of tracking (2048 rays, decoder + embeddings frozen, eager launches),
   section 2: the same step replayed as a hipGraph,
   section 3: one rank's share of an 8-GPU mapping iteration (16 384 interleaved rays, trainable decoder),
   sections 4, 5: sections 1 and 3 through one C call per iteration (nl_iteration, what the API loops use: fused launches).
Sections are separated in the trace by runs of k_pose_matrix launches (2, 3, ... of them), which occur nowhere else in the loops.

    run   : the workload (under rocprofv3; also prints the untraced-equivalent wall time per step measured by the host)
    parse : python scripts/timeline_probe.py parse <kernel_trace.csv>
"""
import csv, os, re, sys, time
from collections import Counter
from statistics import median

MARK = "k_pose_matrix"
ANCHOR = "k_ray_intersect_q"


def short(name):
    name = re.sub(r"^void ", "", name)
    if "FillFunctor<int>" in name:
        return "torch fill<int> (counters.zero_)"
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def parse(path, out=sys.stdout):
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        cols = {c.lower(): c for c in rd.fieldnames}
        cn = next(cols[c] for c in cols if "kernel_name" in c)
        cs = next(cols[c] for c in cols if c.startswith("start"))
        ce = next(cols[c] for c in cols if c.startswith("end"))
        rows = [(r[cn], int(r[cs]), int(r[ce])) for r in rd]
    rows.sort(key=lambda r: r[1])
    # split into sections at runs of marker kernels
    sections, cur, run = {}, None, 0
    for r in rows:
        if MARK in r[0]:
            run += 1
            cur = None
            continue
        if run:
            cur = sections.setdefault(run - 1, []) if run >= 2 else None     # a single launch is set_poses, not a separator
            run = 0
        if cur is not None:
            cur.append(r)
    titles = {1: "pose-refinement step, 2048 rays, eager", 2: "pose-refinement step, 2048 rays, hipGraph replay",
              3: "rank share of an 8-GPU mapping iteration: 16 384 interleaved rays, trainable decoder, eager",
              4: "pose-refinement step, 2048 rays, one C call per iteration (nl_iteration: fused launches)",
              5: "rank share of an 8-GPU mapping iteration, one C call per iteration",
              6: "rank share + the exchanges on a one-rank RCCL communicator, every collective on the launch stream (overlap=False)",
              7: "rank share + the exchanges, [pose | embedding] all-reduce on the side stream under dW2 + slab reduction (overlap=True; gaps < 0 = concurrent)",
              8: "section 6 (every collective on the launch stream) replayed as a hipGraph",
              9: "section 7 (overlapped gradient exchange) replayed as a hipGraph"}
    for sid in sorted(sections):
        ks = sections[sid]
        anchors = [i for i, r in enumerate(ks) if ANCHOR in r[0]]
        its = [ks[a:b] for a, b in zip(anchors[:-1], anchors[1:])]
        its = its[len(its) // 5:]                                   # drop the warm-up fifth
        if not its:
            continue
        n_mode = Counter(len(it) for it in its).most_common(1)[0][0]
        its = [it for it in its if len(it) == n_mode]
        prev_end = {}
        print(f"\n== section {sid}: {titles.get(sid, '')}  ({len(its)} iterations of {n_mode} kernels)", file=out)
        print(f"{'kernel':62s} {'dur us':>8s} {'gap-before us':>14s}", file=out)
        tot_d = tot_g = 0.0
        for pos in range(n_mode):
            durs = [(it[pos][2] - it[pos][1]) / 1e3 for it in its]
            if pos == 0:      # the gap in front of the anchor = to the last kernel of the previous iteration
                gaps = [(its[i][0][1] - its[i - 1][-1][2]) / 1e3 for i in range(1, len(its)) if its[i][0][1] - its[i - 1][-1][2] < 1e6]
            else:
                gaps = [(it[pos][1] - it[pos - 1][2]) / 1e3 for it in its]
            d, g = median(durs), (median(gaps) if gaps else 0.0)
            tot_d += d; tot_g += g
            print(f"{short(its[0][pos][0]):62s} {d:8.2f} {g:14.2f}", file=out)
        period = median([(its[i][0][1] - its[i - 1][0][1]) / 1e3 for i in range(1, len(its))]) if len(its) > 1 else 0.0
        print(f"{'sum of medians':62s} {tot_d:8.2f} {tot_g:14.2f}   -> {tot_d + tot_g:.1f} us;  median start-to-start period {period:.1f} us", file=out)


def run():
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nerf_loam_amd import _lib as L, pipeline as P, dist as D, ops
    L.require_gpu()
    dev = torch.device("cuda")
    w = bench.build_workload(dev)
    mark_in, mark_out = torch.zeros(1, 6, device=dev), torch.zeros(1, 12, device=dev)

    def mark(n):
        torch.cuda.synchronize()
        for _ in range(n):
            ops.pose_matrices(mark_in, mark_out)
        torch.cuda.synchronize()

    def timed(fn, n):
        for _ in range(n // 5):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # sections 1 + 2: pose refinement (bench.pose_refine_bench's step)
    rng = np.random.default_rng(3)
    sel = np.sort(rng.choice(len(w["points"]), 2048, replace=False))
    eng = P.SdfEngine(max_rays=2048, samples_per_ray_cap=96, device=dev)
    eng.set_rays(w["dirs"][sel], w["points"][sel], w["cos"][sel])
    pose = w["pose"].copy(); pose[:3] += np.array([0.03, -0.02, 0.01], np.float32)
    eng.set_poses(pose[None], [1])
    cfg = P.IterConfig(step_size=0.04)
    eng.begin_call(w["map"], None)
    fb = dict(train_decoder=False, want_emb_grad=False, want_pose_grad=True)
    op = dict(update_emb=False, update_decoder=False, update_pose=True, lr_pose=0.005 / 3)

    def eager():
        eng.forward_backward(w["map"], w["dec"], cfg, **fb)
        eng.optimiser_step(w["map"], w["dec"], cfg, **op)
    eager(); mark(2)
    print(f"section 1 host-timed: {timed(eager, 100):.4f} ms/step", flush=True)
    try:
        eng.capture_iteration(w["map"], w["dec"], cfg, **fb, **op)
        eng.replay(); mark(3)
        print(f"section 2 host-timed: {timed(eng.replay, 100):.4f} ms/step", flush=True)
    except Exception as e:                                          # noqa: BLE001
        print("graph capture failed:", repr(e)[:200])
    # section 3: rank 0 of 8, interleaved shard, full mapping iteration
    N = len(w["points"])
    order = D.interleaved_order(N, 8)
    lo, hi = D.shard_bounds(N, 0, 8)
    s = order[lo:hi]
    e2 = P.SdfEngine(max_rays=len(s), samples_per_ray_cap=48, device=dev)
    e2.set_rays(w["dirs"][s], w["points"][s], w["cos"][s]); e2.set_poses(w["pose"][None], [1])
    cfg2 = P.IterConfig()
    e2.begin_call(w["map"], w["dec"])

    def shard():
        e2.forward_backward(w["map"], w["dec"], cfg2, train_decoder=True)
        e2.optimiser_step(w["map"], w["dec"], cfg2)
    shard(); mark(4)
    print(f"section 3 host-timed: {timed(shard, 40):.4f} ms/step", flush=True)
    # sections 4 + 5: the same two steps through ONE C call per iteration (nl_iteration: the fused launches)
    eng.graph = None
    eng.bind(w["map"], w["dec"], cfg, **fb, **op)
    eng.run_bound(); mark(5)
    print(f"section 4 host-timed: {timed(eng.run_bound, 100):.4f} ms/step", flush=True)
    e2.bind(w["map"], w["dec"], cfg2, train_decoder=True)
    e2.run_bound(); mark(6)
    print(f"section 5 host-timed: {timed(e2.run_bound, 40):.4f} ms/step", flush=True)
    # sections 6 + 7: the same rank share with the exchanges of the ray-sharded iteration on a ONE-RANK RCCL communicator (what issuing them
    # costs; the messages themselves go nowhere): serial and overlapped
    try:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
        for sec, overlap in ((6, False), (7, True)):
            e3 = P.SdfEngine(max_rays=len(s), samples_per_ray_cap=48, device=dev)
            ex = D.RayShardedExchange(e3, backend="rccl", overlap=overlap)
            e3.set_rays(w["dirs"][s], w["points"][s], w["cos"][s]); e3.set_poses(w["pose"][None], [1])
            e3.begin_call(w["map"], w["dec"])
            e3.bind(w["map"], w["dec"], cfg2, train_decoder=True)
            e3.run_bound(); e3.run_bound(); mark(sec + 1)
            print(f"section {sec} host-timed: {timed(e3.run_bound, 40):.4f} ms/step  ({ex.backend}, rows: {ex._rows_cap})", flush=True)
            # the same iteration captured in a hipGraph (what a multi-GPU run replays: the host cost of issuing ~25 launches + 5 RCCL calls
            # per iteration is off the critical path)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                e3.run_bound()
            gr.replay(); mark(sec + 3)
            print(f"section {sec + 2} host-timed: {timed(gr.replay, 40):.4f} ms/step  (hipGraph replay, overlap={overlap})", flush=True)
            del gr, ex, e3
        mark(11)
        tdist.destroy_process_group()
    except Exception as e:                                          # noqa: BLE001
        print("sharded sections failed:", repr(e)[:300]); mark(11)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "parse":
        parse(sys.argv[2])
    else:
        run()
