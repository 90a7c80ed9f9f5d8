#!/usr/bin/env bash
# PMC passes for the two geometry / scatter kernels of the bench that run far below the HBM roofline: where do k_trilinear_bwd and
# k_ray_intersect_q spend their cycles?  LDS traffic and conflicts, VALU issue, L2 atomics.  Each pass in its own run (--kernel-trace + --pmc only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-scat}
run() { name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path --no-large-map --no-settings --no-pmc > /tmp/pmc_${TAG}_$name.log 2>&1; echo "$name rc=$?" )
  cp /tmp/pmc_${TAG}_$name/p_counter_collection.csv $OUT/${TAG}_${name}_counters.csv 2>/dev/null; tail -1 /tmp/pmc_${TAG}_$name.log | cut -c1-160; }
run a SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
run b SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN
run c SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run d TCC_ATOMIC_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum TCP_TOTAL_ATOMIC_WITH_RET_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
TAG=$TAG python - <<'PY'
import csv, collections, glob, os, re
out = "gpurun_out"; TAG = os.environ["TAG"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out}/{TAG}_*_counters.csv")):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]); m_ = re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)", k); k = m_.group(1) if m_ else k[:30]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
def big(d, name):
    v = d.get(name, [])
    if not v: return float("nan")
    m = max(v); sel = [x for x in v if x >= 0.5 * m] or v
    return sum(sel) / len(sel)
lines = []
for k, d in sorted(agg.items()):
    if not (k.startswith("k_trilinear_bwd") or k.startswith("k_ray_intersect_q") or k.startswith("k_sample") or k.startswith("k_gather")): continue
    cyc = big(d, "GRBM_GUI_ACTIVE") / 8                      # (8 XCDs count the launch's cycles)
    simd_cyc = 1024 * cyc                                     # SIMD-cycles of the launch
    waves = big(d, "SQ_WAVES")
    lines.append(f"== {k}: {cyc:.0f} cycles per launch, {waves:.0f} waves")
    lines.append(f"   VALU instructions / wave {big(d, 'SQ_INSTS_VALU') / waves:8.0f}   LDS instructions / wave {big(d, 'SQ_INSTS_LDS') / waves:8.0f}   SALU / wave {big(d, 'SQ_INSTS_SALU') / waves:8.0f}"
                 f"   VMEM rd / wr per wave {big(d, 'SQ_INSTS_VMEM_RD') / waves:6.0f} / {big(d, 'SQ_INSTS_VMEM_WR') / waves:6.0f}")
    lines.append(f"   share of SIMD cycles:  VALU issue (4 cycles each) {100 * 4 * big(d, 'SQ_INSTS_VALU') / simd_cyc:5.1f} %   SQ_ACTIVE_INST_VALU {100 * big(d, 'SQ_ACTIVE_INST_VALU') / simd_cyc:5.1f} %"
                 f"   SQ_ACTIVE_INST_LDS {100 * big(d, 'SQ_ACTIVE_INST_LDS') / simd_cyc:5.1f} %   any instruction active {100 * big(d, 'SQ_ACTIVE_INST_ANY') / simd_cyc:5.1f} %")
    lines.append(f"   wave-cycles: {big(d, 'SQ_WAVE_CYCLES'):.3g} (occupancy {big(d, 'SQ_WAVE_CYCLES') / simd_cyc:4.2f} waves per SIMD on average); waiting on any instruction {100 * big(d, 'SQ_WAIT_INST_ANY') / big(d, 'SQ_WAVE_CYCLES'):5.1f} % of them,"
                 f" on an LDS instruction {100 * big(d, 'SQ_WAIT_INST_LDS') / big(d, 'SQ_WAVE_CYCLES'):5.1f} %")
    lines.append(f"   LDS: SQ_LDS_IDX_ACTIVE {big(d, 'SQ_LDS_IDX_ACTIVE'):.3g} cycles, bank conflicts {big(d, 'SQ_LDS_BANK_CONFLICT'):.3g} ({100 * big(d, 'SQ_LDS_BANK_CONFLICT') / max(big(d, 'SQ_LDS_IDX_ACTIVE'), 1):4.1f} % of the active cycles),"
                 f" address conflicts {big(d, 'SQ_LDS_ADDR_CONFLICT'):.3g}, LDS atomics with return {big(d, 'SQ_LDS_ATOMIC_RETURN'):.3g}")
    lines.append(f"   L2: atomics {big(d, 'TCC_ATOMIC_sum'):.3g} (from the CUs: {big(d, 'TCP_TOTAL_ATOMIC_WITHOUT_RET_sum'):.3g} without / {big(d, 'TCP_TOTAL_ATOMIC_WITH_RET_sum'):.3g} with return), requests {big(d, 'TCC_REQ_sum'):.3g},"
                 f" hits {big(d, 'TCC_HIT_sum'):.3g}, misses {big(d, 'TCC_MISS_sum'):.3g}")
open(f"{out}/{TAG}_scatter_intersect_counters.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
