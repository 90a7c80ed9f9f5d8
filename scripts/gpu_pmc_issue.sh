#!/usr/bin/env bash
# PMC pass for the instruction-issue picture of every kernel of the bench: VALU / SALU / LDS / VMEM instruction counts per launch against
# the launch's cycles (own run, --kernel-trace only).  VALU issue share = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-issue}
run() { name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path --no-large-map > /tmp/pmc_${TAG}_$name.log 2>&1; echo "$name rc=$?" )
  cp /tmp/pmc_${TAG}_$name/p_counter_collection.csv $OUT/${TAG}_${name}_counters.csv 2>/dev/null; tail -2 /tmp/pmc_${TAG}_$name.log | cut -c1-200; }
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES
TAG=$TAG python - <<'PY'
import csv, collections, glob, os, re
out = "gpurun_out"; TAG = os.environ["TAG"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out}/{TAG}_*_counters.csv")):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]); m_ = re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)", k); k = m_.group(1) if m_ else k[:30]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-32s %9s %10s %8s %8s %8s %8s %9s" % ("kernel (largest launches)", "cycles", "VALU/wave", "valu%", "lds%", "LDS", "VMEM", "waves"))
for k, d in sorted(agg.items()):
    if not k.startswith("k_"): continue
    def big(name):
        v = d.get(name, [])
        if not v: return 0.0
        m = max(v); sel = [x for x in v if x >= 0.5 * m]
        return sum(sel) / len(sel)
    cyc = big("GRBM_GUI_ACTIVE") / 8
    if cyc <= 0: continue
    valu, lds, vm, waves = big("SQ_INSTS_VALU"), big("SQ_INSTS_LDS"), big("SQ_INSTS_VMEM_RD") + big("SQ_INSTS_VMEM_WR"), big("SQ_WAVES")
    print("%-32s %9.0f %10.0f %7.1f%% %7.1f%% %8.0f %8.0f %9.0f" % (k[:32], cyc, valu / max(waves, 1), 100 * valu * 4 / (1024 * cyc), 100 * big("SQ_ACTIVE_INST_LDS") / (1024 * cyc) if big("SQ_ACTIVE_INST_LDS") else 0, lds / max(waves, 1), vm / max(waves, 1), waves))
PY
