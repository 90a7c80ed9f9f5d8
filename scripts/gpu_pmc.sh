#!/usr/bin/env bash
# PMC passes (own runs, --kernel-trace only as the pool requires): MFMA utilisation and HBM traffic per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-pmc}
run() { # name, counters...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path --no-large-map > /tmp/pmc_${TAG}_$name.log 2>&1; echo "$name rc=$?" )
  ls /tmp/pmc_${TAG}_$name | head -5
  cp /tmp/pmc_${TAG}_$name/p_counter_collection.csv $OUT/${TAG}_${name}_counters.csv 2>/dev/null
  tail -2 /tmp/pmc_${TAG}_$name.log
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
# (round 5: the default decoder arithmetic issues fp16 matrix instructions - their own counter, in its own pass so that an unknown counter name cannot take the others down)
run mfma16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<'PY'
import csv, collections, glob, os
out = os.environ.get("OUT", "gpurun_out")
for f in sorted(glob.glob(f"{out}/*_counters.csv")):
    rows = list(csv.DictReader(open(f)))
    if not rows: print(f, "empty"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(f))
    for k, d in agg.items():
        if not any(s in k for s in ("k_decoder", "k_trilinear", "k_ray", "k_gather", "k_sample", "k_adam_emb", "k_reduce")): continue
        print("  ", k.ljust(36), {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
