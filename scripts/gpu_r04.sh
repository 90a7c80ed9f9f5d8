#!/usr/bin/env bash
# round-4 validation call: `scripts/gpu_r04.sh TAG [pytest selection]` - the -m gpu suite (log + durations), smoke(), the DEFAULT bench line
# (in-run PMC passes, settings legs, CPU baseline: its wall time is printed), everything under gpurun_out/TAG_*
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r04}
SEL=${2:-tests}
t0=$(date +%s)
timeout 1200 python -m pytest $SEL -m gpu -q --tb=short -p no:cacheprovider --durations=10 > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -25 $OUT/${TAG}_pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/${TAG}_smoke.log
t1=$(date +%s)
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$? ($(( $(date +%s) - t1 )) s)"; tail -3 $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step %.4f  value %.3e  decoder %.4f ms frac %.3f  traffic %s (%s)" % (d["ms_per_step"], d["value"], r["avg_launch_ms"], r["frac"], r["traffic"], r["traffic_source"][:40]))
print("second", r["second_kernel"]["avg_launch_ms"], "end_to_end", r["end_to_end"]["frac"], "stages", {k: round(v, 4) for k, v in r["end_to_end"]["stage_ms"].items()})
print("parity", {k: v for k, v in d["parity"].items() if k != "bars"})
print("settings", {k: {a: b for a, b in v.items() if a != "parity"} | {"parity_ok": v.get("parity", {}).get("ok")} for k, v in d.get("settings", {}).items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["sample"][:200])
print("pose_refine", d["pose_refine"]); print("api", d["api_path"]); print("shard_probe", d["shard_probe"])
print("large_map", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a in ("ms_per_iter", "ok", "geometry_bit_exact")}) for k, v in d["large_map"].items()})
PY
