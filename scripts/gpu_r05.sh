#!/usr/bin/env bash
# round-5 GPU call: `scripts/gpu_r05.sh TAG "<pytest selection or ->" [bench flags | -]` - micro/f16_pair when built, the selected -m gpu tests
# (durations), optionally a bench line; everything under gpurun_out/TAG_*
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r05}; SEL=${2:-tests}; BENCH=${3:--}
if [ -x scripts/micro/f16_pair ] && [ "${MICRO:-1}" = 1 ]; then timeout 120 scripts/micro/f16_pair > $OUT/${TAG}_f16_pair.txt 2>&1; cat $OUT/${TAG}_f16_pair.txt; fi
if [ "$SEL" != "-" ]; then
  t0=$(date +%s)
  timeout 1500 python -m pytest $SEL -m gpu -q --tb=short -p no:cacheprovider --durations=8 ${PYTEST_ARGS:-} > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -${TAIL:-30} $OUT/${TAG}_pytest_gpu.log | cut -c1-400
fi
if [ "$BENCH" != "-" ]; then
  t1=$(date +%s)
  timeout 900 python bench.py $BENCH > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$? ($(( $(date +%s) - t1 )) s)"; tail -3 $OUT/${TAG}_bench.err
  python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step %.4f  value %.3e  decoder %.4f ms frac %.3f" % (d["ms_per_step"], d["value"], r["avg_launch_ms"], r["frac"]))
print("second", r["second_kernel"]["avg_launch_ms"], "stages", {k: round(v, 4) for k, v in r["end_to_end"]["stage_ms"].items()})
if "parity" in d: print("parity", {k: v for k, v in d["parity"].items() if k != "bars"})
if "pose_refine" in d: print("pose_refine", d["pose_refine"])
if "large_map" in d:
    lm = d["large_map"]
    print("large_map", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a in ("ms_per_iter", "ok", "geometry_bit_exact")}) for k, v in lm.items() if k != "track_2048"})
    for k, v in lm.get("track_2048", {}).items(): print("track_2048", k, {a: b for a, b in v.items() if a in ("ms_per_step", "samples_per_hit_ray", "max_samples_per_ray", "valid_samples", "parity_vs_oracle")})
PY
fi
