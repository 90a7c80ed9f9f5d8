#!/usr/bin/env bash
# round-6 GPU call: `scripts/gpu_r06.sh TAG "<pytest selection or ->" [probe args | -] [bench flags | -]`
#   selected -m gpu tests (durations), scripts/decoder_layout_probe.py, optionally a bench line; everything under gpurun_out/TAG_*
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r06}; SEL=${2:-tests}; PROBE=${3:--}; BENCH=${4:--}
if [ "$SEL" != "-" ]; then
  t0=$(date +%s)
  timeout ${PYTEST_TIMEOUT:-1500} python -m pytest $SEL -m gpu -q --tb=short -p no:cacheprovider --durations=8 ${PYTEST_ARGS:-} > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -${TAIL:-30} $OUT/${TAG}_pytest_gpu.log | cut -c1-400
fi
if [ "$PROBE" != "-" ]; then
  timeout 600 python scripts/decoder_layout_probe.py $PROBE > $OUT/${TAG}_decoder_layouts.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $OUT/${TAG}_decoder_layouts.txt | tail -60
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
PY
fi
