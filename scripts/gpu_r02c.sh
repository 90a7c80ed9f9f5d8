#!/usr/bin/env bash
# round 2, third GPU call: parity suite again (oracle fp64 weight-grad sums, flip-tolerant bars), micro-benchmark of the
# register-chained decoder's weight stream (scripts/micro/chain_stream.hip), bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r02c}
timeout 300 scripts/micro/chain_stream 2>&1 | tee $OUT/${TAG}_chain_stream.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/${TAG}_pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r02c_bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}); print(d["pose_refine"]); print(d["api_path"]); print(d["cpu_baseline"]); print(d["parity"])
print([(e["stage"], round(e["avg_ms"],4), round(e["frac"],3)) for e in d["roofline"]["hbm"]]); print(d["roofline"]["end_to_end"]["frac"])
PY
tail -3 $OUT/${TAG}_bench.err
