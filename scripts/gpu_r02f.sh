#!/usr/bin/env bash
# round 2: full parity suite (chained decoder v1 + prefetch, two-launch selection), bench incl. API path
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r02f}
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/${TAG}_pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}); print(d["pose_refine"]); print(d["api_path"]); print(d["parity"]["ok"])
PY
tail -3 $OUT/${TAG}_bench.err
