#!/usr/bin/env bash
# round-3 GPU loop: `scripts/gpu_r03.sh TAG "pytest selection" [bench]` - selected GPU tests (log + durations), optionally the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r3}
SEL=${2:-tests}
timeout 1500 python -m pytest $SEL -m gpu -q --tb=short -p no:cacheprovider --durations=15 > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/${TAG}_pytest_gpu.log
if [ "${3:-}" = "bench" ]; then
  timeout 900 python bench.py --steps 20 --warmup 3 2>$OUT/${TAG}_bench.err | tee $OUT/${TAG}_bench.json | cut -c1-600
fi
