#!/usr/bin/env bash
# round 2, second GPU call: parity suite after the API-path / fp64 pose-gradient / skip-mode changes, the new bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r02b}
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/${TAG}_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; tail -c 6000 $OUT/${TAG}_bench.json; tail -5 $OUT/${TAG}_bench.err
