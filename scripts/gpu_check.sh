#!/usr/bin/env bash
# Run on the GPU box through gpurun: parity tests, smoke, bench, rocprof kernel stats.
# Every step is bounded by its own timeout and writes under gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-run}
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -4
python - <<'PY' 2>&1 | tail -3
import os; print("host cores", os.cpu_count())
PY
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/${TAG}_pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $OUT/${TAG}_smoke.log
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench.json; tail -5 $OUT/${TAG}_bench.err
echo "== rocprof kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path > /tmp/prof_$TAG.log 2>&1 ; echo "rocprof rc=$?" )
find /tmp/prof_$TAG -type f | head -10
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/${TAG}_kernel_stats.csv; head -30 $f; done
tail -3 /tmp/prof_$TAG.log
