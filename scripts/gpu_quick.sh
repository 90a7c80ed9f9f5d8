#!/usr/bin/env bash
# quick GPU loop: parity tests + phase probe + bench (no CPU baseline) + rocprof stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-q}
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/${TAG}_pytest_gpu.log
timeout 300 python scripts/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_phases.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee $OUT/${TAG}_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path > /tmp/prof_$TAG.log 2>&1 ; echo "rocprof rc=$?" )
cp /tmp/prof_$TAG/prof_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null; cut -d, -f1-4 $OUT/${TAG}_kernel_stats.csv | head -12
