#!/usr/bin/env bash
# chained decoder, quick look: phase stamps + one bench line per gemm mode (no parity run)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-chainq}
NL_GEMM_MODE=3 timeout 300 python scripts/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_phases.log
SETTINGS="${SETTINGS:-NL_GEMM_MODE=1;NL_GEMM_MODE=3}" TESTS="nothing_selected" bash scripts/gpu_ab.sh ${TAG}_ab 2>&1 | tail -3
