#!/usr/bin/env bash
# round 2, fourth GPU call: the register-chained decoder (gemm mode 3 / 4): parity suite, A/B against modes 1 / 2
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r02d}
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "iteration_matches or get_scores or fused_intersect" > $OUT/${TAG}_pytest_first.log 2>&1; echo "pytest(first) rc=$?"; tail -25 $OUT/${TAG}_pytest_first.log | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/${TAG}_pytest_gpu.log | cut -c1-300
SETTINGS="NL_GEMM_MODE=1;NL_GEMM_MODE=3;NL_GEMM_MODE=2;NL_GEMM_MODE=4" TESTS="nothing_selected" bash scripts/gpu_ab.sh ${TAG}_ab 2>&1 | tail -5
