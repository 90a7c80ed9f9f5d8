#!/usr/bin/env bash
# paired-sets decoder (gemm mode 5): parity subset, A/B bench against mode 1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "iteration_matches and (1-5 or frozen-5 or kitti_1f_1it-5)" > $OUT/pair_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pair_pytest.log | cut -c1-220
SETTINGS="${SETTINGS:-NL_GEMM_MODE=1;NL_GEMM_MODE=5}" TESTS="nothing_selected" bash scripts/gpu_ab.sh pair_ab 2>&1 | tail -3
