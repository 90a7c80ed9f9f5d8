#!/usr/bin/env bash
# chained-decoder iteration loop: parity subset in modes 3 / 4, phase stamps, A/B bench against mode 1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-chain}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_mirror.py -m gpu -q --tb=short -p no:cacheprovider -k "iteration_matches or full_scan or three_steps or tracking_matches or get_scores or hipgraph" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/${TAG}_pytest.log | cut -c1-300
NL_GEMM_MODE=3 timeout 300 python scripts/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_phases.log
SETTINGS="${SETTINGS:-NL_GEMM_MODE=1;NL_GEMM_MODE=3;NL_GEMM_MODE=4}" TESTS="nothing_selected" bash scripts/gpu_ab.sh ${TAG}_ab 2>&1 | tail -4
