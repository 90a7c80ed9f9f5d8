#!/usr/bin/env bash
# round 2, first GPU call: full parity suite (ungated kitti/ncd goldens, gemm mode 2, full-size oracle case, reference grid
# kernels), phase probe of the decoder as it stands, A/B of gemm modes 1 and 2
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r02a}
python -c "import os; print('host cores', os.cpu_count())"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -60 $OUT/${TAG}_pytest_gpu.log
timeout 300 python scripts/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_phases.log
SETTINGS="NL_GEMM_MODE=1;NL_GEMM_MODE=2" TESTS="nothing_selected" bash scripts/gpu_ab.sh ${TAG}_ab 2>&1 | tail -4
