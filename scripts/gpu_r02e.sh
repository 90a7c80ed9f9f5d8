#!/usr/bin/env bash
# round 2, fifth GPU call: phase stamps of the register-chained decoder, parity of the ncd cases again
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r02e}
NL_GEMM_MODE=3 timeout 300 python scripts/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_phases_chain.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/${TAG}_pytest_gpu.log | cut -c1-300
