#!/usr/bin/env bash
# round 5: the single-launch mid-size scans (k_scan_single) - tests, then same-box A/B against the two-launch path (NL_SCAN_SINGLE=0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r05_s}
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_dist_rccl.py tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/${TAG}_pytest_gpu.log | cut -c1-300
for round in 1 2; do
  for v in 0 1; do
    echo "NL_SCAN_SINGLE=$v: $(NL_SCAN_SINGLE=$v timeout 300 python scripts/timeline_probe.py run 2>&1 | grep 'host-timed' | tr '\n' ';')"
    echo "NL_SCAN_SINGLE=$v: $(NL_SCAN_SINGLE=$v timeout 300 python scripts/rank_share_probe.py 2>&1 | grep -v amdgpu.ids | tr '\n' ';')"
  done
done 2>&1 | tee $OUT/${TAG}_scan_ab.txt
