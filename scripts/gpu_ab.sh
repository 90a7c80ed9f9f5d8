#!/usr/bin/env bash
# A/B of kernel modes: parity subset + bench of each setting given as "VAR=val ..." lines in $SETTINGS (';'-separated)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-ab}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "${TESTS:-iteration or three_steps or hipgraph or full_scan}" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -${TAIL:-3} $OUT/${TAG}_pytest.log
run() {
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$*: ms/step %.3f  decoder %.3f ms (%.1f TF, %.3f)  wgrad2 %.3f ms  pose-refine %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], r['second_kernel']['avg_launch_ms'], d['pose_refine']['ms_per_step_eager']))"
}
IFS=';' read -ra SETS <<< "${SETTINGS:-NL_WGRAD2_MODE=0;NL_WGRAD2_MODE=1}"
for s in "${SETS[@]}"; do run $s; done
