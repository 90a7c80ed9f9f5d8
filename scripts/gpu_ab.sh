#!/usr/bin/env bash
# A/B of the decoder tilings / stagger settings: parity subset + bench of each
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-ab}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "iteration or three_steps or hipgraph" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log
run() {
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$*: ms/step %.3f  decoder %.3f ms (%.1f TF, %.3f)  wgrad2 %.3f ms  pose-refine %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], r['second_kernel']['avg_launch_ms'], d['pose_refine']['ms_per_step_eager']))"
}
run NL_DECODER_VARIANT=0
for s in ${STAGGERS:-0 15000 29000 45000}; do run NL_DECODER_VARIANT=1 NL_DECODER_STAGGER=$s; done
