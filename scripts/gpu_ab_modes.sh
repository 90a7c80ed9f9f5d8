#!/usr/bin/env bash
# same-box A/B of the decoder arithmetic: one quick bench line per NL_GEMM_MODE in $MODES (alternating, $ROUNDS rounds)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in $(seq 1 ${ROUNDS:-2}); do
  for m in ${MODES:-3 4 5}; do
    NL_GEMM_MODE=$m timeout 300 python bench.py --no-cpu-baseline --no-api-path --no-large-map --no-settings --no-pmc ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
st = {h['stage']: h['avg_ms'] for h in r['hbm']}
p = d.get('parity', {})
print('mode $m  ms/step %.4f  sustained %.4f  decoder %.4f (frac %.3f)  dW2 %.4f  pose-refine %.4f  parity sdf %.2e dsdf %.2e dX %.2e ok=%s' % (d['ms_per_step'], d['steady_state']['ms_per_step'], r['avg_launch_ms'], r['frac'], r['second_kernel']['avg_launch_ms'], d['pose_refine']['ms_per_step_one_c_call'], p.get('sdf_max_abs_err', -1), p.get('dsdf_max_err_rel_to_max', -1), p.get('dX_rel_l2', -1), p.get('ok')))"
  done
done
