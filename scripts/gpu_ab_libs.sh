#!/usr/bin/env bash
# same-box A/B of library builds: every ab_libs/*.so (built here from variants of the sources) takes the product's place in turn (NL_LIB_PATH),
# ROUNDS times in alternation; one bench line each (box-to-box spread is ~3 %, run-to-run on one box ~0.5 %)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in ab_libs/*.so; do
    NL_LIB_PATH=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-parity --no-api-path --no-large-map --no-settings --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
st = {h['stage']: h['avg_ms'] for h in r['hbm']}
print('%-16s ms/step %.4f  sustained %.4f  decoder %.4f  dW2 %.4f  scatter %.4f  intersect %.4f  pose-refine %.4f' % ('$(basename $lib .so)', d['ms_per_step'], d['steady_state']['ms_per_step'], r['avg_launch_ms'], r['second_kernel']['avg_launch_ms'], st['scatter'], st['intersect'], d['pose_refine']['ms_per_step_one_c_call']))"
  done
done
