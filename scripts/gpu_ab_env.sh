#!/usr/bin/env bash
# same-box A/B of an environment switch: `scripts/gpu_ab_env.sh VAR a b` runs the bench with VAR=a and VAR=b in alternation (ROUNDS times)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; shift
for round in $(seq 1 ${ROUNDS:-2}); do
  for val in "$@"; do
    env $VAR=$val timeout 300 python bench.py --no-cpu-baseline --no-parity --no-large-map 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
st = {h['stage']: h['avg_ms'] for h in r['hbm']}
a = d['api_path']
print('%-14s ms/step %.4f  decoder %.4f  dW2 %.4f  scatter %.4f  reduce %.4f  pose-refine %.4f  rank1/8 %.4f  ba2048 %.4f (eng %.4f)  ba4096x4 %.4f' % ('$VAR=$val', d['ms_per_step'], r['avg_launch_ms'], r['second_kernel']['avg_launch_ms'], st['scatter'], st['reduce'], d['pose_refine']['ms_per_step_one_c_call'], d['shard_probe']['rank_share_1_of_8']['ms_per_step'], a['bundle_adjust_2048x1_ms_per_iter'], a['engine_2048x1_ms_per_iter'], a['bundle_adjust_4096x4_frozen_decoder_ms_per_iter']))"
  done
done
