#!/usr/bin/env bash
# same-box A/B of library builds in ab_libs/*.so at the launch-bound sizes (scripts/rank_share_probe.py), ROUNDS times in alternation
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in ab_libs/*.so; do
    case $lib in *asan*) continue;; esac
    NL_LIB_PATH=$PWD/$lib timeout 300 python scripts/rank_share_probe.py 2>/dev/null | tail -1
  done
done
