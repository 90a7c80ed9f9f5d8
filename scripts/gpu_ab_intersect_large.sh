#!/usr/bin/env bash
# same-box A/B of library builds in ab_libs/*.so: the intersect alone on the single-scan and the 150-scan map (scripts/intersect_probe_large.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in $(seq 1 ${ROUNDS:-1}); do
  for lib in ab_libs/*.so; do
    case $lib in *asan*) continue;; esac
    echo "== $(basename $lib)"
    NL_LIB_PATH=$PWD/$lib timeout 300 python scripts/intersect_probe_large.py 2>/dev/null | cut -c1-170
  done
done
