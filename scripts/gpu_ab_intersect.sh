#!/usr/bin/env bash
# same-box A/B of library builds on the intersect alone (scripts/intersect_probe.py) + the pose-refine step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp nerf_loam_amd/libnerfloam_hip.so /tmp/product.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in ab_libs/*.so; do
    cp $lib nerf_loam_amd/libnerfloam_hip.so
    echo "== $(basename $lib .so)"
    timeout 200 python scripts/intersect_probe.py 2>&1 | grep "N=" | cut -c1-60
  done
done
cp /tmp/product.so nerf_loam_amd/libnerfloam_hip.so
