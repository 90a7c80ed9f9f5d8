#!/usr/bin/env bash
# same-box A/B of library builds in ab_libs/*.so on the decoder layout probe: `scripts/gpu_ab_layout.sh TAG [probe args]`
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-ab}; shift || true
for round in 1 2; do
  for lib in ab_libs/*.so; do
    echo "== $(basename $lib .so) (round $round)"
    NL_LIB_PATH=$PWD/$lib timeout 300 python scripts/decoder_layout_probe.py ${@:-131072} 2>&1 | grep -v amdgpu.ids | grep -E "layout 2|total/tile|span|H/I|E:dH2|C:loop|F:loop" | grep -v "layout 1"
  done
done 2>&1 | tee $OUT/${TAG}_ab_layout.txt
