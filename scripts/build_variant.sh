#!/usr/bin/env bash
# build a variant of the library into ab_libs/NAME.so: `scripts/build_variant.sh NAME "<extra hipcc flags>" [source.hip ...]` - recompiles the named
# sources (default nl_decoder.hip) with the extra flags and links them with the product's other objects (nerf_loam_amd/build/*.o, built by
# `python -m nerf_loam_amd.build`).  Same-box A/B through NL_LIB_PATH (scripts/gpu_ab_libs.sh, scripts/gpu_r06.sh).
set -eu
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=${2:-}; shift; shift || true
SRCS=${@:-nl_decoder.hip}
mkdir -p ab_libs/obj_$NAME
OBJS=""
for o in nerf_loam_amd/build/*.o; do
  b=$(basename $o .o); skip=0
  for s in $SRCS; do [ "${s%.*}" = "$b" ] && skip=1; done
  [ $skip = 1 ] || OBJS="$OBJS $o"
done
for s in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result $EXTRA -x hip -c nerf_loam_amd/csrc/$s -o ab_libs/obj_$NAME/${s%.*}.o
  OBJS="$OBJS ab_libs/obj_$NAME/${s%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ab_libs/$NAME.so
echo ab_libs/$NAME.so
