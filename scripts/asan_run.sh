#!/usr/bin/env bash
# AddressSanitizer passes with ab_libs/libnerfloam_hip_asan.so (scripts/asan_build.py: host AND device code instrumented).
#   scripts/asan_run.sh host    build container, CPU: the host code of the library (flat-array octree with resumed descents, C-ABI argument
#                               handling) under the ASan runtime through the CPU tests -> profiles/r04_a_asan_host.txt (0 reports)
#   scripts/asan_run.sh device  GPU box: what a device-side pass does in this image -> profiles/r04_a_asan_device_attempt.txt: the ASan runtime's
#                               HSA interceptors need the instrumented ROCr / HIP runtime (/opt/rocm/lib/asan), which the image does not ship -
#                               every process aborts at its first device allocation.  Each step runs under a short timeout.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
LIB=$PWD/ab_libs/libnerfloam_hip_asan.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
[ -f "$LIB" ] || { echo "no $LIB - run python scripts/asan_build.py first"; exit 1; }
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0
if [ "${1:-host}" = "host" ]; then
  LD_PRELOAD=$RT NL_LIB_PATH=$LIB timeout 900 python -m pytest tests/test_octree_host.py tests/test_api_host.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tee /tmp/asan_host.log | tail -3
  echo "AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' /tmp/asan_host.log)"
else
  echo "--1 python under the ASan runtime"; LD_PRELOAD=$RT python -c "print('plain ok')"; echo rc=$?
  echo "--2 first device allocation"; HSA_XNACK=1 LD_PRELOAD=$RT timeout 120 python -c "import torch; x = torch.ones(4, device='cuda'); print(x.sum().item())" 2>&1 | tail -6
  echo "--3 the library's smoke test"; HSA_XNACK=1 LD_PRELOAD=$RT NL_LIB_PATH=$LIB timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
fi
