#!/usr/bin/env bash
# AddressSanitizer pass: the small parity goldens through ab_libs/libnerfloam_hip_asan.so (scripts/asan_build.py: host and
# device code instrumented, gfx950:xnack+) on the GPU box.  `scripts/asan_run.sh TAG` -> gpurun_out/TAG_asan.log; zero "ERROR: AddressSanitizer"
# lines is the pass criterion.  Every step runs under its own short timeout: a hung instrumented kernel must not hold the box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-asan}
LIB=$PWD/ab_libs/libnerfloam_hip_asan.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
[ -f "$LIB" ] || { echo "no $LIB - run python scripts/asan_build.py first"; exit 1; }
export NL_LIB_PATH=$LIB HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0:abort_on_error=0 LD_PRELOAD=$RT
SEL=${2:-"tests/test_gpu_parity.py tests/test_gpu_reference_shapes.py"}
KEXPR=${3:-"iteration_matches_oracle or three_steps or intersect_cap or one_launch_sampler or one_call_iteration or tracking_matches or unit_directions or fused_intersect"}
timeout ${ASAN_TIMEOUT:-420} python -m pytest $SEL -k "$KEXPR" -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_asan.log 2>&1
echo "pytest under ASan rc=$?"
grep -c "ERROR: AddressSanitizer" $OUT/${TAG}_asan.log | sed 's/^/AddressSanitizer reports: /'
tail -15 $OUT/${TAG}_asan.log
