#!/usr/bin/env bash
# round-3: multi-GPU plumbing on one GPU - virtual ranks through the C exchanges, the world-1 RCCL probe, bench.py's sharded branch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-b1}
timeout 900 python -m pytest tests/test_gpu_dist_rccl.py "tests/test_gpu_parity.py::test_two_virtual_ranks_match_the_single_rank_iteration" "tests/test_gpu_parity.py::test_eight_virtual_ranks_with_the_touched_rows_exchange" tests/test_gpu_parity.py::test_dist_counter_merge_kernel_matches_the_host_rig tests/test_gpu_parity.py::test_hipgraph_replay_matches_eager tests/test_gpu_parity.py::test_one_call_iteration_equals_the_stage_calls -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -60 $OUT/${TAG}_pytest_gpu.log
timeout 600 python bench.py --rccl-world1 --steps 10 --warmup 3 2>$OUT/${TAG}_bench_world1.err | tee $OUT/${TAG}_bench_world1.json | cut -c1-1500; tail -5 $OUT/${TAG}_bench_world1.err
timeout 60 python bench.py --gpus 2 2>&1 | tail -2
