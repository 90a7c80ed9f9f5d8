#!/usr/bin/env bash
# round-6 end-of-round call: the whole -m gpu suite, smoke, the DEFAULT bench line (driver flags), rocprofv3 kernel stats of the bench, PMC passes
# (summarised on the box: the raw counter CSVs stay there - gpurun_out/ must stay under 64 MiB), decoder layout-2 stamps, timeline, world-1 RCCL
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r06_z}
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -12 $OUT/${TAG}_pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/${TAG}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; tail -2 $OUT/${TAG}_bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path --no-large-map --no-settings --no-pmc > /tmp/prof_$TAG.log 2>&1 ; echo "rocprof rc=$?" )
cp $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null; cut -d, -f1-4 $OUT/${TAG}_kernel_stats.csv | head -16
OUT=/tmp/pmcout_$TAG; mkdir -p $OUT
run() { name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path --no-large-map --no-settings --no-pmc > /tmp/pmc_${TAG}_$name.log 2>&1; echo "$name rc=$?" )
  cp /tmp/pmc_${TAG}_$name/p_counter_collection.csv $OUT/${TAG}pmc_${name}_counters.csv 2>/dev/null; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
run mfma16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python scripts/pmc_summary.py $OUT/${TAG}pmc gpurun_out/${TAG}_pmc_summary.json 2>&1 | tail -30 | cut -c1-260
OUT=gpurun_out
NL_PROBE_STAMP_LAYOUT=2 timeout 400 python scripts/decoder_layout_probe.py 131072 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_decoder_layouts.txt; tail -40 $OUT/${TAG}_decoder_layouts.txt | cut -c1-220
timeout 300 python scripts/large_map_legs.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_large_map_legs.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o tl -- python ${GRAFT_REPO_ROOT:-/root/repo}/scripts/timeline_probe.py run > ${GRAFT_REPO_ROOT:-/root/repo}/$OUT/${TAG}_timeline_run.log 2>&1; echo "timeline rc=$?" )
python scripts/timeline_probe.py parse $(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline.txt 2>&1; grep -v amdgpu.ids $OUT/${TAG}_timeline_run.log | tail -14 | cut -c1-220
timeout 600 python bench.py --rccl-world1 --no-cpu-baseline --no-api-path --no-large-map --no-settings --no-pmc > $OUT/${TAG}_bench_rccl_world1.json 2> $OUT/${TAG}_bench_rccl_world1.err; echo "rccl-world1 bench rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench_rccl_world1.json").read().strip().splitlines()[-1])
    print("rccl-world1: ms/step %.4f" % d["ms_per_step"], {k: v for k, v in d.get("sharded", {}).items() if k in ("launch_mode", "ms_per_step_without_exchanges", "exchange_ms_per_step", "embedding_exchange")})
except Exception as e:
    print("rccl-world1 parse failed", e)
PY
du -sh gpurun_out
