#!/usr/bin/env bash
# the profile set of a round: bench line, rocprofv3 kernel stats of the same command, PMC passes (own runs), decoder phase stamps,
# timeline of the latency-bound steps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r04}
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady-state --no-parity --no-api-path --no-large-map > /tmp/prof_$TAG.log 2>&1 ; echo "rocprof rc=$?" )
cp $(find /tmp/prof_$TAG -name "*kernel_stats*.csv" | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null; cut -d, -f1-4 $OUT/${TAG}_kernel_stats.csv | head -14
bash scripts/gpu_pmc.sh ${TAG}pmc > $OUT/${TAG}_pmc.log 2>&1; tail -30 $OUT/${TAG}_pmc.log | cut -c1-260
timeout 300 python scripts/phase_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_phases_mode1.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o tl -- python ${GRAFT_REPO_ROOT:-/root/repo}/scripts/timeline_probe.py run > ${GRAFT_REPO_ROOT:-/root/repo}/$OUT/${TAG}_timeline_run.log 2>&1; echo "timeline rc=$?" )
python scripts/timeline_probe.py parse $(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline.txt 2>&1; tail -40 $OUT/${TAG}_timeline.txt | cut -c1-200
