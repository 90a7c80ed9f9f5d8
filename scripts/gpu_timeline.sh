#!/usr/bin/env bash
# `scripts/gpu_timeline.sh TAG`: rocprofv3 kernel trace of scripts/timeline_probe.py -> gpurun_out/TAG_timeline.txt (+ the host-timed lines)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-tl}
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o tl -- python ${GRAFT_REPO_ROOT:-/root/repo}/scripts/timeline_probe.py run > ${GRAFT_REPO_ROOT:-/root/repo}/$OUT/${TAG}_timeline_run.log 2>&1; echo "timeline rc=$?" )
python scripts/timeline_probe.py parse $(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline.txt 2>&1
grep "host-timed\|failed" $OUT/${TAG}_timeline_run.log
sed -n '/section 5/,$p' $OUT/${TAG}_timeline.txt | cut -c1-150
