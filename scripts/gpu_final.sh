#!/usr/bin/env bash
# end-of-round validation: the whole -m gpu suite, smoke(), then the profile set
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r02_c}
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/${TAG}_pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/gpu_profiles.sh $TAG
