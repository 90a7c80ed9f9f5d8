#!/usr/bin/env bash
# Build the REFERENCE sparse-octree extension (svo) unmodified, from the sources
# where they lie under /root/reference, into oracle/_ref/svo_ref.so.
# Test infrastructure only: used to pin oracle/nl_oracle.c's octree restatement
# and to generate tests/golden/*.npz.  Nothing is copied into the repo.
# Recipe: plain g++ on the two source files (no setup.py, no cmake).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${NL_REFERENCE_ROOT:-/root/reference}"
SRC="$REF/third_party/sparse_octree"
if [ ! -d "$SRC" ]; then echo "reference not present at $REF - skipping oracle/_ref build"; exit 0; fi
OUT="$HERE/_ref"; mkdir -p "$OUT"
PY=${PYTHON:-python3}
TORCH_INC=$($PY - <<'PY'
import torch.utils.cpp_extension as e, sysconfig
print(" ".join("-I"+p for p in e.include_paths()+[sysconfig.get_paths()["include"]]))
PY
)
TORCH_LIB=$($PY -c "import torch.utils.cpp_extension as e; print(e.library_paths()[0])")
ABI=$($PY -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
g++ -O2 -std=c++17 -fPIC -shared -Wno-narrowing -w \
    -D_GLIBCXX_USE_CXX11_ABI=$ABI \
    -I"$HERE/shim" -I"$SRC/include" $TORCH_INC \
    "$SRC/src/octree.cpp" "$SRC/src/bindings.cpp" \
    -L"$TORCH_LIB" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TORCH_LIB" \
    -o "$OUT/svo_ref.so"
echo "built $OUT/svo_ref.so"
