#!/usr/bin/env bash
# Builds the reference's own Connect4 environment (env_cython.pyx, the only native
# code on the reference hot path) from the source WHERE IT LIES under /root/reference.
# Output: oracle/_ref/env_cython.so only (git-ignored).  The generated C++ is written to a
# temp dir and removed; no reference source is copied into the repo.
# Test infrastructure only: used by tests/golden/make_fixtures.py to produce golden vectors.
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -f "$REF/env_cython.pyx" ] || { echo "reference not present, skipping oracle/_ref build"; exit 0; }
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
cython -3 --cplus "$REF/env_cython.pyx" -o "$TMP/env_cython.cpp"
PYINC=$(python3 -c "import sysconfig;print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy;print(numpy.get_include())")
g++ -O2 -shared -fPIC -std=c++17 -w -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
    -I"$PYINC" -I"$NPINC" "$TMP/env_cython.cpp" -o "$OUT/env_cython.so"
echo "built $OUT/env_cython.so"
