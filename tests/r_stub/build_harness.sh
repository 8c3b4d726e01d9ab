#!/bin/sh
# Builds tests/harness/librshim_harness.so: the REAL shim (r/src/chicdiff_hip_shim.c) + the functional R-API stand-in (rstub.c)
# + the allocation counter, linked against the REAL libchicdiff_hip.so.  Test infrastructure (tests/r_stub/README.md).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$ROOT/tests/harness"
mkdir -p "$OUT"
# warnings are errors only where somebody asked for that (the test run does: CHICDIFF_HARNESS_WERROR=1); a new compiler's new
# warning must not break __graft_entry__.build()
WERROR=""
[ "${CHICDIFF_HARNESS_WERROR:-0}" = "1" ] && WERROR="-Werror"
CFLAGS="-std=gnu99 -O1 -g -fPIC -Wall -Wextra $WERROR -Wno-cast-function-type -I$ROOT/tests/r_stub -I$ROOT/include"
gcc $CFLAGS -Dchicdiff_hip_malloc=rstub_counted_malloc -Dchicdiff_hip_free=rstub_counted_free -c "$ROOT/r/src/chicdiff_hip_shim.c" -o "$OUT/rshim_shim.o"
gcc $CFLAGS -c "$ROOT/tests/r_stub/rstub.c" -o "$OUT/rshim_rstub.o"
gcc $CFLAGS -c "$ROOT/tests/r_stub/rstub_count.c" -o "$OUT/rshim_count.o"
gcc -shared -o "$OUT/librshim_harness.so" "$OUT/rshim_shim.o" "$OUT/rshim_rstub.o" "$OUT/rshim_count.o" \
    -L"$ROOT/chicdiff_amd/lib" -lchicdiff_hip -Wl,-rpath,"$ROOT/chicdiff_amd/lib" -Wl,-rpath,'$ORIGIN/../../chicdiff_amd/lib' -lm
echo "$OUT/librshim_harness.so"
