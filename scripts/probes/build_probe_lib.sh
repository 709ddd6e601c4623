#!/bin/bash
# Builds a PROFILING library from the shipped sources + one of the patches in this directory (the
# arithmetic-altering experiment switches live here, not in graphcast_amd/csrc):
#
#     scripts/probes/build_probe_lib.sh <patch> <out.so> [-DGC_H_DROP=1 ...]
#
# The result carries ";PROFILING_BUILD" in gc_build_info and is refused by graphcast_amd._native.load;
# scripts/half_probe.py loads it next to the product library (HALF_BUILDS="tag:@path.so").
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PATCH=$1; OUT=$2; shift 2
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
cp "$ROOT"/graphcast_amd/csrc/*.hip "$ROOT"/graphcast_amd/csrc/*.inc "$ROOT"/include/gcast.h "$TMP"/
(cd "$TMP" && patch -s -p0 < "$ROOT/scripts/probes/$(basename "$PATCH")")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-inline-asm -DGC_PROFILING_BUILD -DGC_PIPE=2 "$@" \
  -I "$TMP" -I "$ROOT/include" -shared -fPIC "$TMP/gcast.hip" -o "$OUT"
echo "built $OUT ($*)"
# (f16x3_gemm_phase_semaphore.patch also patches gcast.h -- one pointer appended to gc_rowmlp_desc, filled in by the
#  library itself: the copy of the header in $TMP is the one compiled against; run it with GCAST_GEMM_MUTEX=1)
