#!/bin/bash
# Column-owner formulation micro-benchmark (scripts/ubench/colown_stream.hip) on the GPU box.
#   colown_session.sh <tag> [tiles]   tiles given: sustained run with random vs zero operands
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-co}
mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/colown_stream.hip -o /tmp/colown_stream 2> "$OUT/build.log" || { cat "$OUT/build.log"; exit 1; }
timeout 120 /tmp/colown_stream ${2:-} > "$OUT/colown_stream.txt" 2>&1
echo "rc=$?"; cat "$OUT/colown_stream.txt"
