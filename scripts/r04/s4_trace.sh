#!/bin/bash
# Round-4 session 4: phase traces (wave 0 = a multiplying wave) of the same launches in pair / lone / helper-wave form.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s4; mkdir -p "$OUT"
run() { tag=$1; shift; echo "== $tag"; env "$@" HALF_TRACE=1 PROBE_SHAPES=proc_edge,gemm_only_mlp,node_grid timeout 300 python -u scripts/half_probe.py --out "$OUT/trace_$tag.json" 2>&1 | grep htrace | cut -c1-900; }
run pair GCAST_HELPERS=0
run lone GCAST_HELPERS=0 GCAST_GRID_CAP=256
run helpers GCAST_HELPERS=1
