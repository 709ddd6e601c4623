#!/bin/bash
# Round-4 session 12: upper bound of routing the layer-1 row fragments through LDS-DMA: a profiling library that never
# re-loads them (results wrong), pair and helper form, single-launch timings.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s12; mkdir -p "$OUT"
for H in 0 1; do
  echo "== GCAST_HELPERS=$H"
  GCAST_HELPERS=$H HALF_BUILDS="norows:@ab_libs/libgcast_norows.so" PROBE_SHAPES=proc_edge,gemm_only_mlp,node_grid timeout 300 python -u scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe_helpers$H.json" 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -4
done
