#!/bin/bash
# Round-3 session 6: phase trace (wave 0 of every tile) of the edge-update shapes, two-pass vs one-pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s6}
mkdir -p "$OUT"
HALF_TRACE=1 PROBE_SHAPES=proc_edge,dec_edge,dec_edge_onepass timeout 600 python scripts/half_probe.py --out "$OUT/htrace.json" 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tee "$OUT/htrace.log"
PROBE_SHAPES=proc_edge,dec_edge,dec_edge_onepass,gemm_only_mlp timeout 600 python scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe.json" 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tee "$OUT/probe.log"
