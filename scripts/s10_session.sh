#!/bin/bash
# Round-3 session 10: phase trace + single-launch timing of the GC_PREC_BF16 tier's launch shapes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s10}
mkdir -p "$OUT"
HALF_TRACE=1 PROBE_SHAPES=proc_edge_bf16,dec_edge_bf16,node_grid_bf16 timeout 600 python scripts/half_probe.py --out "$OUT/htrace_bf16.json" 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tee "$OUT/htrace_bf16.log"
PROBE_SHAPES=proc_edge_bf16,dec_edge_bf16,node_grid_bf16 timeout 600 python scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe_bf16.json" 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tee "$OUT/probe_bf16.log"
