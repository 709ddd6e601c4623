#!/bin/bash
# Round-3 session 12: the GC_PREC_BF16 launch with 128-row workgroups (eight waves, one weight stream)
# -- parity first (gate), then single-launch timing against the 64-row form.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s12}
mkdir -p "$OUT"
GCAST_BF16_ROWS=128 timeout 600 python -m pytest tests/test_bf16_tier_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest_bf16_rows128.log"
grep -q "passed" "$OUT/pytest_bf16_rows128.log" && ! grep -q "failed\|error" "$OUT/pytest_bf16_rows128.log" || { echo "GATE: parity failed"; exit 1; }
for R in 64 128; do
  GCAST_BF16_ROWS=$R PROBE_SHAPES=proc_edge_bf16,dec_edge_bf16,node_grid_bf16 timeout 400 python scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe_bf16_rows$R.json" 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tee "$OUT/probe_bf16_rows$R.log"
done
for R in 64 128; do
  GCAST_BF16_ROWS=$R timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --precision bf16 2>&1 | grep -v amdgpu.ids | tail -1 | tee "$OUT/bench_bf16_rows$R.json"
done
