#!/bin/bash
# Round-3 session 14: bf16 tier after the fast-reciprocal swish and the per-launch choice of rows per workgroup.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s14}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_bf16_tier_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/pytest_bf16.log"
grep -q "passed" "$OUT/pytest_bf16.log" && ! grep -q "failed\|error" "$OUT/pytest_bf16.log" || { echo "GATE: parity failed"; exit 1; }
for R in auto 64; do
  if [ $R = auto ]; then unset GCAST_BF16_ROWS; else export GCAST_BF16_ROWS=$R; fi
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --precision bf16 2>&1 | grep -v amdgpu.ids | tail -1 | tee "$OUT/bench_bf16_rows_$R.json"
done
