#!/bin/bash
# Round-3 session 16: DeviceRollout in the bf16 tier (parity test), then BASELINE.json config 3 -- the
# 40-step 0.25 deg rollout resident in HBM -- in both arithmetics.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s16}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_rollout_gpu.py -m gpu -x -q -s 2>&1 | grep -E "bf16 tier|passed|failed|Error" | cut -c1-400 | tee "$OUT/pytest_rollout.log"
grep -q "passed" "$OUT/pytest_rollout.log" && ! grep -q "failed" "$OUT/pytest_rollout.log" || { echo "GATE: test failed"; exit 1; }
timeout 900 python scripts/rollout_bench.py --steps 40 --out "$OUT/rollout40_f16x3.json" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
timeout 900 python scripts/rollout_bench.py --steps 40 --precision bf16 --out "$OUT/rollout40_bf16_tier.json" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
