#!/bin/bash
# Round-3 session 17: gc_advance_state with the per-channel tables held in registers -- rollout parity
# tests (gate), then the 40-step 0.25 deg rollout in both arithmetics.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s17}
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_rollout_gpu.py tests/test_rollout40_gpu.py tests/test_rollout3_fullsize_gpu.py -m gpu -x -q -s 2>&1 | grep -E "PARITY|device rollout|bf16 tier|passed|failed|Error" | cut -c1-500 | tee "$OUT/pytest_rollout.log"
grep -q "passed" "$OUT/pytest_rollout.log" && ! grep -q "failed" "$OUT/pytest_rollout.log" || { echo "GATE: test failed"; exit 1; }
timeout 900 python scripts/rollout_bench.py --steps 40 --out "$OUT/rollout40_f16x3.json" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600
timeout 900 python scripts/rollout_bench.py --steps 40 --precision bf16 --out "$OUT/rollout40_bf16_tier.json" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600
