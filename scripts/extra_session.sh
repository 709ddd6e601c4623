#!/bin/bash
# Partitioned-step tests (config 5, emulated ranks) + HBM-resident 40-step rollout (config 3).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-ex}
mkdir -p "$OUT"
echo "== pytest partition + rollout (gpu)"
timeout 900 python -m pytest tests/test_partition_gpu.py tests/test_rollout_gpu.py -m gpu -q -s --timeout=600 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; grep -E "parts:|rollout|passed|failed|Error" "$OUT/pytest.log" | tail -15
echo "== rollout bench (config 3)"
timeout 900 python scripts/rollout_bench.py --steps ${ROLLOUT_STEPS:-40} --out "$OUT/rollout.json" > "$OUT/rollout.log" 2>&1
echo "rollout rc=$?"; tail -3 "$OUT/rollout.log" | cut -c1-800
echo "== partition emulated bench (config 5)"
timeout 900 python scripts/partition_emulated_bench.py --parts ${PARTS:-8} --out "$OUT/partition.json" > "$OUT/partition.log" 2>&1
echo "partition rc=$?"; tail -2 "$OUT/partition.log" | cut -c1-1500
