#!/bin/bash
# Round-3 session 9: partitioned step with split edge updates (exchange under the sender-local launches).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s9}
mkdir -p "$OUT"
for v in 1 0; do
  echo "== pytest partition (GCAST_OVERLAP=$v)"
  GCAST_OVERLAP=$v timeout 900 python -m pytest tests/test_partition_gpu.py -m gpu -x -q -s --timeout=600 > "$OUT/pytest_overlap$v.log" 2>&1
  echo "rc=$?"; grep -E "parts:|passed|failed|Error|error" "$OUT/pytest_overlap$v.log" | tail -8 | cut -c1-300
done
for v in 1 0; do
  echo "== partition emulated bench 8-way (GCAST_OVERLAP=$v)"
  GCAST_OVERLAP=$v timeout 900 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_overlap$v.json" > "$OUT/partition8_overlap$v.log" 2>&1
  echo "rc=$?"; tail -1 "$OUT/partition8_overlap$v.log" | cut -c1-1400
done
