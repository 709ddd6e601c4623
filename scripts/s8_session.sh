#!/bin/bash
# Round-3 session 8: DeepGNN tests first (new code), then the whole GPU suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s8}
mkdir -p "$OUT"
echo "== pytest DeepGNN"
timeout 600 python -m pytest tests/test_deepgnn_gpu.py -m gpu -x -q -s --timeout=300 > "$OUT/pytest_deepgnn.log" 2>&1
rc=$?; echo "pytest deepgnn rc=$rc"; grep -E "DeepGNN|passed|failed|Error|error" "$OUT/pytest_deepgnn.log" | tail -20 | cut -c1-300
if [ "${FULL:-1}" = "1" ]; then
  echo "== pytest -m gpu (everything)"
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -rA > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed|FULLSIZE_PARITY|ROLLOUT40_PARITY|ROLLOUT3|BF16_TIER" "$OUT/pytest_gpu.log" | tail -12 | cut -c1-500
  grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | head -20 | cut -c1-300
fi
