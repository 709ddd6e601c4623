#!/bin/bash
# Round-4 session 7: the helper-wave form (now with the parked accumulators in LDS) on the big node-side launches only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s7; mkdir -p "$OUT"
GCAST_HELPERS=1 timeout 300 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=100 -k "chain or persistent or mlp_ln or edge_block" 2>&1 | tail -3 | tee "$OUT/pytest.log"
grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
bash scripts/session.sh bench-ab r04_s7 "GCAST_HELPERS_MIN_ROWS=0" "GCAST_HELPERS_MIN_ROWS=65536" "GCAST_HELPERS=1" "GCAST_HELPERS_MIN_ROWS=0" "GCAST_HELPERS_MIN_ROWS=65536"
