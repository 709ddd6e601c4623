#!/bin/bash
# Round-4 session 3: the eight-wave "helper waves" form of the half-N launches (GCAST_HELPERS=1): parity gate, then A/B bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s3; mkdir -p "$OUT"
echo "== parity, GCAST_HELPERS=1 (per-launch tests first: a hang costs 120 s, not the session)"
GCAST_HELPERS=1 timeout 240 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=100 2>&1 | tail -4 | tee "$OUT/pytest_rowmlp.log"
grep -q " passed" "$OUT/pytest_rowmlp.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest_rowmlp.log" || { echo "GATE: per-launch parity failed"; exit 1; }
GCAST_HELPERS=1 timeout 400 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_deepgnn_gpu.py -m gpu -q -x --timeout=200 2>&1 | tail -4 | tee "$OUT/pytest_step.log"
grep -q " passed" "$OUT/pytest_step.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest_step.log" || { echo "GATE: step parity failed"; exit 1; }
bash scripts/session.sh bench-ab r04_s3 "GCAST_HELPERS=0" "GCAST_HELPERS=1" "GCAST_HELPERS=1 GCAST_TILE_MAP=xcd" "GCAST_HELPERS=0"
