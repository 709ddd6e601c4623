#!/bin/bash
# Round-4 session 5: the per-CU GEMM-phase semaphore (GCAST_GEMM_MUTEX=1): parity gate, A/B bench, phase trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s5; mkdir -p "$OUT"
GCAST_GEMM_MUTEX=1 timeout 400 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q -x --timeout=150 2>&1 | tail -4 | tee "$OUT/pytest.log"
grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
bash scripts/session.sh bench-ab r04_s5 "GCAST_GEMM_MUTEX=0" "GCAST_GEMM_MUTEX=1" "GCAST_GEMM_MUTEX=0" "GCAST_GEMM_MUTEX=1"
echo "== trace with the semaphore"
GCAST_GEMM_MUTEX=1 HALF_TRACE=1 PROBE_SHAPES=proc_edge,gemm_only_mlp,node_grid timeout 300 python -u scripts/half_probe.py --out "$OUT/trace_mutex.json" 2>&1 | grep htrace | cut -c1-900
