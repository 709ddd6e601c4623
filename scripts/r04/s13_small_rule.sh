#!/bin/bash
# Round-4 session 13: launches of at most one tile per CU in the helper form (GCAST_HELPERS_SMALL, default on): parity on the
# small-graph suites (every launch there is small), then the emulated 8-way partition A/B and the 1 deg step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s13; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_partition_gpu.py tests/test_conditioned_gpu.py tests/test_deepgnn_gpu.py -m gpu -q -x --timeout=300 2>&1 | tail -3 | tee "$OUT/pytest.log"
grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
for v in 0 1; do
  echo "== partition emulated 8-way, GCAST_HELPERS_SMALL=$v"
  GCAST_HELPERS_SMALL=$v timeout 900 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_small$v.json" > "$OUT/partition8_small$v.log" 2>&1; echo "rc=$?"; tail -1 "$OUT/partition8_small$v.log" | cut -c1-900
done
for v in 0 1; do
  echo "== bench 1deg_13L_M5, GCAST_HELPERS_SMALL=$v"
  GCAST_HELPERS_SMALL=$v timeout 300 python bench.py --config 1deg_13L_M5 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], {k: round(v['ms'],3) for k,v in j['roofline']['stages'].items()})"
done
