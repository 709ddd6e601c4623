#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-bf}
mkdir -p "$OUT"
echo "== pytest (gpu): rowmlp, step, rollout"
timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_rollout_gpu.py -m gpu -q -s --timeout=600 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; grep -E "bf16|passed|failed|Error|rel-RMSE" "$OUT/pytest.log" | tail -25
echo "== bench bf16 tier"
timeout 900 python bench.py --steps 3 --warmup 1 --precision bf16 --no-cpu-baseline > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"
echo "bench rc=$?"; python -c "
import json
b=json.load(open('$OUT/bench_bf16.json'))
print(b['value'], b['ms_per_step'], b['stages_ms'], b.get('cross_check'), b['roofline']['achieved'])"
