#!/bin/bash
# Round-3 session 5: one-pass edge launches (GC_W2_NATURAL) -- kernel tests, whole-step tests, A/B bench; bf16 tier suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s5}
mkdir -p "$OUT"
echo "== pytest one-pass (edge block, step, plan; f16x3h)"
timeout 600 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py -m gpu -x -q --timeout=300 -k "(edge_block or step or plan or fewer) and (f16x3h or fewer)" > "$OUT/pytest_onepass.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 "$OUT/pytest_onepass.log" | cut -c1-300
[ $rc -eq 0 ] || { grep -E "Error|error|assert|fault" "$OUT/pytest_onepass.log" | head -20 | cut -c1-300; }
echo "== pytest bf16 tier"
timeout 600 python -m pytest tests/test_bf16_tier_gpu.py -m gpu -q -s --timeout=300 > "$OUT/pytest_bf16.log" 2>&1
echo "pytest bf16 rc=$?"; grep -E "^bf16 |BF16_TIER|passed|failed|Error|fault" "$OUT/pytest_bf16.log" | tail -30 | cut -c1-300
if [ $rc -eq 0 ]; then
  for v in 1 0; do
    echo "== bench f16x3 GCAST_ONEPASS=$v"
    GCAST_ONEPASS=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check > "$OUT/bench_onepass$v.json" 2> "$OUT/bench_onepass$v.err"; echo "bench rc=$?"
    python -c "
import json
b=json.load(open('$OUT/bench_onepass$v.json'))
print(b['ms_per_step'], b['stages_ms'], b['roofline']['frac'])"
  done
fi
