#!/bin/bash
# Round-3 session 7: scalar segment-sum run logic -- tests + benches (f16x3, bf16 tier); then the 0.25 deg 3-step
# rollout oracle fixture (host cores of the GPU box, ~7 minutes) and the full-size rollout parity test against it.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s7}
mkdir -p "$OUT"
echo "== pytest (f16x3h rowmlp/step/plan + bf16 tier)"
timeout 600 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_bf16_tier_gpu.py -m gpu -x -q --timeout=300 -k "f16x3h or fewer or bf16_tier or node_like or edge_launch or external or whole_step" > "$OUT/pytest.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 "$OUT/pytest.log" | cut -c1-300
[ $rc -eq 0 ] || { grep -E "Error|error|assert|fault" "$OUT/pytest.log" | head -20 | cut -c1-300; exit 1; }
echo "== bench f16x3"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python -c "
import json
b=json.load(open('$OUT/bench.json'))
print(b['ms_per_step'], b['stages_ms'], b['roofline']['frac'])"
echo "== bench bf16 tier"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --precision bf16 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; echo "bench bf16 rc=$?"
python -c "
import json
b=json.load(open('$OUT/bench_bf16.json'))
print(b['ms_per_step'], b['stages_ms'], b['roofline']['frac'], b['output_finite'])"
if [ "${DO_FIXTURE:-1}" = "1" ]; then
  echo "== 0.25 deg rollout-3 oracle fixture"
  timeout 1500 python tests/golden/make_golden_rollout40.py --config 0p25deg --out-dir "$OUT" > "$OUT/make_fixture.log" 2>&1; echo "fixture rc=$?"; tail -2 "$OUT/make_fixture.log" | cut -c1-300
  if [ -f "$OUT/rollout3_0p25deg_rows.npz" ]; then
    cp "$OUT/rollout3_0p25deg_rows.npz" tests/golden/
    timeout 900 python -m pytest tests/test_rollout3_fullsize_gpu.py -m gpu -q -s --timeout=800 > "$OUT/pytest_rollout3.log" 2>&1; echo "rollout3 rc=$?"
    grep -E "ROLLOUT3|passed|failed|Error" "$OUT/pytest_rollout3.log" | cut -c1-600
  fi
fi
