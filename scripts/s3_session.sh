#!/bin/bash
# Round-3 session 3: gated -- smoke of the three half-N modes, then the row-MLP suites (f16x3h + bf16 tier),
# then short benches.  Stops at the first failing stage (GPU minutes are scarce).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s3}
mkdir -p "$OUT"
for mode in linear mlp_out mlp_ln; do
  timeout 60 python scripts/debug_half.py graphcast_amd/csrc/libgcast_hip.so $mode 1000 2>&1 | grep -E "OK|fault|rc" | tee -a "$OUT/debug.log"
done
grep -c "OK" "$OUT/debug.log" | grep -q 3 || { echo "STOP: smoke failed"; exit 1; }
echo "== pytest rowmlp (half) + plan"
timeout 600 python -m pytest tests/test_rowmlp_gpu.py tests/test_plan_gpu.py -m gpu -x -q --timeout=300 -k "f16x3h or fewer" > "$OUT/pytest_half.log" 2>&1
rc=$?; echo "pytest half rc=$rc"; tail -4 "$OUT/pytest_half.log" | cut -c1-300
[ $rc -eq 0 ] || { grep -E "Error|error|assert|fault" "$OUT/pytest_half.log" | head -20 | cut -c1-300; echo "STOP: half tests failed"; exit 1; }
echo "== pytest bf16 tier"
timeout 600 python -m pytest tests/test_bf16_tier_gpu.py -m gpu -x -q -s --timeout=300 > "$OUT/pytest_bf16.log" 2>&1
rcb=$?; echo "pytest bf16 rc=$rcb"; grep -E "bf16 |BF16_TIER|passed|failed|Error|fault" "$OUT/pytest_bf16.log" | tail -30 | cut -c1-300
echo "== bench f16x3"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python -c "
import json
b=json.load(open('$OUT/bench.json'))
print(b['ms_per_step'], b['stages_ms'], b['roofline']['frac'])"
if [ $rcb -eq 0 ]; then
  echo "== bench bf16 tier"
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --precision bf16 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; echo "bench bf16 rc=$?"
  python -c "
import json
b=json.load(open('$OUT/bench_bf16.json'))
print(b['ms_per_step'], b['stages_ms'], b['roofline']['frac'], b['output_finite'])"
fi
