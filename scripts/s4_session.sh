#!/bin/bash
# Round-3 session 4: bf16 tier suite + its bench; HBM-traffic PMC passes of the f16x3 bench (persistent kernels).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s4}
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest bf16 tier"
timeout 600 python -m pytest tests/test_bf16_tier_gpu.py -m gpu -q -s --timeout=300 > "$OUT/pytest_bf16.log" 2>&1
rcb=$?; echo "pytest bf16 rc=$rcb"; grep -E "bf16 |BF16_TIER|passed|failed|Error|fault" "$OUT/pytest_bf16.log" | tail -30 | cut -c1-300
echo "== bench bf16 tier"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --precision bf16 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; echo "bench bf16 rc=$?"
python -c "
import json
b=json.load(open('$OUT/bench_bf16.json'))
print(b['ms_per_step'], b['stages_ms'], b['roofline']['frac'], b['output_finite'])"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $C"
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_$C" -o pmc -- \
      python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/pmc_$C.json" 2> "$OLDPWD/$OUT/pmc_$C.err"); echo "pmc $C rc=$?"
  python scripts/pmc_summary.py "$OUT/pmc_$C" > "$OUT/pmc_$C.csv" 2>> "$OUT/errors.txt"
  head -14 "$OUT/pmc_$C.csv" | cut -c1-200
  find "$OUT/pmc_$C" -type f -size +8M -delete
done
