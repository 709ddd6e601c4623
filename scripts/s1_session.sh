#!/bin/bash
# Round-3 session 1: correctness of the persistent half-N kernels, single-launch probe, step bench,
# HBM-traffic PMC passes of the bench command (each --pmc pass on its own, kernel-trace only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-s1}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest rowmlp + step + plan (gpu)"
timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py -m gpu -x -q --timeout=600 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -5 "$OUT/pytest.log" | cut -c1-300
echo "== half_probe"
HALF_BIG_SCRATCH=1 HALF_BUILDS="r02:@ab_libs/libgcast_r02.so" timeout 600 python scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe.json" 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee "$OUT/probe.log"
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python -c "
import json
b=json.load(open('$OUT/bench.json'))
print(b['ms_per_step'], b['stages_ms'], b.get('cross_check'), b['roofline']['frac'])"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $C"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_$C" -o pmc -- \
      python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/pmc_$C.json" 2> "$OLDPWD/$OUT/pmc_$C.err"); echo "pmc $C rc=$?"
  python scripts/pmc_summary.py "$OUT/pmc_$C" > "$OUT/pmc_$C.csv" 2>> "$OUT/errors.txt"
  head -12 "$OUT/pmc_$C.csv" | cut -c1-200
  find "$OUT/pmc_$C" -type f -size +8M -delete
done
