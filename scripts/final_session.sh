#!/bin/bash
# Round-end check on the GPU box: smoke, the whole GPU suite, the bench line (with cpu_baseline),
# rocprofv3 kernel trace grouped by (kernel, grid size), HBM-traffic and SQ counter passes of the
# same bench command (each --pmc pass on its own, kernel-trace only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
if [ "${DO_TESTS:-1}" = "1" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -rA > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed|FULLSIZE_PARITY|ROLLOUT40_PARITY" "$OUT/pytest_gpu.log" | tail -6 | cut -c1-700
fi
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-400 "$OUT/bench.json"
echo "== bench, two-deep ring build (A/B baseline)"; GCAST_LIB_VARIANT=ring2 timeout 600 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 --no-cpu-baseline > "$OUT/bench_ring2.json" 2>> "$OUT/bench.err"; echo "bench ring2 rc=$?"; cut -c1-260 "$OUT/bench_ring2.json"
echo "== rocprofv3 kernel trace"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- \
    python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rocprof rc=$?"
python scripts/kernel_trace_by_shape.py "$OUT/prof" > "$OUT/kernel_trace_by_shape.csv" 2>> "$OUT/errors.txt"; head -8 "$OUT/kernel_trace_by_shape.csv" | cut -c1-200
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do cp "$f" "$OUT/kernel_stats.csv"; done
find "$OUT/prof" -type f -size +8M -delete
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT"; do
  N=$(echo $C | cut -d' ' -f1)
  echo "== rocprofv3 --pmc $N ..."
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_$N" -o pmc -- \
      python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/pmc_$N.json" 2> "$OLDPWD/$OUT/pmc_$N.err"); echo "pmc $N rc=$?"
  python scripts/pmc_summary.py "$OUT/pmc_$N" > "$OUT/pmc_$N.csv" 2>> "$OUT/errors.txt"
  head -4 "$OUT/pmc_$N.csv" | cut -c1-180
  find "$OUT/pmc_$N" -type f -size +8M -delete
done
