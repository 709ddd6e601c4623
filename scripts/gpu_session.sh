#!/bin/bash
# One GPU-box session: smoke, GPU parity tests (second build variant if the first fails),
# bench line, rocprofv3 kernel trace.  Everything is logged under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-session}
mkdir -p "$OUT"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > "$OUT/gpu.txt"

echo "== smoke" | tee "$OUT/summary.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"

echo "== pytest -m gpu (main)" | tee -a "$OUT/summary.txt"
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > "$OUT/pytest_gpu.log" 2>&1
RC=$?
echo "pytest rc=$RC" | tee -a "$OUT/summary.txt"; tail -40 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
if [ $RC -ne 0 ]; then
  echo "== pytest -m gpu (pipe1 variant)" | tee -a "$OUT/summary.txt"
  GCAST_LIB_VARIANT=pipe1 timeout 1200 python -m pytest tests -m gpu -q -rA --timeout=600 > "$OUT/pytest_gpu_pipe1.log" 2>&1
  echo "pytest(pipe1) rc=$?" | tee -a "$OUT/summary.txt"; tail -40 "$OUT/pytest_gpu_pipe1.log" | tee -a "$OUT/summary.txt"
  if [ "${FALLBACK_TO_PIPE1:-1}" = "1" ] && tail -1 "$OUT/pytest_gpu_pipe1.log" | grep -q passed && ! tail -1 "$OUT/pytest_gpu_pipe1.log" | grep -q failed; then
    export GCAST_LIB_VARIANT=pipe1
    echo "continuing with GCAST_LIB_VARIANT=pipe1" | tee -a "$OUT/summary.txt"
  fi
fi

echo "== bench" | tee -a "$OUT/summary.txt"
timeout 1500 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"

if [ "${DO_SMI:-0}" = "1" ]; then
  # power / clocks under ~15 s of back-to-back steps (is the chip power-limited under this kernel?)
  echo "== rocm-smi under load" | tee -a "$OUT/summary.txt"
  ( for i in $(seq 1 60); do sleep 1; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; done ) > "$OUT/smi_under_load.txt" 2>&1 &
  SMI_PID=$!
  timeout 600 python bench.py --steps 200 --warmup 1 --no-cpu-baseline --no-cross-check > "$OUT/bench_200.json" 2> "$OUT/bench_200.err"
  kill $SMI_PID 2>/dev/null
  sort "$OUT/smi_under_load.txt" | uniq -c | sort -rn | head -6 | cut -c1-200 | tee -a "$OUT/summary.txt"
fi

if [ "${DO_F32:-0}" = "1" ]; then
  echo "== bench (exact fp32 MFMA mode)" | tee -a "$OUT/summary.txt"
  timeout 900 python bench.py --steps 2 --warmup 1 --precision f32 --no-cpu-baseline --no-cross-check > "$OUT/bench_f32.json" 2> "$OUT/bench_f32.err"
  echo "bench f32 rc=$?" | tee -a "$OUT/summary.txt"; cut -c1-600 "$OUT/bench_f32.json" | tee -a "$OUT/summary.txt"
fi

if [ "${DO_PROF:-1}" = "1" ]; then
  echo "== rocprofv3 kernel trace" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- \
      python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
  echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
  find "$OUT/prof" -type f | head | tee -a "$OUT/summary.txt"
  for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do head -12 "$f" | cut -c1-220 | tee -a "$OUT/summary.txt"; done
  # keep only the small summaries (the raw trace can be large)
  find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
fi

if [ "${DO_PMC:-0}" = "1" ]; then
  # HBM traffic counters: one pass per counter (FETCH_SIZE takes 3 of the 4 TCC slots), kernel-trace only
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== rocprofv3 --pmc $C" | tee -a "$OUT/summary.txt"
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_$C" -o pmc -- \
        python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/pmc_$C.json" 2> "$OLDPWD/$OUT/pmc_$C.err")
    echo "pmc $C rc=$?" | tee -a "$OUT/summary.txt"
    python scripts/pmc_summary.py "$OUT/pmc_$C" > "$OUT/pmc_$C.summary.csv" 2>> "$OUT/summary.txt"
    head -8 "$OUT/pmc_$C.summary.csv" | cut -c1-200 | tee -a "$OUT/summary.txt"
    find "$OUT/pmc_$C" -type f -size +20M -delete
  done
fi
