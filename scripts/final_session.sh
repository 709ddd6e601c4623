#!/bin/bash
# Round-end verification on the GPU box: parity tests (all but the 0.25 deg full-size file), smoke,
# rocprofv3 kernel stats of the bench command, the bench line.  Logs under gpurun_out/<tag>/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-final}
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu (without tests/test_fullsize_gpu.py)" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests -m gpu -q --timeout=300 --ignore=tests/test_fullsize_gpu.py > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
echo "== smoke" | tee -a "$OUT/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
echo "== rocprofv3 kernel stats of bench.py" | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- \
    python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do head -8 "$f" | cut -c1-200 | tee -a "$OUT/summary.txt"; done
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 5 --warmup 1 --no-cross-check > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/summary.txt"; cut -c1-700 "$OUT/bench.json" | tee -a "$OUT/summary.txt"
