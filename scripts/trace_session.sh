#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-tr}
mkdir -p "$OUT"
echo "== pytest rowmlp + step (gpu)"
timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q --timeout=600 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -8 "$OUT/pytest.log"
PROBE_BUILDS=${PROBE_BUILDS:-pipe2} timeout 600 python scripts/kernel_probe.py --out "$OUT/probe.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/probe.log"
PROBE_TRACE=1 timeout 600 python scripts/kernel_probe.py --out "$OUT/trace.json" 2>&1 | grep -v amdgpu.ids | tee "$OUT/trace.log"
if [ "${DO_BENCH:-0}" = "1" ]; then
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench rc=$?"; python -c "
import json,sys
b=json.load(open('$OUT/bench.json'))
print(b['value'], b['ms_per_step'], b['stages_ms'], b.get('cross_check'))"
fi
