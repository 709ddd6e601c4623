#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-ub}
mkdir -p "$OUT"
for f in scripts/ubench/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $f -o /tmp/$b 2> "$OUT/$b.build.log" && timeout 120 /tmp/$b > "$OUT/$b.txt" 2>&1
  echo "== $b rc=$?"; cat "$OUT/$b.txt"
done
echo "== pytest rowmlp + step (gpu)"
timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q --timeout=600 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -8 "$OUT/pytest.log"
