#!/bin/bash
# Round-4 session 10: VALU bursts of the layer-1 loops woven behind the previous chunk's MFMAs (GC_H_WEAVE, default) vs bursts
# (ab_libs/libgcast_noweave.so = -DGC_H_WEAVE=0): parity gate, A/B bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r04_s10}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q -x --timeout=200 2>&1 | tail -3 | tee "$OUT/pytest.log"
grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
bash scripts/session.sh bench-ab ${1:-r04_s10} "GCAST_LIB_PATH=ab_libs/libgcast_noweave.so" "GCAST_X=weave" "GCAST_LIB_PATH=ab_libs/libgcast_noweave.so" "GCAST_X=weave"
