#!/bin/bash
# Round-4 session 1: A/B of the persistent launches' tile -> workgroup map (GCAST_TILE_MAP=xcd) and of
# one workgroup per CU (GCAST_GRID_CAP=256), each as its own bench.py process of the headline workload.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s1; mkdir -p "$OUT"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check"
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 400 $B > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "rc=$?"; python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], {k: round(v["ms"], 3) for k, v in j["roofline"]["stages"].items()})
PY
}
run default GCAST_X=0
run xcd GCAST_TILE_MAP=xcd
run cap256 GCAST_GRID_CAP=256
run cap256_xcd GCAST_GRID_CAP=256 GCAST_TILE_MAP=xcd
run default2 GCAST_X=0
echo "== parity with the xcd map"
GCAST_TILE_MAP=xcd timeout 600 python -m pytest tests/test_step_gpu.py tests/test_rowmlp_gpu.py -m gpu -q -x 2>&1 | tail -3
