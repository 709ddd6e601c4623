#!/bin/bash
# One parameterised GPU-box session runner (replaces the round-2/3 one-off scripts/s<N>_session.sh files; what each
# of those ran is in the git history and in the header of the profiles/r0*_s<N>_* files it produced).
#
#   scripts/session.sh bench-ab <out> [bench args --] "<ENV=.. ENV=..>" ...   one bench.py process per environment
#   scripts/session.sh probe    <out> <shapes> "<tag:@lib.so>" ...            scripts/half_probe.py per build variant
#   scripts/session.sh pytest   <out> "<ENV=..>" <pytest args ...>            a (gating) GPU test selection
#
# Everything lands under gpurun_out/<out>/; copy what is to be kept into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
MODE=$1; OUT=gpurun_out/$2; shift 2
mkdir -p "$OUT"
export TMPDIR=/tmp
summary() { python - "$1" <<'PY'
import json, sys
try:
  j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(j["ms_per_step"], "ms/step", {k: round(v["ms"], 3) for k, v in j["roofline"]["stages"].items()})
except Exception as e:
  print("no bench line:", e)
PY
}
case "$MODE" in
  bench-ab)
    ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0"
    if [[ " $* " == *" -- "* ]]; then ARGS=""; while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done; shift; fi
    i=0
    for ENVS in "$@"; do
      i=$((i + 1)); tag=$(echo "${ENVS:-default}" | tr ' =/' '__-' | cut -c1-60)
      echo "== [$i] $ENVS"
      env $ENVS timeout 600 python bench.py $ARGS > "$OUT/bench_${i}_$tag.json" 2> "$OUT/bench_${i}_$tag.err"; echo "rc=$?"
      summary "$OUT/bench_${i}_$tag.json"
    done;;
  probe)
    SHAPES=$1; shift
    for B in "$@"; do
      tag=${B%%:*}
      HALF_BUILDS="$B" PROBE_SHAPES=$SHAPES timeout 300 python -u scripts/half_probe.py --rounds 2 --iters 10 \
        --out "$OUT/probe_$tag.json" 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -4 | tee "$OUT/probe_$tag.log"
    done;;
  pytest)
    ENVS=$1; shift
    env $ENVS timeout 1500 python -m pytest "$@" -q --timeout=900 > "$OUT/pytest.log" 2>&1; rc=$?
    echo "pytest rc=$rc"; tail -5 "$OUT/pytest.log" | cut -c1-300; exit $rc;;
  *) echo "unknown mode $MODE"; exit 2;;
esac
