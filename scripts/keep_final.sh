#!/bin/bash
# Copies what a round-end session (scripts/final_session.sh <tag>) produced under gpurun_out/<tag>/ into profiles/<tag>_*
# (the files the docs and tests cite) and its stamped counter summaries into profiles/current_* (what bench.py attaches
# as roofline.traffic / roofline.pmc when the loaded library was built from the same sources).
set -eu
cd "$(dirname "$0")/.."
TAG=${1:?tag}
SRC=gpurun_out/$TAG
for f in bench.json bench_bf16.json bench_partition_n1.json bench_single_process_2.json bench_torchrun_n1.json bench_wall.txt \
         kernel_stats.csv kernel_trace_by_stage.csv pmc_by_stage.json sq_by_stage.json power_probe.json smoke.log pytest_gpu.log \
         pmc_FETCH_SIZE_rowmlp_launches.csv pmc_WRITE_SIZE_rowmlp_launches.csv pmc_sq1_rowmlp_launches.csv pmc_sq2_rowmlp_launches.csv \
         pmc_bf16_FETCH_SIZE_rowmlp_launches.csv pmc_bf16_WRITE_SIZE_rowmlp_launches.csv pmc_bf16_sq1_rowmlp_launches.csv pmc_bf16_sq2_rowmlp_launches.csv; do
  [ -f "$SRC/$f" ] && cp "$SRC/$f" "profiles/${TAG}_$f" || echo "missing: $SRC/$f"
done
for f in current_pmc_by_stage.json current_sq_by_stage.json current_pmc_by_stage_bf16.json current_sq_by_stage_bf16.json; do
  [ -f "$SRC/$f" ] && cp "$SRC/$f" "profiles/$f" || echo "missing: $SRC/$f"
done
for f in fullsize_parity.json partition8_fullsize_parity.json rollout40_fullsize_parity.json; do
  [ -f "gpurun_out/$f" ] && cp "gpurun_out/$f" "profiles/${TAG}_$f" || true
done
python - <<PY
import json
for f in ("current_pmc_by_stage.json", "current_sq_by_stage.json", "current_pmc_by_stage_bf16.json", "current_sq_by_stage_bf16.json"):
  print(f, json.load(open("profiles/" + f)).get("_stamp"))
PY
