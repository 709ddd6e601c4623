#!/bin/bash
# Round-end check on the GPU box: smoke, the whole GPU suite (gate), the bench line exactly as the
# driver runs it (with cpu_baseline, wall time noted), the bf16 tier's line, a rocprofv3 kernel trace
# of the same bench command summarised per stage, and the HBM-traffic PMC passes (each --pmc pass on
# its own, kernel-trace only) summarised per stage.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 "$OUT/smoke.log" | cut -c1-300
[ $rc = 0 ] || { echo "GATE: smoke failed"; exit 1; }
if [ "${DO_TESTS:-1}" = "1" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -rA > "$OUT/pytest_gpu.log" 2>&1; rc=$?; echo "pytest rc=$rc"
  grep -E "passed|failed|FULLSIZE_PARITY|ROLLOUT" "$OUT/pytest_gpu.log" | tail -6 | cut -c1-500
  [ $rc = 0 ] || { echo "GATE: GPU suite failed"; exit 1; }
fi
echo "== bench (driver's command)"; t0=$(date +%s); timeout 1200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s" | tee "$OUT/bench_wall.txt"; cut -c1-300 "$OUT/bench.json"
echo "== bench, bf16 tier"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 --precision bf16 > "$OUT/bench_bf16.json" 2>> "$OUT/bench.err"; echo "bench bf16 rc=$?"; cut -c1-200 "$OUT/bench_bf16.json"
echo "== rocprofv3 kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- \
    python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-cross-check --rollout-steps 0 --op-timing-iters 1 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rocprof rc=$?"
python scripts/kernel_trace_by_stage.py "$OUT/prof" > "$OUT/kernel_trace_by_stage.csv" 2>> "$OUT/errors.txt"; head -c 600 "$OUT/kernel_trace_by_stage.csv"; echo
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do cp "$f" "$OUT/kernel_stats.csv"; done
find "$OUT/prof" -type f -size +8M -delete
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $C"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_$C" -o pmc -- \
      python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --rollout-steps 0 --op-timing-iters 1 > "$OLDPWD/$OUT/pmc_$C.json" 2> "$OLDPWD/$OUT/pmc_$C.err"); echo "pmc $C rc=$?"
done
# (the row-MLP launches' counter rows, kept small enough to commit: the per-stage summaries are reproducible from them)
rows() { python - "$1" "$2" "${3:-rowmlp16}" <<'PY'
import csv, glob, os, sys
src, dst, pat = sys.argv[1], sys.argv[2], sys.argv[3]
out = None
for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
  with open(f, newline="") as fh:
    r = csv.DictReader(fh)
    for row in r:
      if pat in row["Kernel_Name"]:
        if out is None:
          out = csv.DictWriter(open(dst, "w", newline=""), fieldnames=r.fieldnames); out.writeheader()
        out.writerow(row)
PY
}
for C in FETCH_SIZE WRITE_SIZE; do rows "$OUT/pmc_$C" "$OUT/pmc_${C}_rowmlp_launches.csv"; done
python scripts/pmc_by_stage.py "$OUT/pmc_FETCH_SIZE_rowmlp_launches.csv" "$OUT/pmc_WRITE_SIZE_rowmlp_launches.csv" > "$OUT/pmc_by_stage.json" 2>> "$OUT/errors.txt"; head -c 400 "$OUT/pmc_by_stage.json"; echo
cp "$OUT/pmc_by_stage.json" "$OUT/current_pmc_by_stage.json"      # -> profiles/current_pmc_by_stage.json (bench.py: roofline.traffic)
if [ "${DO_SQ:-1}" = "1" ]; then
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT"; do
    i=$((i + 1))
    echo "== rocprofv3 --pmc (SQ set $i)"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_sq$i" -o pmc -- \
        python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --rollout-steps 0 --op-timing-iters 1 > "$OLDPWD/$OUT/pmc_sq$i.json" 2> "$OLDPWD/$OUT/pmc_sq$i.err"); echo "pmc sq$i rc=$?"
  done
  for i in 1 2; do rows "$OUT/pmc_sq$i" "$OUT/pmc_sq${i}_rowmlp_launches.csv"; done
  python scripts/sq_by_stage.py "$OUT/pmc_sq1_rowmlp_launches.csv" "$OUT/pmc_sq2_rowmlp_launches.csv" > "$OUT/sq_by_stage.json" 2>> "$OUT/errors.txt"
  cp "$OUT/sq_by_stage.json" "$OUT/current_sq_by_stage.json"      # -> profiles/current_sq_by_stage.json (bench.py: roofline.pmc)
  python -c "
import json; j=json.load(open('$OUT/sq_by_stage.json')); print({k: {m: round(v[m], 3) for m in ('mfma_busy_per_simd', 'wave_waiting', 'wave_waiting_on_lds', 'lds_bank_conflict') if m in v} for k, v in j.items() if not k.startswith('_')})"
fi
# round 6: the same counter passes for the Bfloat16Cast tier (bench.py --precision bf16 attaches them from
# profiles/current_{pmc,sq}_by_stage_bf16.json under the same source-hash rule)
if [ "${DO_BF16_PMC:-1}" = "1" ]; then
  BF="--precision bf16 --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --rollout-steps 0 --op-timing-iters 1"
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== rocprofv3 --pmc $C (bf16 tier)"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_bf16_$C" -o pmc -- \
        python "$OLDPWD/bench.py" $BF > "$OLDPWD/$OUT/pmc_bf16_$C.json" 2> "$OLDPWD/$OUT/pmc_bf16_$C.err"); echo "pmc bf16 $C rc=$?"
    rows "$OUT/pmc_bf16_$C" "$OUT/pmc_bf16_${C}_rowmlp_launches.csv" rowmlpbf
  done
  python scripts/pmc_by_stage.py "$OUT/pmc_bf16_FETCH_SIZE_rowmlp_launches.csv" "$OUT/pmc_bf16_WRITE_SIZE_rowmlp_launches.csv" --elem 2 > "$OUT/current_pmc_by_stage_bf16.json" 2>> "$OUT/errors.txt"; head -c 300 "$OUT/current_pmc_by_stage_bf16.json"; echo
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT"; do
    i=$((i + 1))
    echo "== rocprofv3 --pmc (SQ set $i, bf16 tier)"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_bf16_sq$i" -o pmc -- \
        python "$OLDPWD/bench.py" $BF > "$OLDPWD/$OUT/pmc_bf16_sq$i.json" 2> "$OLDPWD/$OUT/pmc_bf16_sq$i.err"); echo "pmc bf16 sq$i rc=$?"
    rows "$OUT/pmc_bf16_sq$i" "$OUT/pmc_bf16_sq${i}_rowmlp_launches.csv" rowmlpbf
  done
  python scripts/sq_by_stage.py "$OUT/pmc_bf16_sq1_rowmlp_launches.csv" "$OUT/pmc_bf16_sq2_rowmlp_launches.csv" --bf16 > "$OUT/current_sq_by_stage_bf16.json" 2>> "$OUT/errors.txt"
  python -c "
import json; j=json.load(open('$OUT/current_sq_by_stage_bf16.json')); print({k: {m: round(v[m], 3) for m in ('mfma_busy_per_simd', 'wave_waiting', 'wave_waiting_on_lds', 'lds_bank_conflict') if m in v} for k, v in j.items() if not k.startswith('_')})"
fi
find "$OUT" -type f -size +8M -delete
# round 5: the step at the part's power limit (socket power / shader clock polled while it runs), the partition-mode
# line at N = 1 (per-rank roofline, exchange probe) and the driver's SCALE launch form at N = 1
echo "== power probe"; timeout 300 python scripts/power_probe.py --seconds 6 --out "$OUT/power_probe.json" 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-400
echo "== bench --mode partition (N = 1)"; timeout 600 python bench.py --mode partition --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_partition_n1.json" 2>> "$OUT/bench.err"; echo "rc=$?"; cut -c1-200 "$OUT/bench_partition_n1.json"
echo "== bench under torch.distributed.run (N = 1)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_torchrun_n1.json" 2>> "$OUT/bench.err"; echo "rc=$? lines=$(wc -l < "$OUT/bench_torchrun_n1.json")"

# round 6: the second launch form of --gpus N -- ONE process driving N engines (the reference's pmap_devices form); on a
# 1-GPU box the two engines share the device
echo "== bench --gpus 2 --single-process"; timeout 600 python bench.py --gpus 2 --single-process --steps 5 --warmup 2 > "$OUT/bench_single_process_2.json" 2>> "$OUT/bench.err"; echo "rc=$?"; cut -c1-200 "$OUT/bench_single_process_2.json"
