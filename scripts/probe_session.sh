#!/bin/bash
# Kernel-level probe session: GPU parity tests of the row-MLP family, A/B timing of build
# variants + profiling-only experiment builds, SQ/LDS PMC counters of the dominant launch.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-probe}
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest rowmlp + step (gpu)" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q --timeout=600 -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -15 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
echo "== probe" | tee -a "$OUT/summary.txt"
timeout 900 python scripts/kernel_probe.py --out "$OUT/probe.json" > "$OUT/probe.log" 2>&1
echo "probe rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/probe.log" | tail -12 | tee -a "$OUT/summary.txt"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  echo "== pmc $SET" | tee -a "$OUT/summary.txt"
  (cd /tmp && PROBE_BUILDS=pipe2 PROBE_SHAPES=proc_edge,linear_grid timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OLDPWD/$OUT/pmc_$TAG" -o pmc -- \
      python "$OLDPWD/scripts/kernel_probe.py" --out "$OLDPWD/$OUT/probe_pmc.json" > "$OLDPWD/$OUT/pmc_$TAG.log" 2>&1)
  echo "rc=$?" | tee -a "$OUT/summary.txt"
  python scripts/pmc_summary.py "$OUT/pmc_$TAG" > "$OUT/pmc_$TAG.csv" 2>> "$OUT/summary.txt"
  grep rowmlp "$OUT/pmc_$TAG.csv" | cut -c1-160 | tee -a "$OUT/summary.txt"
  find "$OUT/pmc_$TAG" -type f -size +5M -delete
done
