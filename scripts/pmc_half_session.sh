#!/bin/bash
# PMC counters of the half-N kernel (and the chunked one beside it) on single launches of the
# 0.25 deg shapes: two rocprofv3 passes (SQ slots: 8 per pass), kernel-trace only (gpurun refuses
# --pmc together with the other trace domains).  Summaries -> gpurun_out/<tag>/pmc_*.csv
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pmc_half}
mkdir -p "$OUT"
export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
      "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU")
TAGS=(sq lds_mfma)
for k in 0 1; do
  (cd /tmp && PROBE_SHAPES=${PROBE_SHAPES:-proc_edge,gemm_only_mlp} timeout 600 rocprofv3 --kernel-trace --pmc ${SETS[$k]} --output-format csv \
      -d "$OLDPWD/$OUT/pmc_${TAGS[$k]}" -o pmc -- python "$OLDPWD/scripts/half_probe.py" --rounds 1 --iters 3 \
      --out "$OLDPWD/$OUT/probe_under_pmc_${TAGS[$k]}.json" > "$OLDPWD/$OUT/pmc_${TAGS[$k]}.log" 2>&1)
  echo "pmc ${TAGS[$k]} rc=$?"
  python scripts/pmc_summary.py "$OUT/pmc_${TAGS[$k]}" > "$OUT/pmc_${TAGS[$k]}.csv" 2>> "$OUT/errors.txt"
  find "$OUT/pmc_${TAGS[$k]}" -type f -size +5M -delete
  grep -E "rowmlp16h_kernel<1>|rowmlp16_kernel<1>" "$OUT/pmc_${TAGS[$k]}.csv" | cut -c1-160
done
