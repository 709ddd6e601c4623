#!/bin/bash
# L2 / TA counter passes of the bench command (one --pmc pass each, kernel-trace only): how busy the
# texture-address path is under the half-N kernels and where their L2 requests are served.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pmc_cache}
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE TCC_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OLDPWD/$OUT/pmc_$N" -o pmc -- \
      python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OLDPWD/$OUT/pmc_$N.json" 2> "$OLDPWD/$OUT/pmc_$N.err"); echo "pmc $N rc=$?"
  python scripts/pmc_summary.py "$OUT/pmc_$N" > "$OUT/pmc_$N.csv" 2>> "$OUT/errors.txt"
  grep -E "rowmlp16h_kernel<1>.*,(1310720|164096|4153088|12656640)," "$OUT/pmc_$N.csv" | cut -c40-200
  tail -2 "$OUT/pmc_$N.err" | cut -c1-200
  find "$OUT/pmc_$N" -type f -size +8M -delete
done
