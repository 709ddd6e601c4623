#!/bin/bash
# Shader clock / package power under ~7 s of back-to-back steps, for the chunked (GCAST_COLOWN=0)
# and the column-owner (GCAST_COLOWN=1) split-f16 kernels.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-smi}
mkdir -p "$OUT"
for V in 0 1; do
  ( for i in $(seq 1 200); do sleep 0.25; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; done ) > "$OUT/smi_co$V.txt" 2>&1 &
  SMI_PID=$!
  GCAST_COLOWN=$V timeout 300 python bench.py --steps 100 --warmup 2 --no-cpu-baseline --no-cross-check --op-timing-iters 1 > "$OUT/bench_co$V.json" 2> "$OUT/bench_co$V.err"
  kill $SMI_PID 2>/dev/null
  echo "== COLOWN=$V"; python - <<P
import json,re
d=json.loads(open("$OUT/bench_co$V.json").read().strip().splitlines()[-1]); print("ms_per_step", round(d["ms_per_step"],2))
rows=[]
for l in open("$OUT/smi_co$V.txt"):
    m=re.search(r"\((\d+)Mhz\).*?\(W\): ([\d.]+)", l)
    if m: rows.append((int(m.group(1)), float(m.group(2))))
busy=[r for r in rows if r[1] > 600]
print("samples", len(rows), "busy", len(busy))
if busy:
    import statistics as st
    print("sclk MHz median", st.median(r[0] for r in busy), "min", min(r[0] for r in busy), "max", max(r[0] for r in busy), "| power W median", st.median(r[1] for r in busy), "max", max(r[1] for r in busy))
P
done
