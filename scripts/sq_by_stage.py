#!/usr/bin/env python
"""Per-STAGE SQ counters of one 0.25 deg step from rocprofv3 --pmc passes of
`bench.py --steps 1 --warmup 0 --op-timing-iters 1` (any number of pass directories; a launch is identified
by its position in the step exactly as in scripts/pmc_by_stage.py).  Per stage and counter: mean per launch;
derived where the inputs are present (units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES and the
SQ_WAIT_* / SQ_ACTIVE_INST_* counters are QUAD-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES is cycles
summed over SIMDs = 16 x #MFMA for the 16x16x32 shapes):

  mfma_busy_per_simd   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES / 2)     two resident waves per SIMD (both
                         forms of the launch: two four-wave workgroups, or one eight-wave workgroup, per CU)
  wave_waiting         = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  wave_waiting_on_lds  = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  lds_bank_conflict    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE

    python scripts/sq_by_stage.py <pass dir | counter csv> [...] > profiles/r03_..._sq_by_stage.json
    (the committed summary is reproducible from the committed rows:
     python scripts/sq_by_stage.py profiles/r03_final2_pmc_sq1_rowmlp_launches.csv profiles/r03_final2_pmc_sq2_rowmlp_launches.csv)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamp                                     # noqa: E402

STAGES = (["enc_embed_grid", "enc_edge", "enc_node_mesh", "enc_node_grid"]
          + ["proc_edge", "proc_node"] * 16 + ["dec_edge", "dec_node"])


def main():
  per = defaultdict(dict)           # counter -> {dispatch id: value}
  # (--bf16: the GC_PREC_BF16 tier's kernel instead of the f16x3 forms)
  kernels = ("rowmlpbf_kernel",) if "--bf16" in sys.argv else ("rowmlp16h_kernel", "rowmlp16d_kernel", "rowmlp16w_kernel")
  for root in [a for a in sys.argv[1:] if not a.startswith("--")]:
    files = [root] if os.path.isfile(root) else glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
      with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
          k = r["Kernel_Name"]
          if any(name in k for name in kernels) and "<0" not in k:
            d = per[r["Counter_Name"]]
            d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
  out = {}
  for counter, by_id in per.items():
    vals = [v for _, v in sorted(by_id.items())]
    if len(vals) < len(STAGES):
      continue
    last = vals[-len(STAGES):]
    acc = defaultdict(list)
    for stage, v in zip(STAGES, last):
      acc[stage].append(v)
    for stage, v in acc.items():
      out.setdefault(stage, {})[counter] = sum(v) / len(v)
  for stage, c in out.items():
    g = c.get
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_WAVE_CYCLES"):
      c["mfma_busy_per_simd"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (2.0 * g("SQ_WAVE_CYCLES"))
    if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
      c["wave_waiting"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAIT_INST_LDS") is not None and g("SQ_WAVE_CYCLES"):
      c["wave_waiting_on_lds"] = g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
      c["lds_bank_conflict"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
  if "--no-stamp" not in sys.argv:
    out["_stamp"] = _stamp.stamp()
  json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
  main()
