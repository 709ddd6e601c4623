#!/usr/bin/env python
"""Shader clock per STAGE of one 0.25 deg step, from ONE rocprofv3 pass `--kernel-trace --pmc SQ_WAVE_CYCLES` of
`bench.py --steps 1 --warmup 0 --op-timing-iters 1`: the persistent row-MLP launches keep a fixed number of waves
resident for their whole duration (512 four-wave or 256 eight-wave workgroups = 2048 waves on the 1,024 SIMDs), so

    clock = 4 * SQ_WAVE_CYCLES / (waves * duration)          (SQ_WAVE_CYCLES counts quad-cycles summed over waves)

Launches are identified by position as in scripts/pmc_by_stage.py.  Evidence for DESIGN.md section 9.14: what a faster
processor edge update does to the clock of the node update behind it.

    python scripts/clock_by_stage.py <pass dir> > clock_by_stage.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STAGES = (["enc_embed_grid", "enc_edge", "enc_node_mesh", "enc_node_grid"]
          + ["proc_edge", "proc_node"] * 16 + ["dec_edge", "dec_node"])


def main():
  root = sys.argv[1]
  cyc, dur, grid = {}, {}, {}
  for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
      for r in csv.DictReader(fh):
        k = r["Kernel_Name"]
        if ("rowmlp16h_kernel" in k or "rowmlp16d_kernel" in k or "rowmlp16w_kernel" in k) and "<0" not in k and r["Counter_Name"] == "SQ_WAVE_CYCLES":
          i = int(r["Dispatch_Id"])
          cyc[i] = cyc.get(i, 0.0) + float(r["Counter_Value"])
          grid[i] = int(r["Grid_Size"]) // 64          # waves of the launch
  for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
      for r in csv.DictReader(fh):
        i = int(r["Dispatch_Id"])
        if i in cyc:
          dur[i] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
  ids = sorted(i for i in cyc if i in dur)
  if len(ids) < len(STAGES):
    raise SystemExit(f"only {len(ids)} row-MLP launches with both a counter and a duration")
  acc = defaultdict(list)
  for stage, i in zip(STAGES, ids[-len(STAGES):]):
    acc[stage].append((4.0 * cyc[i] / (grid[i] * dur[i]) / 1e9, dur[i] * 1e3))
  out = {s: {"clock_ghz": round(sum(c for c, _ in v) / len(v), 3), "ms_per_launch_under_the_counter_pass": round(sum(d for _, d in v) / len(v), 4),
             "launches": len(v)} for s, v in acc.items()}
  json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
  main()
