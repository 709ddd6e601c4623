#!/usr/bin/env python
"""Groups a rocprofv3 --kernel-trace CSV (`*kernel_trace.csv`) by (kernel, grid size): calls, total /
mean / min / max duration, registers, LDS.  The per-kernel-NAME stats of `--stats` mix every launch
shape of the one templated kernel; the dominant launch's time is only visible per grid size.

    python scripts/kernel_trace_by_shape.py gpurun_out/<run>/prof > profiles/<name>.csv
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
  files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
  if not files:
    raise SystemExit(f"no *kernel_trace.csv under {root}")
  acc = defaultdict(list)
  meta = {}
  for f in files:
    with open(f, newline="") as fh:
      for row in csv.DictReader(fh):
        key = (row["Kernel_Name"], int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))
        acc[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"],
                     row["Scratch_Size"])
  total = sum(sum(v) for v in acc.values()) or 1.0
  w = csv.writer(sys.stdout)
  w.writerow(["kernel", "grid_x", "workgroup_x", "workgroups", "vgpr", "agpr", "sgpr", "lds_bytes", "scratch_bytes",
              "calls", "total_us", "mean_us", "min_us", "max_us", "pct_of_gpu_time"])
  for key, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    name, grid, wg = key
    name = name if len(name) < 100 else name[:97] + "..."
    w.writerow([name, grid, wg, grid // max(wg, 1), *meta[key], len(v), f"{sum(v):.1f}", f"{sum(v) / len(v):.2f}",
                f"{min(v):.2f}", f"{max(v):.2f}", f"{100 * sum(v) / total:.2f}"])


if __name__ == "__main__":
  main(sys.argv[1])
