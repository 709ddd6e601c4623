#!/usr/bin/env python
"""Summarises a rocprofv3 --pmc run (CSV `*counter_collection.csv`) per (kernel, grid size,
counter): calls, mean, min, max of the counter value.

    python scripts/pmc_summary.py gpurun_out/<run>/pmc_fetch > profiles/<name>.csv
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
  files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
  if not files:
    raise SystemExit(f"no *counter_collection.csv under {root}")
  acc = defaultdict(list)
  for f in files:
    with open(f, newline="") as fh:
      for row in csv.DictReader(fh):
        name = row.get("Kernel_Name", "")
        grid = row.get("Grid_Size", row.get("Grid_Size_X", ""))
        acc[(name, grid, row["Counter_Name"])].append(float(row["Counter_Value"]))
  w = csv.writer(sys.stdout)
  w.writerow(["kernel", "grid_size", "counter", "calls", "mean", "min", "max"])
  for (name, grid, counter), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    name = name if len(name) < 100 else name[:97] + "..."
    w.writerow([name, grid, counter, len(v), f"{sum(v) / len(v):.1f}", f"{min(v):.1f}", f"{max(v):.1f}"])


if __name__ == "__main__":
  main(sys.argv[1])
