#!/usr/bin/env python
"""Summarises a rocprofv3 rocpd database (trace_results.db) into a small CSV:
one row per (kernel, grid size) with calls / total / avg / min / max duration (us).

    python scripts/rocpd_summary.py gpurun_out/<run>/prof/trace_results.db > profiles/<name>.csv
"""
import csv
import sqlite3
import sys


def main(path):
  c = sqlite3.connect(path)
  rows = c.execute(
      "select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, "
      "count(*), sum(duration), avg(duration), min(duration), max(duration) "
      "from kernels group by name, grid_x order by sum(duration) desc").fetchall()
  total = sum(r[8] for r in rows) or 1
  w = csv.writer(sys.stdout)
  w.writerow(["kernel", "grid_x", "workgroup_x", "lds_bytes", "vgpr", "agpr", "sgpr", "calls",
              "total_us", "avg_us", "min_us", "max_us", "pct"])
  for r in rows:
    name = r[0] if len(r[0]) < 120 else r[0][:117] + "..."
    w.writerow([name, *r[1:8], f"{r[8] / 1e3:.1f}", f"{r[9] / 1e3:.2f}", f"{r[10] / 1e3:.2f}",
                f"{r[11] / 1e3:.2f}", f"{100 * r[8] / total:.2f}"])


if __name__ == "__main__":
  main(sys.argv[1])
