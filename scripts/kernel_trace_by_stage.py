#!/usr/bin/env python
"""Per-STAGE kernel durations of the 0.25 deg step from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K ...`.  The persistent row-MLP kernels launch every stage with the same grid (512 workgroups),
so a launch is identified by its POSITION in a step (see scripts/pmc_by_stage.py): the trace's MLP_LN dispatches
end with whole steps of 38 launches -- embed_grid, enc_edge, enc_node_mesh, enc_node_grid, 16 x (proc_edge,
proc_node), dec_edge, dec_node.  All whole steps at the end of the trace are averaged.

    python scripts/kernel_trace_by_stage.py gpurun_out/<run>/prof [--kernel rowmlp16] > profiles/<name>.csv
"""
import csv
import glob
import os
import sys

STAGES = (["enc_embed_grid", "enc_edge", "enc_node_mesh", "enc_node_grid"]
          + ["proc_edge", "proc_node"] * 16 + ["dec_edge", "dec_node"])


def main():
  root = sys.argv[1]
  # (default: both forms of the half-N launch -- rowmlp16h_kernel = two four-wave workgroups per CU, rowmlp16d_kernel =
  #  one eight-wave workgroup with weight-staging waves; they interleave within a step)
  kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "rowmlp16"
  rows = []
  for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
      for r in csv.DictReader(fh):
        if kernel in r["Kernel_Name"] and "<0" not in r["Kernel_Name"] and "ILi0E" not in r["Kernel_Name"]:
          rows.append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                       r["VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Grid_Size_X"],
                       "rowmlp16d" if "rowmlp16d" in r["Kernel_Name"] else "rowmlp16w" if "rowmlp16w" in r["Kernel_Name"] else "rowmlp16h"))
  rows.sort()
  n = len(STAGES)
  steps = len(rows) // n
  if steps < 1:
    raise SystemExit(f"fewer than {n} MLP_LN dispatches of {kernel} in the trace ({len(rows)})")
  steps = min(steps, int(os.environ.get("TRACE_STEPS", "4")))
  tail = rows[-steps * n:]
  w = csv.writer(sys.stdout)
  w.writerow(["stage", "launches_per_step", "steps_averaged", "mean_us_per_launch", "min_us", "max_us", "total_us_per_step",
              "vgpr", "scratch_bytes", "lds_bytes", "grid_x", "kernel"])
  total = 0.0
  for name in dict.fromkeys(STAGES):
    idx = [k for k, s in enumerate(STAGES) if s == name]
    v = [tail[s * n + k][1] for s in range(steps) for k in idx]
    meta = tail[idx[0]]
    per_step = sum(v) / steps
    total += per_step
    w.writerow([name, len(idx), steps, f"{sum(v) / len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}", f"{per_step:.1f}",
                meta[2], meta[3], meta[4], meta[5], meta[6]])
  w.writerow(["_all_row_mlp_launches", n, steps, "", "", "", f"{total:.1f}", "", "", "", "", ""])


if __name__ == "__main__":
  main()
