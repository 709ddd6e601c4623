#!/usr/bin/env python
"""Per-STAGE HBM traffic of one 0.25 deg step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 1 --warmup 0 --op-timing-iters 1`.  The persistent kernels launch every stage with the same
grid (512 workgroups), so a launch is identified by its POSITION: the row-MLP dispatches of a pass end with
whole steps of 38 launches -- embed_grid, enc_edge, enc_node_mesh, enc_node_grid, 16 x (proc_edge, proc_node),
dec_edge, dec_node -- and the last step (bench.py's per-op timing pass) is the one summarised.

FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for gfx950.  Algorithmic bytes per stage are computed from the graph sizes.

    python scripts/pmc_by_stage.py <dir of FETCH pass> <dir of WRITE pass> [--elem 4|2] > profiles/r03_pmc_by_stage.json
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamp                                     # noqa: E402

STAGES = (["enc_embed_grid", "enc_edge", "enc_node_mesh", "enc_node_grid"]
          + ["proc_edge", "proc_node"] * 16 + ["dec_edge", "dec_node"])


def rowmlp_values(root, kernel_sub):
  rows = []
  files = [root] if os.path.isfile(root) else glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
  for f in files:
    with open(f, newline="") as fh:
      for r in csv.DictReader(fh):
        # (rowmlp16h_kernel / rowmlp16d_kernel / rowmlp16w_kernel: the four-wave, helper-wave and wide form of the same launch)
        if any(k in r["Kernel_Name"] for k in kernel_sub) and "<0" not in r["Kernel_Name"]:
          rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
  rows.sort()
  return [v for _, v in rows]


def algorithmic(elem):
  """HBM bytes a stage has to move at 0.25 deg / M6 (rows of 512 values of `elem` bytes; x fp32)."""
  ng, nm = 1038240, 40962
  r_g2m, r_mesh, r_m2g = 1654528, 327680, 3164160          # packed edge rows (tests/test_fullsize_gpu.py sizes)
  row = 512 * elem
  return {
      "enc_embed_grid": ng * (471 + 32) * 4 + 2 * ng * row,              # x + tail in; h_grid, h.W_s out
      "enc_edge": r_g2m * row + ng * row + nm * row,                     # folded edge term, every grid row once, aggregate
      "enc_node_mesh": 5 * nm * row,
      "enc_node_grid": ng * row + 2 * ng * row,                          # h_grid in; h_grid2, h'.W_r out
      "proc_edge": 2 * r_mesh * row + 2 * nm * row + nm * row,           # e in/out; the two node tables; aggregate
      "proc_node": 5 * nm * row,                                         # h, agg in; h, h.W_s, h.W_r out
      "dec_edge": r_m2g * row + ng * row + nm * row + ng * row,          # folded edge term, h'.W_r rows, mesh table, aggregate
      "dec_node": 2 * ng * row + ng * 227 * 4,                           # h_grid2, aggregate in; y out
  }


def main():
  fetch_dir, write_dir = sys.argv[1], sys.argv[2]
  elem = int(sys.argv[sys.argv.index("--elem") + 1]) if "--elem" in sys.argv else 4
  kernel = ("rowmlpbf_kernel",) if elem == 2 else ("rowmlp16h_kernel", "rowmlp16d_kernel", "rowmlp16w_kernel")
  f, w = rowmlp_values(fetch_dir, kernel), rowmlp_values(write_dir, kernel)
  n = len(STAGES)
  if len(f) < n or len(w) < n:
    raise SystemExit(f"expected at least {n} row-MLP dispatches per pass, found {len(f)} / {len(w)}")
  f, w = f[-n:], w[-n:]
  alg = algorithmic(elem)
  out = {}
  for name in dict.fromkeys(STAGES):
    idx = [k for k, s in enumerate(STAGES) if s == name]
    fetch = sum(2.0 * 1024.0 * f[k] for k in idx) / len(idx)
    write = sum(1024.0 * w[k] for k in idx) / len(idx)
    out[name] = {"launches_per_step": len(idx), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                 "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": alg[name],
                 "traffic_over_algorithmic": (fetch + write) / alg[name]}
  tot_t = sum(v["traffic_bytes_per_launch"] * v["launches_per_step"] for v in out.values())
  tot_a = sum(v["algorithmic_bytes_per_launch"] * v["launches_per_step"] for v in out.values())
  out["_step"] = {"traffic_bytes": tot_t, "algorithmic_bytes": tot_a, "traffic_over_algorithmic": tot_t / tot_a,
                  "note": "FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 (KiB counters; gfx950 FETCH correction of MI355X_MICROARCH.md)"}
  out["_stamp"] = _stamp.stamp()
  json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
  main()
