#!/usr/bin/env python
"""Where a SMALL step's time goes between its kernels: from a rocprofv3 --kernel-trace CSV of `bench.py --steps K`, the
last K steps' dispatches in start order -- per step the span from the first kernel's start to the last one's end, the
sum of the kernel durations, and the idle gaps between consecutive kernels grouped by (previous kernel -> next kernel).

    python scripts/kernel_gaps.py gpurun_out/<run>/prof [--steps 3] [--skip-tail 3] > profiles/<name>.json

--skip-tail: whole step patterns at the END of the trace that are not timed steps (bench.py's per-launch timing pass,
gc_time_program, runs the same kernels one by one behind the timed steps).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
  m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?(\w+)(<[^>]*>)?", name)
  return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
  root = sys.argv[1]
  steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
  skip = int(sys.argv[sys.argv.index("--skip-tail") + 1]) if "--skip-tail" in sys.argv else 3
  rows = []
  for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
      for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
  rows.sort()
  # a step starts with the grid embedder's input preparation: the first kernel of the repeating pattern = the kernel
  # that follows the longest idle gaps; simpler: split the tail into `steps` equal runs between the K largest gaps
  gaps = sorted(range(1, len(rows)), key=lambda i: rows[i][0] - rows[i - 1][1], reverse=True)
  # the dispatch count per step: distance between the starts of the last two occurrences of the most frequent period
  names = [r[2] for r in rows]
  period = None
  for p in range(20, len(names) // 2):
    if names[-p:] == names[-2 * p:-p]:
      period = p
      break
  if period is None:
    raise SystemExit("no repeating step pattern at the end of the trace")
  if skip:
    rows = rows[:len(rows) - skip * period]
  out = {"dispatches_per_step": period, "skipped_patterns_at_the_end": skip, "steps": []}
  by_pair = collections.defaultdict(list)
  for s in range(steps, 0, -1):
    seg = rows[len(rows) - s * period: len(rows) - (s - 1) * period]
    span = (seg[-1][1] - seg[0][0]) / 1e3
    busy = sum(e - b for b, e, _ in seg) / 1e3
    out["steps"].append({"span_us": round(span, 1), "kernels_us": round(busy, 1), "gaps_us": round(span - busy, 1)})
    for a, b in zip(seg, seg[1:]):
      by_pair[(a[2], b[2])].append((b[0] - a[1]) / 1e3)
  out["gaps_by_pair_us"] = {f"{a} -> {b}": {"n_per_step": len(v) / steps, "mean": round(sum(v) / len(v), 2), "max": round(max(v), 2),
                                            "total_per_step": round(sum(v) / steps, 1)}
                            for (a, b), v in sorted(by_pair.items(), key=lambda kv: -sum(kv[1]))}
  kern = collections.defaultdict(list)
  for b, e, n in rows[len(rows) - steps * period:]:
    kern[n].append((e - b) / 1e3)
  out["kernels_us"] = {n: {"n_per_step": len(v) / steps, "mean": round(sum(v) / len(v), 2), "total_per_step": round(sum(v) / steps, 1)}
                       for n, v in sorted(kern.items(), key=lambda kv: -sum(kv[1]))}
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()
