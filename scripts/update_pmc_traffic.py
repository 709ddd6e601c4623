#!/usr/bin/env python
"""profiles/pmc_traffic.json["f16x3h:proc_edge"] (what bench.py attaches as roofline.traffic) from a
per-stage PMC summary written by scripts/pmc_by_stage.py.

    python scripts/update_pmc_traffic.py profiles/r03_..._pmc_by_stage.json [key]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  src = sys.argv[1]
  key = sys.argv[2] if len(sys.argv) > 2 else "f16x3h:proc_edge"
  stage = key.split(":")[1]
  with open(src) as f:
    s = json.load(f)[stage]
  path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
  with open(path) as f:
    table = json.load(f)
  table[key] = {
      "bytes_per_launch": s["traffic_bytes_per_launch"],
      "fetch_bytes_per_launch": s["fetch_bytes_per_launch"],
      "write_bytes_per_launch": s["write_bytes_per_launch"],
      "algorithmic_bytes": s["algorithmic_bytes_per_launch"],
      "source": f"{os.path.relpath(os.path.abspath(src), ROOT)} (scripts/pmc_by_stage.py: the {stage} launches of the last "
                "step of separate FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 per MI355X_MICROARCH.md; the counters sit "
                "at the L2 <-> fabric boundary, so Infinity-Cache hits -- the weight stream's L2 misses, the parked "
                "accumulators -- are included)",
  }
  with open(path, "w") as f:
    json.dump(table, f, indent=1)
  print(key, json.dumps(table[key])[:300])


if __name__ == "__main__":
  main()
