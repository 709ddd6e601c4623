#!/usr/bin/env python
"""Is the 0.25 deg step power-bound?  Polls the GPU's socket power, power cap and shader clock (amd-smi / rocm-smi,
whichever the box has) every ~100 ms while the step runs back to back for a few seconds, and reports their distribution
next to the idle reading.  Evidence for DESIGN.md section 9.14 (round 5): a launch that gets faster per tile makes the
NEXT launch slower by the same energy -- profiles/r05_s11_*.

    python scripts/power_probe.py [--seconds 6] [--precision bf16] [--config 1deg_13L_M5] [--out gpurun_out/power_probe.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_once():
  """-> dict(power_w, cap_w, sclk_mhz, raw) from whichever tool answers."""
  out = {}
  for cmd in (["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"],
              ["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showmaxpower", "--json"]):
    try:
      r = subprocess.run(cmd, capture_output=True, text=True, timeout=5)
    except Exception as e:      # noqa: BLE001
      out.setdefault("errors", []).append(f"{cmd[0]}: {e}")
      continue
    if r.returncode != 0 or not r.stdout.strip():
      out.setdefault("errors", []).append(f"{cmd[0]}: rc {r.returncode} {r.stderr[:200]}")
      continue
    out["tool"] = cmd[0]
    out["raw"] = r.stdout
    txt = r.stdout
    m = re.search(r'"(?:socket_power|current_socket_power|Current Socket Graphics Package Power \(W\)|Average Graphics Package Power \(W\))"\s*:\s*(?:\{[^}]*"value"\s*:\s*)?"?([\d.]+)', txt)
    if m:
      out["power_w"] = float(m.group(1))
    m = re.search(r'"(?:Max Graphics Package Power \(W\)|power_cap|power_limit)"\s*:\s*(?:\{[^}]*"value"\s*:\s*)?"?([\d.]+)', txt)
    if m:
      out["cap_w"] = float(m.group(1))
    m = re.search(r'"sclk clock speed:"\s*:\s*"\((\d+)Mhz\)"', txt) or re.search(r'"gfx_0"\s*:\s*\{[^}]*?"clk"\s*:\s*(?:\{[^}]*"value"\s*:\s*)?"?(\d+)', txt, re.S)
    if m:
      out["sclk_mhz"] = float(m.group(1))
    return out
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--seconds", type=float, default=6.0)
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "power_probe.json"))
  ap.add_argument("--precision", default=None, choices=["f16x3", "f32", "bf16"])
  ap.add_argument("--config", default="0.25deg_37L_M6")
  args = ap.parse_args()
  import numpy as np
  import torch
  import bench as B
  from graphcast_amd import graphcast as gc
  idle = read_once()
  res, mesh_size, levels, gnn_steps = B.CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=gnn_steps, hidden_layers=1,
                       radius_query_fraction_edge_length=0.6)
  model = gc.GraphCast(cfg, task, params=B.fast_params(c_in, c_out, gnn_steps),
                       precision=args.precision).init_from_coordinates(lat, lon)
  x = torch.from_numpy(np.random.default_rng(0).standard_normal((len(lat) * len(lon), 1, c_in), dtype=np.float32)).cuda()
  y = model.forward_grid_node_features(x)
  torch.cuda.synchronize()
  samples, stop = [], threading.Event()

  def poll():
    while not stop.is_set():
      s = read_once()
      s.pop("raw", None)
      s["t"] = time.perf_counter()
      samples.append(s)
      time.sleep(0.05)

  th = threading.Thread(target=poll)
  th.start()
  t0 = time.perf_counter()
  n = 0
  while time.perf_counter() - t0 < args.seconds:
    for _ in range(10):
      model._engine(x, y)
    torch.cuda.synchronize()
    n += 10
  dt = time.perf_counter() - t0
  stop.set()
  th.join()
  pw = [s["power_w"] for s in samples if "power_w" in s]
  ck = [s["sclk_mhz"] for s in samples if "sclk_mhz" in s]
  raw_idle = idle.pop("raw", "")
  rep = {"config": args.config, "precision": model._engine.precision, "idle": idle, "idle_raw_head": raw_idle[:1500], "steps": n, "ms_per_step": 1e3 * dt / n, "samples": len(samples),
         "power_w": {"min": min(pw), "median": sorted(pw)[len(pw) // 2], "max": max(pw)} if pw else None,
         "sclk_mhz": {"min": min(ck), "median": sorted(ck)[len(ck) // 2], "max": max(ck)} if ck else None,
         "cap_w": next((s["cap_w"] for s in samples if "cap_w" in s), idle.get("cap_w")),
         "tool": next((s.get("tool") for s in samples if s.get("tool")), None),
         "errors": sorted({e for s in samples for e in s.get("errors", [])})[:4]}
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(rep, f, indent=1)
  print(json.dumps({k: v for k, v in rep.items() if k != "idle_raw_head"}))
  print("idle raw:", raw_idle[:600].replace("\n", " "))


if __name__ == "__main__":
  main()
