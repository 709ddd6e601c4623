#!/usr/bin/env python
"""BASELINE.json config 3: 0.25 deg / 37-level autoregressive rollout (default 40 x 6 h) on one
MI355X with everything resident in HBM (rollout_device.DeviceRollout: step + one fused
gc_advance_state per step; the de-normalised trajectory stays on the device).

    python scripts/rollout_bench.py [--steps 40] [--res 0.25] [--out gpurun_out/rollout.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphcast_amd import graphcast as gc          # noqa: E402
from graphcast_amd import params as gparams        # noqa: E402
from graphcast_amd import rollout_device           # noqa: E402
from graphcast_amd import synthetic                # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=40)
  ap.add_argument("--res", type=float, default=0.25)
  ap.add_argument("--mesh", type=int, default=6)
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rollout.json"))
  ap.add_argument("--precision", default=None, help='None = f16x3 (fp32-grade); "bf16" = the Bfloat16Cast tier')
  args = ap.parse_args()
  task = gc.TASK
  lat = np.arange(-90, 90 + args.res / 2, args.res)
  lon = np.arange(0, 360, args.res)
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * len(task.pressure_levels)) + 2 * 5 + 2 + 5
  cfg = gc.ModelConfig(resolution=args.res, mesh_size=args.mesh, latent_size=512, gnn_msg_steps=16,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  t0 = time.perf_counter()
  model = gc.GraphCast(cfg, task, params=gparams.random_params(c_in, c_out, 512, 16))
  model.set_precision(args.precision)
  inputs, template, forcings = synthetic.make_example(task, lat, lon, num_target_steps=args.steps)
  mean, std, dstd = synthetic.make_stats(task)
  roll = rollout_device.DeviceRollout(model, std, mean, dstd)
  traj = roll.run(inputs, template.isel(time=slice(0, 1)), forcings.isel(time=slice(0, 1)))   # builds the plan
  torch.cuda.synchronize()
  setup = time.perf_counter() - t0
  del traj
  t0 = time.perf_counter()
  traj = roll.run(inputs, template, forcings)
  torch.cuda.synchronize()
  total = time.perf_counter() - t0
  loop_ms = roll.last_loop_ms()          # the device loop alone: steps + fused state advances
  advance_ms = roll.advance_ms()
  n_rows = traj.shape[1] * traj.shape[2]
  advance_bytes = 4.0 * n_rows * (2 * c_in + 2 * c_out + 2 * 5)      # x in, x_next out, y in, prediction out, forcings
  finite = bool(torch.isfinite(traj).all().item())
  res = {"config": f"GraphCast {args.res} deg / {len(task.pressure_levels)} levels / M{args.mesh}, "
                   f"{args.steps} x 6 h autoregressive rollout, HBM-resident (DeviceRollout)",
         "steps": args.steps, "device_loop_ms": loop_ms, "ms_per_step": loop_ms / args.steps,
         "advance_state_ms": advance_ms, "advance_state_gb_per_s": advance_bytes / advance_ms / 1e6,
         "steps_per_second": 1e3 * args.steps / loop_ms,
         "seconds_total_including_host_prep_and_upload": total,
         "trajectory_gb_in_hbm": traj.numel() * 4 / 1e9, "finite": finite,
         "setup_seconds": setup, "precision": model._engine.precision,
         "hbm_allocated_gb": torch.cuda.max_memory_allocated() / 1e9}
  print(json.dumps(res))
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(res, f, indent=1)


if __name__ == "__main__":
  main()
