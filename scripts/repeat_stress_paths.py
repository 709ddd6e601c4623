#!/usr/bin/env python
"""Repeatability of the paths AROUND the single step (scripts/repeat_stress.py covers the step itself): every result
compared BITWISE with the first of its kind.
  * the emulated 8-way partitioned step at 0.25 deg (each rank's small launches run in the helper / four-wave forms the
    unpartitioned headline step never uses), f16x3 and bf16;
  * the HBM-resident autoregressive rollout at 0.25 deg (step + fused state advance, K steps), f16x3 and bf16;
  * the exact-fp32 chunked kernels at 1 deg.

    python scripts/repeat_stress_paths.py [--reps 6] [--rollout-steps 8] [--out gpurun_out/repeat_stress_paths.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench as B                                   # noqa: E402
import repeat_stress                                # noqa: E402
from graphcast_amd import graphcast as gc           # noqa: E402
from graphcast_amd import params as gparams         # noqa: E402
from graphcast_amd import partition, rollout_device, synthetic      # noqa: E402


def compare(make, reps):
  first = make().clone()
  torch.cuda.synchronize()
  differing, worst = 0, 0.0
  for _ in range(reps):
    y = make()
    torch.cuda.synchronize()
    if not torch.equal(y, first):
      differing += 1
      worst = max(worst, float((y.float() - first.float()).abs().max()))
  return {"reps": reps, "runs_differing_from_the_first": differing, "max_abs_diff": worst,
          "finite": bool(torch.isfinite(first.float()).all())}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--reps", type=int, default=6)
  ap.add_argument("--rollout-steps", type=int, default=8)
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "repeat_stress_paths.json"))
  args = ap.parse_args()
  out = []

  def report(what, r):
    r = dict(what=what, **r)
    out.append(r)
    print(r, flush=True)

  # ---- the emulated 8-way partition at 0.25 deg
  for precision in ("f16x3", "bf16"):
    model, x, _ = repeat_stress.build("0.25deg_37L_M6", precision)
    g = model.graph_arrays()
    res, mesh_size, levels, gnn_steps = B.CONFIGS["0.25deg_37L_M6"]
    c_out = gc.num_output_channels(gc.TASK)
    c_in = x.shape[2]
    params = B.fast_params(c_in, c_out, gnn_steps)
    step = partition.EmulatedPartitionedStep(g, params, model._grid_nodes_lon, model._mesh_nodes_lon, 8, num_steps=gnn_steps,
                                             c_in=c_in, c_out=c_out, grid_lat=model._grid_nodes_lat,
                                             mesh_lat=model._mesh_nodes_lat, precision=precision)
    report(f"emulated 8-way partitioned step, 0.25 deg, {precision}", compare(lambda: step(x), args.reps))
    del step, model, x
    torch.cuda.empty_cache()

  # ---- the HBM-resident rollout at 0.25 deg
  task = gc.TASK
  lat, lon = np.arange(-90, 90 + 0.125, 0.25), np.arange(0, 360, 0.25)
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * len(task.pressure_levels)) + 2 * 5 + 2 + 5
  cfg = gc.ModelConfig(resolution=0.25, mesh_size=6, latent_size=512, gnn_msg_steps=16, hidden_layers=1,
                       radius_query_fraction_edge_length=0.6)
  inputs, template, forcings = synthetic.make_example(task, lat, lon, num_target_steps=args.rollout_steps)
  mean, std, dstd = synthetic.make_stats(task)
  for precision in (None, "bf16"):
    model = gc.GraphCast(cfg, task, params=gparams.random_params(c_in, c_out, 512, 16))
    model.set_precision(precision)
    roll = rollout_device.DeviceRollout(model, std, mean, dstd)
    report(f"HBM-resident rollout, {args.rollout_steps} steps, 0.25 deg, {precision or 'f16x3'}",
           compare(lambda: roll.run(inputs, template, forcings), max(2, args.reps // 2)))
    del roll, model
    torch.cuda.empty_cache()

  # ---- exact fp32 at 1 deg
  model, x, y = repeat_stress.build("1deg_13L_M5", "f32")
  engine = model._engine

  def f32_step():
    y.fill_(float("nan"))
    engine(x, y)
    return y
  report("exact-fp32 chunked kernels, 1 deg step", compare(f32_step, 4 * args.reps))

  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(out, f, indent=1)
  sys.exit(1 if any(r["runs_differing_from_the_first"] or not r["finite"] for r in out) else 0)


if __name__ == "__main__":
  main()
