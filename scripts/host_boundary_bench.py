#!/usr/bin/env python
"""What the Dataset boundary costs when the caller hands over HOST buffers: GraphCast.__call__
(reference graphcast.py:298-329) on numpy-backed Datasets at 0.25 deg / 37 levels -- one H2D copy per variable
(1.96 GB in all), stacking to [N_grid, B, C_in] on the device, the step, un-stacking on the device, one D2H copy per
predicted variable (0.94 GB) -- against the same call on device-resident (torch-backed) Datasets.  bench.py's `value` never includes any of this.

    python scripts/host_boundary_bench.py [--iters 3] [--out gpurun_out/host_boundary.json]
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
import bench as B                                   # noqa: E402
from graphcast_amd import graphcast as gc           # noqa: E402
from graphcast_amd import synthetic                 # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--iters", type=int, default=3)
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "host_boundary.json"))
  args = ap.parse_args()
  res, mesh_size, levels, gnn_steps = B.CONFIGS["0.25deg_37L_M6"]
  task = gc.TASK
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=gnn_steps, hidden_layers=1,
                       radius_query_fraction_edge_length=0.6)
  model = gc.GraphCast(cfg, task, params=B.fast_params(c_in, c_out, gnn_steps)).init_from_coordinates(lat, lon)
  inputs, template, forcings = synthetic.make_example(task, lat, lon, num_target_steps=1)
  dev_inputs, dev_forcings = synthetic.to_device(inputs, "cuda:0"), synthetic.to_device(forcings, "cuda:0")

  def timed(fn):
    fn()                                            # warm-up (builds the engine the first time)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.iters):
      t0 = time.perf_counter()
      out = fn()
      torch.cuda.synchronize()
      ts.append(time.perf_counter() - t0)
      del out
    return float(np.median(ts))

  t_host = timed(lambda: model(inputs, template, forcings))
  t_dev = timed(lambda: model(dev_inputs, template, dev_forcings))
  x = torch.randn((len(lat) * len(lon), 1, c_in), device="cuda:0")
  y = torch.empty((len(lat) * len(lon), 1, c_out), device="cuda:0")
  t_step = timed(lambda: model.forward_grid_node_features(x, y))
  # reference-style use: the normalisation wrapper around the predictor, called on host Datasets -- with the
  # wrapper's Dataset arithmetic on the device (predictor_base.host_datasets_on_device: the outermost wrapper
  # uploads) and, for comparison, in numpy on the host (GCAST_WRAPPERS_ON_HOST=1)
  from graphcast_amd import normalization
  mean, std, dstd = synthetic.make_stats(task)
  wrapped = normalization.InputsAndResiduals(model, std, mean, dstd)
  t_wrapped = timed(lambda: wrapped(inputs, template, forcings))
  os.environ["GCAST_WRAPPERS_ON_HOST"] = "1"
  t_wrapped_host = timed(lambda: wrapped(inputs, template, forcings))
  del os.environ["GCAST_WRAPPERS_ON_HOST"]
  # the host call's pieces, timed one by one
  t_upload = timed(lambda: (model._upload(inputs), model._upload(forcings)))
  y_host = torch.empty(y.shape, dtype=y.dtype, pin_memory=True)
  t_d2h = timed(lambda: y_host.copy_(y))
  t_unstack = timed(lambda: model._grid_node_outputs_to_prediction(y_host.numpy(), template))
  t_stack = timed(lambda: model._inputs_to_grid_node_features(dev_inputs, dev_forcings))
  res_ = {"config": "GraphCast 0.25deg_37L_M6, batch 1, one 6-h step per call",
          "seconds_per_call": {"host_datasets_in_and_out (H2D per variable + device stack + step + device unstack + D2H per variable)": t_host,
                               "device_resident_datasets (stack + step + unstack on the device)": t_dev,
                               "tensor_boundary (forward_grid_node_features: what bench.py times)": t_step,
                               "InputsAndResiduals(GraphCast) on host datasets, wrapper arithmetic on the device": t_wrapped,
                               "InputsAndResiduals(GraphCast) on host datasets, wrapper arithmetic in numpy on the host": t_wrapped_host},
          "host_call_pieces_seconds": {"upload_variables": t_upload, "stack_on_device": t_stack, "d2h_into_pinned": t_d2h,
                                       "unstack_on_host": t_unstack},
          "bytes": {"h2d": 4 * x.numel(), "d2h": 4 * y.numel()},
          "steps_per_second": {"host_boundary": 1.0 / t_host, "device_datasets": 1.0 / t_dev, "tensor_boundary": 1.0 / t_step}}
  print(json.dumps(res_))
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(res_, f, indent=1)


if __name__ == "__main__":
  main()
