#!/usr/bin/env python
"""Does the NUMBERING of the mesh nodes matter to the step?  The processor's edge update gathers two 512-float rows per
edge from tables of 40,962 rows (84 MB each at 0.25 deg): receiver-sorted edges make the receiver gathers sequential,
the SENDER gathers follow the mesh's numbering -- the reference's: coarse levels first, then every refinement's new
vertices in the order its faces create them (utils/icosahedral_mesh.py:321-363).  A step's output on the GRID does not
depend on how mesh nodes are labelled (only the fp32 order of the segment sums moves), so the question can be asked
without touching the library: the same model on graphs whose mesh nodes are renumbered along a space-filling curve
(Morton code of the unit-sphere position), A/B in one process, stage times by HIP events.

    python scripts/probes/mesh_order_probe.py [--config 0.25deg_37L_M6] [--out gpurun_out/.../mesh_order.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                       # noqa: E402
from graphcast_amd import engine                   # noqa: E402
from graphcast_amd import graphcast as gc          # noqa: E402
from graphcast_amd.launch import TAGS              # noqa: E402


def morton_rank(xyz, bits=10):
  q = np.clip(((xyz * 0.5 + 0.5) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
  code = np.zeros(len(xyz), np.int64)
  for b in range(bits):
    for a in range(3):
      code |= ((q[:, a] >> b) & 1) << (3 * b + a)
  order = np.argsort(code, kind="stable")            # order[new] = old
  new_of_old = np.empty_like(order)
  new_of_old[order] = np.arange(len(order))
  return order, new_of_old


def renumber(g, order, new_of_old):
  out = dict(g)
  out["mesh_node_feat"] = np.ascontiguousarray(np.asarray(g["mesh_node_feat"])[order])
  out["g2m"] = dict(g["g2m"], receivers=new_of_old[np.asarray(g["g2m"]["receivers"])])
  out["mesh"] = dict(g["mesh"], senders=new_of_old[np.asarray(g["mesh"]["senders"])],
                     receivers=new_of_old[np.asarray(g["mesh"]["receivers"])])
  out["m2g"] = dict(g["m2g"], senders=new_of_old[np.asarray(g["m2g"]["senders"])])
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--config", default="0.25deg_37L_M6")
  ap.add_argument("--precision", default="f16x3")
  ap.add_argument("--rounds", type=int, default=4)
  ap.add_argument("--steps", type=int, default=8)
  ap.add_argument("--out", default="")
  args = ap.parse_args()
  res, mesh_size, levels, gnn_steps = bench.CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=gnn_steps, hidden_layers=1,
                       radius_query_fraction_edge_length=0.6)
  params = bench.fast_params(c_in, c_out, gnn_steps)
  model = gc.GraphCast(cfg, task, params=params, device="cuda:0", precision=args.precision).init_from_coordinates(lat, lon)
  g = model.graph_arrays()
  xyz = np.asarray(model._finest_mesh.vertices, np.float64)
  order, new_of_old = morton_rank(xyz)
  variants = {"reference numbering": g, "morton numbering": renumber(g, order, new_of_old)}
  x = torch.from_numpy(np.random.default_rng(0).standard_normal((g["n_grid"], 1, c_in), dtype=np.float32)).to("cuda:0")
  engines, outs = {}, {}
  for name, graphs in variants.items():
    e = engine.StepEngine(graphs, params, num_steps=gnn_steps, c_in=c_in, c_out=c_out, device="cuda:0", precision=args.precision)
    outs[name] = e(x).clone()
    torch.cuda.synchronize()
    e.check_range()
    engines[name] = e
  names = list(variants)
  rel = float((outs[names[1]].double() - outs[names[0]].double()).norm() / outs[names[0]].double().norm())
  ms = {n: [] for n in names}
  for _ in range(args.rounds):                       # ABAB: both see the same clocks
    for n in names:
      e = engines[n]
      y = torch.empty_like(outs[n])
      e(x, y)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(args.steps):
        e(x, y)
      torch.cuda.synchronize()
      ms[n].append((time.perf_counter() - t0) / args.steps * 1e3)
  inv = {v: k for k, v in TAGS.items()}
  stages = {}
  for n in names:
    acc = {}
    for tag, kind, t in engines[n].time_ops(x, iters=3):
      acc[inv.get(tag, str(tag))] = acc.get(inv.get(tag, str(tag)), 0.0) + t
    stages[n] = {k: round(v, 3) for k, v in acc.items()}
  report = dict(config=args.config, precision=args.precision, rel_diff_of_the_outputs=rel,
                ms_per_step={n: [round(v, 3) for v in ms[n]] for n in names},
                ms_per_step_median={n: round(float(np.median(ms[n])), 3) for n in names}, stages_ms=stages)
  print(json.dumps(report))
  if args.out:
    with open(args.out, "w") as f:
      json.dump(report, f, indent=1)


if __name__ == "__main__":
  main()
