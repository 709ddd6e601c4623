#!/usr/bin/env python
"""Config 5 on ONE GPU: the 0.25 deg step partitioned into P regions (octants for 8), all P ranks
emulated in this process (partition.EmulatedPartitionedStep).  Reports the per-rank row / halo
counts of the real 0.25 deg / M6 graphs, agreement with the unpartitioned step, and the summed
device time of the P local steps (what P GPUs would each do 1/P of, plus 18 exchanges).

    python scripts/partition_emulated_bench.py [--parts 8] [--out gpurun_out/partition.json]
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
from graphcast_amd import partition                 # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--parts", type=int, default=8)
  ap.add_argument("--config", default="0.25deg_37L_M6", choices=sorted(B.CONFIGS))
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "partition.json"))
  args = ap.parse_args()
  res, mesh_size, levels, gnn_steps = B.CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=B.LATENT,
                       gnn_msg_steps=gnn_steps, hidden_layers=1, radius_query_fraction_edge_length=0.6)
  params = B.fast_params(c_in, c_out, gnn_steps)
  model = gc.GraphCast(cfg, task, params=params).init_from_coordinates(lat, lon)
  g = model.graph_arrays()
  x = torch.from_numpy(np.random.default_rng(0).standard_normal((g["n_grid"], 1, c_in), dtype=np.float32)).cuda()
  y_full = model.forward_grid_node_features(x).clone()

  def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

  ms_full = timed(lambda: model.forward_grid_node_features(x))
  t0 = time.perf_counter()
  step = partition.EmulatedPartitionedStep(g, params, model._grid_nodes_lon, model._mesh_nodes_lon,
                                           args.parts, num_steps=gnn_steps, c_in=c_in, c_out=c_out,
                                           grid_lat=model._grid_nodes_lat, mesh_lat=model._mesh_nodes_lat)
  build_s = time.perf_counter() - t0
  y = step(x)
  torch.cuda.synchronize()
  rel = float(torch.linalg.vector_norm((y - y_full).double()) / torch.linalg.vector_norm(y_full.double()))
  ms_part = timed(lambda: step(x))
  ranks = step.ranks
  rows = lambda f: [int(f(r)) for r in ranks]
  # the 18 exchanges alone, back to back (what the blocking exchanges add to the step)
  tables = {n: [e.halo_table(n) for e in step.engines] for n in ("g2m", "mesh", "m2g")}
  def exchanges_only():
    step.exchangers["g2m"].exchange(tables["g2m"])
    for _ in range(gnn_steps):
      step.exchangers["mesh"].exchange(tables["mesh"])
    step.exchangers["m2g"].exchange(tables["m2g"])
  ms_exchange = timed(exchanges_only)
  # the emulation's own shuffle of the global input into the ranks' rows and of their outputs back (a real rank holds its rows)
  owned = step._owned_rows(x.device)
  y_parts = [y.index_select(0, o) for o in owned]
  def shuffle_only():
    for o in owned:
      x.index_select(0, o)
    for o, yl in zip(owned, y_parts):
      y.index_copy_(0, o, yl)
  ms_shuffle = timed(shuffle_only)
  out = {
      "config": f"GraphCast {args.config}, {args.parts} parts ({'octants' if args.parts == 8 else 'hemispheres / quadrants' if args.parts in (2, 4) else 'longitude bands'}), receiver-owned edges",
      "rel_diff_vs_unpartitioned": rel, "exchanges_per_step": step.exchanges_per_call,
      "ms_unpartitioned_step": ms_full, "ms_sum_of_all_ranks_emulated_on_one_gpu": ms_part,
      "ms_per_rank_if_perfectly_parallel": ms_part / args.parts,
      "ms_all_exchanges_alone_all_ranks": ms_exchange,
      "ms_emulation_input_output_shuffle_all_ranks": ms_shuffle,
      "ms_per_rank_local_launches_only": (ms_part - ms_exchange - ms_shuffle) / args.parts,
      "exchange_form": "per rank one packing index_select + one landing index_select (LocalExchanger, round 6); "
                       "a real rank: index_select + all_to_all_single (bench.py --mode partition: roofline.exchange)",
      "grid_rows_per_rank": rows(lambda r: r.n_grid_owned), "mesh_rows_per_rank": rows(lambda r: r.n_mesh_owned),
      "edges_per_rank": {k: rows(lambda r, k=k: len(r.graphs[k]["senders"])) for k in ("g2m", "mesh", "m2g")},
      "halo_rows_per_rank": {"g2m_grid_rows": rows(lambda r: len(r.halo_g2m.halo_global)),
                             "mesh_rows_per_processor_step": rows(lambda r: len(r.halo_mesh.halo_global)),
                             "m2g_mesh_rows": rows(lambda r: len(r.halo_m2g.halo_global))},
      "halo_mb_per_rank_per_processor_step": [round(len(r.halo_mesh.halo_global) * 2048 / 1e6, 3) for r in ranks],
      "plan_and_engines_build_seconds": build_s}
  print(json.dumps(out))
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(out, f, indent=1)


if __name__ == "__main__":
  main()
