#!/usr/bin/env python
"""BASELINE.json config 5: ONE 0.25 deg step strong-scaled over N GPUs -- the icosahedral
multi-mesh and the grid partitioned by octant (hemispheres / quadrants for 2 / 4 ranks; partition.plan), receiver-owned edges, 18 halo exchanges
per step (one RCCL all_to_all_single each).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        scripts/partition_bench.py --steps 5 [--config 0.25deg_37L_M6]

Every rank builds the (deterministic) global graphs on the host, keeps only its part, and runs
its local engine; `value` = steps/s of the whole partitioned step (max over ranks).  With
N = 1 it degenerates to the unpartitioned step.  (Validated numerically on one GPU by
tests/test_partition_gpu.py with emulated ranks; the exchanger by a gloo world_size-2 test.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                   # noqa: E402  (configs, params)
from graphcast_amd import graphcast as gc           # noqa: E402
from graphcast_amd import partition                 # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=1)
  ap.add_argument("--config", default="0.25deg_37L_M6", choices=sorted(B.CONFIGS))
  args = ap.parse_args()
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  torch.cuda.set_device(local_rank)
  device = f"cuda:{local_rank}"
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("MASTER_PORT", "29511")
  dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

  res, mesh_size, levels, gnn_steps = B.CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=B.LATENT,
                       gnn_msg_steps=gnn_steps, hidden_layers=1, radius_query_fraction_edge_length=0.6)
  params = B.fast_params(c_in, c_out, gnn_steps)
  model = gc.GraphCast(cfg, task, params=params, device=device).init_from_coordinates(lat, lon)
  g = model.graph_arrays()
  mine = partition.plan(g, model._grid_nodes_lon, model._mesh_nodes_lon, world, grid_lat=model._grid_nodes_lat,
                        mesh_lat=model._mesh_nodes_lat)[rank]
  step = partition.DistributedPartitionedStep(mine, params, num_steps=gnn_steps, c_in=c_in, c_out=c_out,
                                              device=device)
  x = torch.from_numpy(np.random.default_rng(0).standard_normal(
      (g["n_grid"], 1, c_in), dtype=np.float32)[mine.grid_owned]).to(device)
  y = torch.empty((mine.n_grid_owned, 1, c_out), dtype=torch.float32, device=device)
  for _ in range(args.warmup):
    step(x, y)
  dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step(x, y)
  dist.barrier()
  torch.cuda.synchronize()
  t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  if rank == 0:
    halo = {k: int(len(pl.halo_global)) for k, (pl, _) in partition.tables_of(mine).items()}
    print(json.dumps({
        "metric": "6-h rollout steps/sec at 0.25deg/37-level (one step partitioned over N GPUs)",
        "value": args.steps / float(t.item()), "unit": "steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(t.item()) / args.steps,
        "higher_is_better": True, "scaling": "strong", "data": "synthetic",
        "config": {"workload": f"GraphCast {args.config}, octant partition x{world}, "
                               f"receiver-owned edges, 18 halo exchanges per step",
                   "rank0_rows": {"grid": mine.n_grid_owned, "mesh": mine.n_mesh_owned, "halo": halo}},
        "finite": bool(torch.isfinite(y).all().item())}))
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
