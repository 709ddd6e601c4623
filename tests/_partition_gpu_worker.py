"""Worker of tests/test_partition_gpu.py::test_distributed_partitioned_step_two_ranks_share_one_gpu:
one process per rank, BOTH on cuda:0, gloo process group (RCCL refuses two ranks on one GPU), halo
rows exchanged through host buffers.  Runs partition.DistributedPartitionedStep -- the class an
8-GPU box would run -- and writes this rank's output rows."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphcast_amd import graphcast as gc    # noqa: E402
from graphcast_amd import partition          # noqa: E402
from oracle import params as oparams         # noqa: E402


def main():
  out_dir = sys.argv[1]
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  dist.init_process_group("gloo", rank=rank, world_size=world)
  res, mesh_size, steps = 4.0, 3, 3
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = oparams.init_params(c_in, c_out, 512, steps, seed=1, nontrivial=True)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params).init_from_coordinates(lat, lon)
  me = partition.plan(model.graph_arrays(), model._grid_nodes_lon, model._mesh_nodes_lon, world,
                      grid_lat=model._grid_nodes_lat, mesh_lat=model._mesh_nodes_lat)[rank]
  x = np.random.default_rng(0).standard_normal((len(lat) * len(lon), 2, c_in)).astype(np.float32)
  step = partition.DistributedPartitionedStep(me, params, num_steps=steps, c_in=c_in, c_out=c_out,
                                              device="cuda:0")
  assert all(e.host_staged for e in step.exchangers.values())
  x_local = torch.from_numpy(np.ascontiguousarray(x[me.grid_owned])).to("cuda:0")
  for _ in range(2):                     # twice: exchanges repeat every step, tables are reused
    y_local = step(x_local)
  torch.cuda.synchronize()
  np.save(os.path.join(out_dir, f"y_rank{rank}.npy"), y_local.cpu().numpy())
  np.save(os.path.join(out_dir, f"rows_rank{rank}.npy"), np.asarray(me.grid_owned))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
