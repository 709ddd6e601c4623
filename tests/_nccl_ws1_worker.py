"""Worker of tests/test_partition_gpu.py::test_rccl_path_executes_on_the_gpu: ONE process, the `nccl` (= RCCL)
backend with world_size 1 on the MI355X.  Executes what no GPU lease could execute before (one GPU per lease, RCCL
refuses two ranks on one device): partition.DistExchanger.exchange through torch.distributed.all_to_all_single ON
DEVICE TENSORS (no host staging) -- alone, under the overlap stream pattern of DistributedPartitionedStep -- and a
whole DistributedPartitionedStep (one part: every halo exchange is an all_to_all_single with empty splits) against
the unpartitioned engine."""
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
  dev = torch.device("cuda:0")
  torch.cuda.set_device(dev)
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
  assert dist.get_backend() == "nccl"
  # ---- the exchanger: rank 0 "sends" 300 of its owned rows to itself -> they land in the halo suffix
  n_owned, n_halo = 1000, 300
  rng = np.random.default_rng(0)
  idx = rng.choice(n_owned, n_halo, replace=False)
  plan = partition.HaloPlan(np.arange(n_halo), [n_halo], [idx])
  ex = partition.DistExchanger(plan, n_owned, dev)
  assert not ex.host_staged                                # device tensors straight into RCCL
  table = torch.randn((n_owned + n_halo, 512), device=dev)
  table[n_owned:] = float("nan")
  ex.exchange(table)
  torch.cuda.synchronize()
  assert torch.equal(table[n_owned:], table[torch.as_tensor(idx, device=dev)])
  # ... and on a second stream behind the producing work, the consumer waiting for it (GCAST_OVERLAP's pattern)
  comm, compute = torch.cuda.Stream(device=dev), torch.cuda.current_stream(dev)
  table[:n_owned] = torch.randn((n_owned, 512), device=dev)        # "the producing launch"
  table[n_owned:] = float("nan")
  comm.wait_stream(compute)
  with torch.cuda.stream(comm):
    ex.exchange(table)
  filler = torch.randn((2048, 2048), device=dev) @ torch.randn((2048, 2048), device=dev)   # work under the exchange
  compute.wait_stream(comm)
  got = table[n_owned:].clone()
  torch.cuda.synchronize()
  assert torch.equal(got, table[torch.as_tensor(idx, device=dev)]) and torch.isfinite(filler).all()
  # ---- a whole DistributedPartitionedStep on the nccl group
  res, mesh_size, steps = 4.0, 3, 2
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = oparams.init_params(c_in, c_out, 512, steps, seed=1, nontrivial=True)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params).init_from_coordinates(lat, lon)
  me = partition.plan(model.graph_arrays(), model._grid_nodes_lon, model._mesh_nodes_lon, 1,
                      grid_lat=model._grid_nodes_lat, mesh_lat=model._mesh_nodes_lat)[0]
  x = torch.from_numpy(np.random.default_rng(1).standard_normal((len(lat) * len(lon), 1, c_in)).astype(np.float32)).to(dev)
  step = partition.DistributedPartitionedStep(me, params, num_steps=steps, c_in=c_in, c_out=c_out, device=dev)
  assert not any(e.host_staged for e in step.exchangers.values())
  y = step(x[torch.as_tensor(me.grid_owned, device=dev)].contiguous())
  want = model.forward_grid_node_features(x)
  torch.cuda.synchronize()
  full = torch.empty_like(want)
  full[torch.as_tensor(me.grid_owned, device=dev)] = y
  rel = float(torch.linalg.vector_norm((full - want).double()) / torch.linalg.vector_norm(want.double()))
  assert rel <= 2e-6, rel
  print(f"NCCL_WS1_OK exchanger on device tensors + overlap stream + DistributedPartitionedStep (rel {rel:.1e})")
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
