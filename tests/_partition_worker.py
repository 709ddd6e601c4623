"""Worker of tests/test_partition.py::test_dist_exchanger_over_gloo (one process per rank)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphcast_amd import partition          # noqa: E402
from oracle import graphcast as ogc          # noqa: E402


def main():
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  dist.init_process_group("gloo", rank=rank, world_size=world)
  res, mesh = 6.0, 2
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  g = ogc.build_graphs(lat, lon, mesh)
  glon = np.meshgrid(lon, lat)[0].reshape(-1)
  glat = np.meshgrid(lon, lat)[1].reshape(-1)
  me = partition.plan(g, glon, np.asarray(g["mesh_lon"]), world, grid_lat=glat,
                      mesh_lat=np.asarray(g["mesh_lat"]))[rank]            # (2 ranks: hemispheres)
  rng = np.random.default_rng(0)                     # same global tables on every rank
  for name, key, n_glob, owned in (("g2m", "g2m", g["n_grid"], me.grid_owned),
                                   ("mesh", "mesh", g["n_mesh"], me.mesh_owned),
                                   ("m2g", "m2g", g["n_mesh"], me.mesh_owned)):
    table = rng.standard_normal((n_glob, 16)).astype(np.float32)
    pl, n_owned = partition.tables_of(me)[name]
    t = torch.full((n_owned + len(pl.halo_global), 16), float("nan"))
    t[:n_owned] = torch.from_numpy(table[owned])
    for _ in range(2):                               # exchanges repeat every step
      partition.DistExchanger(pl, n_owned, "cpu").exchange(t)
    e = me.graphs[key]
    want = table[np.asarray(g[key]["senders"])[e["edge_ids"]]]
    np.testing.assert_array_equal(t.numpy()[e["senders"]], want)
  dist.barrier()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
