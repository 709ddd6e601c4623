#!/usr/bin/env python
"""Does replaying the step's ~80 launches as ONE hipGraph beat enqueueing them one by one?  (0.25 deg step, same
model as bench.py; eager and graph replays interleaved, HIP-event time per step.)"""
import json
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B                                   # noqa: E402
from graphcast_amd import graphcast as gc           # noqa: E402

res, mesh_size, levels, gnn_steps = B.CONFIGS[os.environ.get("CFG", "0.25deg_37L_M6")]
task = {37: gc.TASK, 13: gc.TASK_13}[levels]
c_out = gc.num_output_channels(task)
c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=gnn_steps, hidden_layers=1,
                     radius_query_fraction_edge_length=0.6)
model = gc.GraphCast(cfg, task, params=B.fast_params(c_in, c_out, gnn_steps)).init_from_coordinates(lat, lon)
n = len(lat) * len(lon)
x = torch.randn((n, 1, c_in), device="cuda:0")
y = torch.empty((n, 1, c_out), device="cuda:0")
y2 = torch.empty_like(y)
model.forward_grid_node_features(x, y)
engine = model._engine
torch.cuda.synchronize()

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
  for _ in range(2):
    engine(x, y2)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
  engine(x, y2)
graph.replay(); torch.cuda.synchronize()
assert torch.equal(y, y2), "graph replay differs"

def timed(fn, k=10):
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(k):
    fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / k

out = {"eager_ms": [], "graph_ms": []}
for _ in range(4):
  out["eager_ms"].append(timed(lambda: engine(x, y)))
  out["graph_ms"].append(timed(graph.replay))
print(json.dumps(out))
