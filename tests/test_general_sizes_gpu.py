"""ModelConfig.latent_size / hidden_layers other than the published 512 / 1
(``weathernext1_graph/graphcast.py:123-124,138-139``; ``utils/legacy/deep_typed_graph_net.py:205-209``:
``hk.nets.MLP([mlp_hidden_size] * mlp_num_hidden_layers + [latent_size])``) against the oracle, which restates the
MLP for any number of Linear layers and any width (``oracle/gnn.py: _Net.apply``).

The kernels' tile stays 512 columns wide: a narrower latent runs through padded parameters (csrc/gcast_plan.inc:
pad_latent -- zero-padded latent axes, the LayerNorm-fed output columns replicated (plus, where L does not divide 512, the
mean column and a rescaling LayerNorm's scale invariance undoes) so that the statistics over 512 columns are the
statistics over the L real ones), further hidden layers as further launches of the same kernels
(csrc/gcast_plan.inc: push_mlp).  Tolerances: those of tests/test_step_gpu.py (fp32-grade tiers) and of
tests/test_bf16_tier_gpu.py (the Bfloat16Cast tier against its op-by-op restatement).
"""
import numpy as np
import pytest
import torch

from graphcast_amd import graphcast as gc
from oracle import gnn as ognn
from oracle import graphcast as ogc
from oracle import params as oparams

pytestmark = pytest.mark.gpu

REL_RMSE_TOL = 2e-5       # tests/test_step_gpu.py


def rel_rmse(got, want):
  return float(np.linalg.norm(np.asarray(got, np.float64) - want) / np.linalg.norm(want))


def general_params(c_in, c_out, latent, steps, hidden_layers, seed=3):
  """oracle.params.init_params for `hidden_layers` hidden layers of `latent` units (non-trivial biases / LayerNorms)."""
  rng = np.random.default_rng(seed)
  params = {}
  for stem, sizes, ln in oparams.module_specs(c_in, c_out, latent, steps):
    sizes = [sizes[0]] + [latent] * hidden_layers + [sizes[-1]]
    for k in range(len(sizes) - 1):
      fan_in, fan_out = sizes[k], sizes[k + 1]
      w = np.clip(rng.standard_normal((fan_in, fan_out)), -2, 2) / np.sqrt(fan_in)
      params[f"{stem}_mlp/~/linear_{k}"] = {"w": w.astype(np.float32),
                                            "b": (0.1 * rng.standard_normal(fan_out)).astype(np.float32)}
    if ln:
      params[f"{stem}_layer_norm"] = {"scale": (1 + 0.1 * rng.standard_normal(sizes[-1])).astype(np.float32),
                                      "offset": (0.1 * rng.standard_normal(sizes[-1])).astype(np.float32)}
  return params


def build(latent, hidden_layers, precision, c_in=183):
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  res, mesh_size, steps = 4.0, 3, 3
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=latent, gnn_msg_steps=steps,
                       hidden_layers=hidden_layers, radius_query_fraction_edge_length=0.6)
  c_out = gc.num_output_channels(gc.TASK_13)
  params = general_params(c_in, c_out, latent, steps, hidden_layers)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params, precision=precision).init_from_coordinates(lat, lon)
  return model, ogc.build_graphs(lat, lon, mesh_size), params, steps, c_in


CASES = [(256, 1), (64, 1), (128, 1), (512, 2), (512, 3), (256, 2), (384, 1), (100, 2), (320, 1)]


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("latent,hidden_layers", CASES)
def test_step_of_a_general_size_matches_the_oracle(latent, hidden_layers, precision):
  model, graphs, params, steps, c_in = build(latent, hidden_layers, precision)
  for batch in (1, 2):
    x = np.random.default_rng(batch).standard_normal((graphs["n_grid"], batch, c_in)).astype(np.float32)
    want = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
    y = model.forward_grid_node_features(torch.from_numpy(x).to("cuda:0"))
    y2 = model.forward_grid_node_features(torch.from_numpy(x).to("cuda:0"))
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    err = rel_rmse(got, want)
    print(f"GENERAL_SIZE latent={latent} hidden_layers={hidden_layers} {precision} batch={batch}: rel-RMSE vs the float64 "
          f"oracle {err:.3e}")
    assert np.isfinite(got).all() and got.shape == want.shape
    assert err <= REL_RMSE_TOL
    assert torch.equal(y, y2)                     # deterministic, as every other launch sequence of the library


@pytest.mark.parametrize("latent,hidden_layers", [(256, 1), (64, 1), (512, 2), (128, 3), (384, 1), (200, 2)])
def test_bf16_tier_of_a_general_size(latent, hidden_layers):
  """The Bfloat16Cast tier: against the fp64 truth the step must be as good as the op-by-op bf16 restatement of the
  reference is (tests/test_bf16_tier_gpu.py's criterion)."""
  model, graphs, params, steps, c_in = build(latent, hidden_layers, "bf16")
  x = np.random.default_rng(5).standard_normal((graphs["n_grid"], 1, c_in)).astype(np.float32)
  got = model.forward_grid_node_features(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
  truth = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
  with ognn.activations("bf16"):
    ref_bf16 = ogc.forward(params, graphs, x, steps=steps, dtype=np.float32, f32_aggregation=True)
  e_hip, e_ref = rel_rmse(got, truth), rel_rmse(ref_bf16, truth)
  print(f"GENERAL_SIZE_BF16 latent={latent} hidden_layers={hidden_layers}: HIP vs truth {e_hip:.3e}, restatement vs truth "
        f"{e_ref:.3e}")
  assert np.isfinite(got).all()
  assert e_hip <= 1.25 * e_ref


def test_sizes_the_tile_cannot_hold_are_rejected_loudly():
  for latent in (1024, 513, 0):
    cfg = gc.ModelConfig(resolution=4.0, mesh_size=3, latent_size=latent, gnn_msg_steps=1, hidden_layers=1,
                         radius_query_fraction_edge_length=0.6)
    with pytest.raises(NotImplementedError, match="1 .. 512"):
      gc.GraphCast(cfg, gc.TASK_13, params={})
  cfg = gc.ModelConfig(resolution=4.0, mesh_size=3, latent_size=512, gnn_msg_steps=1, hidden_layers=0,
                       radius_query_fraction_edge_length=0.6)
  with pytest.raises(NotImplementedError, match="hidden_layers"):
    gc.GraphCast(cfg, gc.TASK_13, params={})


@pytest.mark.parametrize("latent,hidden_layers,precision", [(256, 2, "f16x3"), (384, 1, "f16x3"), (128, 2, "bf16")])
def test_partitioned_step_of_a_general_size(latent, hidden_layers, precision):
  """The spatially partitioned step (tests/test_partition_gpu.py) on such a model: every rank's plan re-shapes the
  same parameters, the halo exchanges sit in front of the launches that GATHER (the first of an edge MLP's n launches:
  engine.segments), so their number per step is what it is for the published architecture."""
  from graphcast_amd import partition
  model, graphs, params, steps, c_in = build(latent, hidden_layers, precision)
  x = torch.from_numpy(np.random.default_rng(4).standard_normal((graphs["n_grid"], 1, c_in)).astype(np.float32)).to("cuda:0")
  y_full = model.forward_grid_node_features(x).clone()
  step = partition.EmulatedPartitionedStep(
      model.graph_arrays(), params, model._grid_nodes_lon, model._mesh_nodes_lon, 4, num_steps=steps, c_in=c_in,
      c_out=gc.num_output_channels(gc.TASK_13), precision=precision, grid_lat=model._grid_nodes_lat,
      mesh_lat=model._mesh_nodes_lat)
  y = step(x)
  torch.cuda.synchronize()
  assert step.exchanges_per_call == 2 + steps
  rel = float(torch.linalg.vector_norm((y - y_full).double()) / torch.linalg.vector_norm(y_full.double()))
  print(f"GENERAL_SIZE_PARTITION latent={latent} hidden_layers={hidden_layers} {precision}: 4 parts vs unpartitioned {rel:.2e}")
  assert torch.isfinite(y).all()
  assert rel < (2e-2 if precision == "bf16" else 2e-6)


def test_the_c_hosts_two_calls_on_a_general_size():
  """gc_plan_create + gc_step_forward (plan.NativePlan: what a C / C++ host drives, include/gcast.h) on a (384, 2)
  model: the same bits as the Python engine's program of the same plan, and the oracle's step."""
  from graphcast_amd import plan
  model, graphs, params, steps, c_in = build(384, 2, "f16x3")
  x = np.random.default_rng(8).standard_normal((graphs["n_grid"], 2, c_in)).astype(np.float32)
  xd = torch.from_numpy(x).to("cuda:0")
  y_engine = model.forward_grid_node_features(xd).clone()
  native = plan.NativePlan(model.graph_arrays(), params, num_steps=steps, c_in=c_in,
                           c_out=gc.num_output_channels(gc.TASK_13), precision="f16x3")
  y = native(xd)
  native.check_range()
  assert torch.equal(y, y_engine)
  want = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
  assert rel_rmse(y.cpu().numpy(), want) <= REL_RMSE_TOL
  native.close()
