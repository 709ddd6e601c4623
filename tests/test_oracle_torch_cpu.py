"""oracle/torch_cpu.py (what bench.py's cpu_baseline times) against the numpy oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import graphcast as ogc          # noqa: E402
from oracle import params as oparams         # noqa: E402
from oracle import torch_cpu                 # noqa: E402


@pytest.mark.parametrize("batch", [1, 2])
def test_torch_cpu_step_equals_numpy_oracle(batch):
  res, mesh_size, steps = 10.0, 2, 3
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  c_in, c_out = 20, 9
  params = oparams.init_params(c_in, c_out, 64, steps, seed=2, nontrivial=True)
  x = np.random.default_rng(batch).standard_normal((graphs["n_grid"], batch, c_in)).astype(np.float32)
  want = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
  got = torch_cpu.forward(params, graphs, x, steps)
  assert got.shape == want.shape and got.dtype == np.float32
  err = np.linalg.norm(got - want) / np.linalg.norm(want)
  assert err < 5e-6, err
