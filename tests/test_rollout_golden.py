"""The host-side plumbing against the REFERENCE'S OWN CODE: tests/golden/rollout_ref.npz was
produced by executing weathernext/utils/rollout.py (chunked_prediction), normalization.py
(InputsAndResiduals), xarray_tree.py and model_utils.py (stacking helpers) unmodified
(tests/golden/make_golden_rollout.py) around a fixed toy one-step predictor.  graphcast_amd's
rollout / normalization / model_utils must reproduce those arrays: variable order, channel
order, rolling window, residual + normalisation algebra, time coordinates."""
import dataclasses
import os

import numpy as np
import pytest

from graphcast_amd import graphcast as gc
from graphcast_amd import model_utils
from graphcast_amd import normalization
from graphcast_amd import predictor_base
from graphcast_amd import rollout
from graphcast_amd import synthetic
from graphcast_amd import xarray_lite as xarray

LAT = np.arange(-90, 91, 30.0)
LON = np.arange(0, 360, 45.0)
TASK = dataclasses.replace(gc.TASK_13, pressure_levels=(500, 850, 1000))


@pytest.fixture(scope="module")
def ref(golden_dir):
  return np.load(os.path.join(golden_dir, "rollout_ref.npz"))


class Toy(predictor_base.Predictor):
  """Same fixed linear map + tanh as make_golden_rollout.RefToy, on OUR stacking helpers."""

  def __init__(self, w_seed):
    self.w_seed, self.a, self.first = w_seed, None, None

  def __call__(self, inputs, targets_template, forcings, **kw):
    x = xarray.concat([model_utils.dataset_to_stacked(inputs),
                       model_utils.dataset_to_stacked(forcings)], dim="channels")
    data = np.asarray(model_utils.lat_lon_to_leading_axes(x).data, np.float32)
    if self.a is None:
      c_out = model_utils.dataset_to_stacked(targets_template).sizes["channels"]
      self.a = (np.random.default_rng(self.w_seed).standard_normal((data.shape[-1], c_out))
                / np.sqrt(data.shape[-1])).astype(np.float32)
      self.first = data.copy()
    y = xarray.DataArray(np.tanh(data @ self.a), dims=("lat", "lon", "batch", "channels"))
    return model_utils.stacked_to_dataset(model_utils.restore_leading_axes(y).variable, targets_template)


def test_rollout_normalisation_stacking_match_reference_execution(ref):
  steps, seed, stats_seed, w_seed = (int(v) for v in ref["config"])
  inputs, template, forcings = synthetic.make_example(TASK, LAT, LON, num_target_steps=steps, seed=seed)
  mean, std, dstd = synthetic.make_stats(TASK, seed=stats_seed)
  toy = Toy(w_seed)
  wrapped = normalization.InputsAndResiduals(toy, std, mean, dstd)
  preds = rollout.chunked_prediction(lambda rng, **kw: wrapped(**kw), None, inputs, template, forcings)
  # the stacked, normalised input of the very first step: channel order + normalisation
  np.testing.assert_array_equal(toy.first, ref["first_stacked_input"])
  names = sorted(k[5:] for k in ref.files if k.startswith("pred:"))
  assert names == sorted(preds.keys())
  for k in names:
    assert "|".join(preds[k].dims) == str(ref[f"dims:{k}"])
    np.testing.assert_allclose(preds[k].values, ref[f"pred:{k}"], rtol=1e-6, atol=1e-6, err_msg=k)
  got_time = np.asarray(preds.coords["time"].values).astype("timedelta64[ns]").astype(np.int64)
  np.testing.assert_array_equal(got_time, ref["time"])


def test_autoregressive_predictor_matches_reference_execution(golden_dir):
  """tests/golden/autoregressive_ref.npz = the reference's autoregressive.Predictor.__call__
  (utils/autoregressive.py:127-222, hk.scan as a python loop) around the same toy step."""
  from graphcast_amd import autoregressive
  z = np.load(os.path.join(golden_dir, "autoregressive_ref.npz"))
  steps, seed, w_seed = (int(v) for v in z["config"])
  inputs, template, forcings = synthetic.make_example(TASK, LAT, LON, num_target_steps=steps, seed=seed)
  strip = lambda ds: ds.drop_vars(["datetime"])
  preds = autoregressive.Predictor(Toy(w_seed))(strip(inputs), strip(template), strip(forcings))
  for k in sorted(k[5:] for k in z.files if k.startswith("pred:")):
    assert "|".join(preds[k].dims) == str(z[f"dims:{k}"]), k       # time-leading, like hk.scan's stacking
    np.testing.assert_allclose(preds[k].values, z[f"pred:{k}"], rtol=1e-6, atol=1e-6, err_msg=k)


def test_ensemble_generator_matches_reference_execution(ref):
  """chunked_prediction_generator_multiple_runs (reference rollout.py:158-307, un-pmapped branch)
  executed by the reference on an inputs Dataset with a `sample` axis: chunk order (all lead times
  of a member before the next member), the scalar `sample` coordinate of every chunk, and the
  values -- and the same members when the work is sharded over two ranks."""
  steps, seed, stats_seed, w_seed = (int(v) for v in ref["config"])
  inputs, template, forcings = synthetic.make_example(TASK, LAT, LON, num_target_steps=steps, seed=seed)
  mean, std, dstd = synthetic.make_stats(TASK, seed=stats_seed)
  wrapped = normalization.InputsAndResiduals(Toy(w_seed), std, mean, dstd)
  n_members = 3
  members = [xarray.Dataset({k: inputs[k] * np.float32(1.0 + 0.05 * m) for k in inputs.keys()},
                            coords=dict(inputs.coords)) for m in range(n_members)]
  ens_inputs = xarray.concat(members, dim="sample")
  fn = lambda rng, **kw: wrapped(**kw)
  rngs = np.arange(2 * n_members, dtype=np.uint32).reshape(n_members, 2)
  chunks = list(rollout.chunked_prediction_generator_multiple_runs(
      fn, rngs, ens_inputs, template, forcings, num_samples=None, num_steps_per_chunk=1))
  assert len(chunks) == int(ref["ens_n_chunks"]) == n_members * steps
  np.testing.assert_array_equal([int(np.asarray(c.coords["sample"].values)) for c in chunks], ref["ens_sample_of_chunk"])
  np.testing.assert_array_equal(
      [np.asarray(c.coords["time"].values).astype("timedelta64[ns]").astype(np.int64)[0] for c in chunks],
      ref["ens_time_of_chunk"])
  assert "|".join(chunks[0]["2m_temperature"].dims) == str(ref["ens_dims"])
  got = np.stack([np.asarray(c["2m_temperature"].values) for c in chunks])
  np.testing.assert_allclose(got, ref["ens_2m_temperature"], rtol=0, atol=2e-6)
  # one process per GPU: rank r of 2 rolls out members r, r + 2, ... -- the union is the same set
  by_rank = [list(rollout.chunked_prediction_generator_multiple_runs(
      fn, rngs, ens_inputs, template, forcings, num_samples=None, num_steps_per_chunk=1, rank=r, world_size=2))
      for r in (0, 1)]
  assert [int(np.asarray(c.coords["sample"].values)) for c in by_rank[0]] == [0] * steps + [2] * steps
  assert [int(np.asarray(c.coords["sample"].values)) for c in by_rank[1]] == [1] * steps
  for c in by_rank[1]:
    i = steps + [int(np.asarray(x.coords["time"].values).astype("timedelta64[ns]").astype(np.int64)[0]) for x in by_rank[1]].index(
        int(np.asarray(c.coords["time"].values).astype("timedelta64[ns]").astype(np.int64)[0]))
    np.testing.assert_allclose(np.asarray(c["2m_temperature"].values), ref["ens_2m_temperature"][i], rtol=0, atol=2e-6)
