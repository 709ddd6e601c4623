"""Generates tests/golden/rollout_ref.npz by executing the REFERENCE's own host-side modules

    weathernext/utils/rollout.py          (chunked_prediction, _get_next_inputs)
    weathernext/utils/normalization.py    (InputsAndResiduals, normalize / unnormalize)
    weathernext/utils/xarray_tree.py      (map_structure)
    weathernext/utils/model_utils.py      (dataset_to_stacked, stacked_to_dataset, leading axes)

UNMODIFIED, in this container (needs /root/reference; outputs are committed).  jax / chex / absl /
dask are the numpy stand-ins of tests/golden/ref_shims; ``xarray`` is graphcast_amd.xarray_lite
(xarray is not installable here) -- so what this pins is the reference's ALGORITHM (variable
order, channel order, rolling window, residual / normalisation algebra, coordinate handling)
executed by the reference's own code; the labelled-array container underneath is ours.

The one-step predictor is a fixed random linear map + tanh on the stacked channels, built on the
REFERENCE's stacking helpers.  Run:  python tests/golden/make_golden_rollout.py
"""
import dataclasses
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, REF)

from graphcast_amd import xarray_lite                        # noqa: E402

xarray_lite.ufuncs = types.ModuleType("xarray.ufuncs")
sys.modules["xarray"] = xarray_lite
sys.modules["xarray.ufuncs"] = xarray_lite.ufuncs


class _Inert(types.ModuleType):
  def __getattr__(self, name):
    if name.startswith("__"):
      raise AttributeError(name)
    sub = _Inert(f"{self.__name__}.{name}")
    setattr(self, name, sub)
    return sub

  def __call__(self, *a, **k):
    return a[0] if a and callable(a[0]) else None        # decorators (jit / vmap / pmap) pass through


for name in ("xarray_jax", "haiku", "trimesh", "tree"):
  sys.modules.setdefault(name, _Inert(name))
import jax                                                   # noqa: E402  (numpy stand-in)
if not hasattr(jax, "vmap"):
  jax.vmap = lambda f, *a, **k: f
if not hasattr(jax, "random"):
  jax.random = types.SimpleNamespace(split=lambda rng, n=2: (rng, rng))
for name, value in (("Device", object), ("Array", np.ndarray), ("pmap", lambda f, *a, **k: f),
                    ("sharding", _Inert("jax.sharding")), ("NamedSharding", object)):
  if not hasattr(jax, name):
    setattr(jax, name, value)
sys.modules["weathernext.utils.losses"] = _Inert("weathernext.utils.losses")

import typing                                                # noqa: E402
import typing_extensions                                     # noqa: E402
for n in ("Required", "NotRequired"):
  if not hasattr(typing, n):
    setattr(typing, n, getattr(typing_extensions, n))

from weathernext.utils import model_utils as ref_mu          # noqa: E402
from weathernext.utils import normalization as ref_norm      # noqa: E402
from weathernext.utils import rollout as ref_rollout         # noqa: E402

from graphcast_amd import graphcast as gc                    # noqa: E402
from graphcast_amd import synthetic                          # noqa: E402

LAT = np.arange(-90, 91, 30.0)
LON = np.arange(0, 360, 45.0)
TASK = dataclasses.replace(gc.TASK_13, pressure_levels=(500, 850, 1000))
STEPS, SEED, STATS_SEED, W_SEED = 3, 31, 100, 3


def toy_weights(c_in, c_out):
  return (np.random.default_rng(W_SEED).standard_normal((c_in, c_out)) / np.sqrt(c_in)).astype(np.float32)


def main():
  inputs, template, forcings = synthetic.make_example(TASK, LAT, LON, num_target_steps=STEPS, seed=SEED)
  mean, std, dstd = synthetic.make_stats(TASK, seed=STATS_SEED)
  state = {}

  class RefToy:
    """One-step predictor on the REFERENCE's stacking helpers (model_utils.py:645-776)."""

    def __call__(self, inputs, targets_template, forcings, **kw):
      xi = ref_mu.dataset_to_stacked(inputs)
      xf = ref_mu.dataset_to_stacked(forcings)
      x = xarray_lite.concat([xi, xf], dim="channels")
      grid = ref_mu.lat_lon_to_leading_axes(x)
      data = np.asarray(grid.data, np.float32)
      if "a" not in state:
        c_out = ref_mu.dataset_to_stacked(targets_template).sizes["channels"]
        state["a"] = toy_weights(data.shape[-1], c_out)
        state["first_stacked_input"] = data.copy()
      y = np.tanh(data @ state["a"])
      out = xarray_lite.DataArray(y, dims=("lat", "lon", "batch", "channels"))
      return ref_mu.stacked_to_dataset(ref_mu.restore_leading_axes(out).variable, targets_template)

  wrapped = ref_norm.InputsAndResiduals(RefToy(), stddev_by_level=std, mean_by_level=mean,
                                        diffs_stddev_by_level=dstd)
  preds = ref_rollout.chunked_prediction(
      lambda rng, inputs, targets_template, forcings: wrapped(inputs, targets_template, forcings),
      rng=np.array([0, 1], dtype=np.uint32), inputs=inputs, targets_template=template, forcings=forcings,
      num_steps_per_chunk=1)
  out = {f"pred:{k}": np.asarray(preds[k].values) for k in preds.keys()}
  out.update({f"dims:{k}": np.array("|".join(preds[k].dims)) for k in preds.keys()})
  out["time"] = np.asarray(preds.coords["time"].values).astype("timedelta64[ns]").astype(np.int64)
  out["first_stacked_input"] = state["first_stacked_input"]
  out["config"] = np.array([STEPS, SEED, STATS_SEED, W_SEED])

  # ---- ensemble rollouts: chunked_prediction_generator_multiple_runs (rollout.py:158-307), the
  # un-pmapped branch, with a `sample` axis on the inputs (3 members = 3 perturbed initial states)
  n_members = 3
  rng = np.random.default_rng(77)
  members = []
  for m in range(n_members):
    pert = {k: inputs[k] * np.float32(1.0 + 0.05 * m) for k in inputs.keys()}
    members.append(xarray_lite.Dataset(pert, coords=dict(inputs.coords)))
  ens_inputs = xarray_lite.concat(members, dim="sample")
  del rng
  chunks = list(ref_rollout.chunked_prediction_generator_multiple_runs(
      lambda rng, inputs, targets_template, forcings: wrapped(inputs, targets_template, forcings),
      rngs=np.arange(2 * n_members, dtype=np.uint32).reshape(n_members, 2), inputs=ens_inputs,
      targets_template=template, forcings=forcings, num_samples=None, num_steps_per_chunk=1))
  out["ens_n_chunks"] = np.array(len(chunks))
  out["ens_sample_of_chunk"] = np.array([int(np.asarray(c.coords["sample"].values)) for c in chunks])
  out["ens_time_of_chunk"] = np.array([np.asarray(c.coords["time"].values).astype("timedelta64[ns]").astype(np.int64)[0]
                                       for c in chunks])
  key = "2m_temperature"
  out["ens_2m_temperature"] = np.stack([np.asarray(c[key].values) for c in chunks])
  out["ens_dims"] = np.array("|".join(chunks[0][key].dims))
  path = os.path.join(HERE, "rollout_ref.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith("pred:")})


if __name__ == "__main__":
  main()
