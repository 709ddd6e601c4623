"""Synthetic example batches shaped like the reference's demo data.

No dataset or checkpoint is reachable offline, so benchmarks and tests build
inputs / targets_template / forcings with the dims, coordinate conventions and
variable split that the reference's ``data_utils.extract_inputs_targets_forcings``
(``weathernext/utils/data_utils.py:322-362``) produces for a ``TaskConfig``:

  inputs   : time-dependent vars (batch, time=2, lat, lon[, level]), statics (lat, lon);
             ``time`` = [-6h, 0h] relative to the last input frame
  targets  : (batch, time=T, lat, lon[, level]), ``time`` = [6h, 12h, ...]
  forcings : (batch, time=T, lat, lon) for ``forcing_variables``, same lead times

plus per-variable statistics datasets (mean / stddev / diffs_stddev by level)
for the normalisation wrapper.  Values are seeded standard normals (i.e. fields
that are already O(1)); ``datetime`` is a (batch, time) coordinate.
"""
import numpy as np

from graphcast_amd import variables
from graphcast_amd import xarray_lite as xarray

_STEP = np.timedelta64(6, "h")


def _dims(name, with_time=True):
  if name in variables.STATIC_VARS:
    return ("lat", "lon")
  if name in variables.ALL_ATMOSPHERIC_VARS:
    return (("batch", "time") if with_time else ("batch",)) + ("level", "lat", "lon")
  if name in variables.TIME_FORCING_VARS:
    # the reference's time-only forcings are (batch, time); they broadcast over the grid
    return ("batch", "time")
  return (("batch", "time") if with_time else ("batch",)) + ("lat", "lon")


def make_example(task_config, lat, lon, *, batch=1, num_target_steps=1, num_input_frames=2,
                 seed=0, dtype=np.float32, t0="2022-01-01T00"):
  """-> (inputs, targets_template, forcings) for ``task_config`` on the lat/lon grid."""
  rng = np.random.default_rng(seed)
  lat = np.asarray(lat, dtype=np.float32)
  lon = np.asarray(lon, dtype=np.float32)
  levels = np.asarray(task_config.pressure_levels, dtype=np.int32)
  size = dict(batch=batch, lat=len(lat), lon=len(lon), level=len(levels))
  in_time = (np.arange(num_input_frames) - (num_input_frames - 1)) * _STEP
  tgt_time = (np.arange(num_target_steps) + 1) * _STEP
  base = np.datetime64(t0, "ns")

  def draw(name, n_time):
    dims = _dims(name)
    shape = tuple(n_time if d == "time" else size[d] for d in dims)
    return dims, rng.standard_normal(shape, dtype=np.float32).astype(dtype)

  def coords(time):
    return dict(lat=lat, lon=lon, level=levels, time=time,
                datetime=(("batch", "time"), np.broadcast_to(base + time, (batch, len(time))).copy()))

  inputs = xarray.Dataset({n: draw(n, num_input_frames) for n in task_config.input_variables},
                          coords=coords(in_time))
  forcings = xarray.Dataset({n: draw(n, num_target_steps) for n in task_config.forcing_variables},
                            coords=coords(tgt_time))

  def template(name):
    dims = _dims(name)
    shape = tuple(num_target_steps if d == "time" else size[d] for d in dims)
    return dims, np.broadcast_to(np.zeros((), dtype=dtype), shape)

  targets_template = xarray.Dataset({n: template(n) for n in task_config.target_variables},
                                    coords=coords(tgt_time))
  return inputs, targets_template, forcings


def make_stats(task_config, seed=100, dtype=np.float32):
  """(mean_by_level, stddev_by_level, diffs_stddev_by_level) with non-trivial values."""
  rng = np.random.default_rng(seed)
  levels = np.asarray(task_config.pressure_levels, dtype=np.int32)
  names = sorted(set(task_config.input_variables) | set(task_config.target_variables)
                 | set(task_config.forcing_variables))

  def stat(fn):
    out = {}
    for n in names:
      if n in variables.ALL_ATMOSPHERIC_VARS:
        out[n] = (("level",), fn(len(levels)).astype(dtype))
      else:
        out[n] = ((), np.asarray(fn(1)[0], dtype=dtype))
    return xarray.Dataset(out, coords=dict(level=levels))

  mean = stat(lambda n: 0.5 * rng.standard_normal(n))
  std = stat(lambda n: 0.5 + rng.random(n))
  diff_std = stat(lambda n: 0.1 + 0.4 * rng.random(n))
  return mean, std, diff_std


def to_device(dataset, device):
  """Dataset with every data variable moved to ``device`` as a torch tensor
  (coordinates stay numpy) -- the ``device_put_fn`` for HBM-resident rollouts."""
  return xarray.to_device(dataset, device)
