"""Dataset utilities: derived forcings and the inputs / targets / forcings split.

Mirror of the reference's ``weathernext/utils/data_utils.py`` (same public names, arguments,
errors): what turns an ERA5 / HRES sample into the three Datasets ``GraphCast.__call__`` and
``rollout.chunked_prediction`` take.  Works on ``graphcast_amd.xarray_lite`` Datasets (and on
real xarray ones: only the shared API subset is used).  Pinned by the known-answer values of the
reference's own data_utils_test.py and by executing the reference file itself
(tests/golden/make_golden_data_utils.py -> tests/test_data_utils.py).
"""
from typing import Any, Mapping, Sequence, Tuple, Union

import numpy as np
import pandas as pd

from graphcast_amd import solar_radiation
from graphcast_amd import xarray_lite as xarray

TimedeltaLike = Any      # something convertible to pd.Timedelta
TimedeltaStr = str
TargetLeadTimes = Union[TimedeltaLike, Sequence[TimedeltaLike], slice]

_SEC_PER_HOUR = 3600
_HOUR_PER_DAY = 24
SEC_PER_DAY = _SEC_PER_HOUR * _HOUR_PER_DAY
_AVG_DAY_PER_YEAR = 365.24219
AVG_SEC_PER_YEAR = SEC_PER_DAY * _AVG_DAY_PER_YEAR

DAY_PROGRESS = "day_progress"
YEAR_PROGRESS = "year_progress"
_DERIVED_VARS = {DAY_PROGRESS, f"{DAY_PROGRESS}_sin", f"{DAY_PROGRESS}_cos",
                 YEAR_PROGRESS, f"{YEAR_PROGRESS}_sin", f"{YEAR_PROGRESS}_cos"}
TISR = "toa_incident_solar_radiation"


def get_year_progress(seconds_since_epoch: np.ndarray) -> np.ndarray:
  """Year progress in [0, 1) (reference :51-71): float64 until the final cast."""
  years_since_epoch = seconds_since_epoch / SEC_PER_DAY / np.float64(_AVG_DAY_PER_YEAR)
  return np.mod(years_since_epoch, 1.0).astype(np.float32)


def get_day_progress(seconds_since_epoch: np.ndarray, longitude: np.ndarray) -> np.ndarray:
  """[time, lon] local day progress in [0, 1) (reference :74-100)."""
  day_progress_greenwich = np.mod(seconds_since_epoch, SEC_PER_DAY) / SEC_PER_DAY
  longitude_offsets = np.deg2rad(longitude) / (2 * np.pi)
  day_progress = np.mod(day_progress_greenwich[..., np.newaxis] + longitude_offsets, 1.0)
  return day_progress.astype(np.float32)


def featurize_progress(name: str, dims: Sequence[str], progress: np.ndarray) -> Mapping[str, xarray.Variable]:
  """`progress` plus its sin / cos (reference :103-132)."""
  if len(dims) != progress.ndim:
    raise ValueError(f"Number of feature dimensions ({len(dims)}) must be equal to the"
                     f" number of data dimensions: {progress.ndim}.")
  progress_phase = progress * (2 * np.pi)
  return {name: xarray.Variable(dims, progress),
          name + "_sin": xarray.Variable(dims, np.sin(progress_phase)),
          name + "_cos": xarray.Variable(dims, np.cos(progress_phase))}


def get_seconds_since_epoch(datetime_sequence) -> np.ndarray:
  """Reference :135-139."""
  return np.asarray(datetime_sequence.data).astype("datetime64[s]").astype(np.int64)


def add_derived_vars(data) -> None:
  """Adds year / day progress features in place if missing (reference :142-181)."""
  for coord in ("datetime", "lon"):
    if coord not in data.coords:
      raise ValueError(f"'{coord}' must be in `data` coordinates.")
  seconds_since_epoch = get_seconds_since_epoch(data.coords["datetime"])
  batch_dim = ("batch",) if "batch" in data.dims else ()
  if YEAR_PROGRESS not in data.data_vars:
    year_progress = get_year_progress(seconds_since_epoch)
    data.update(featurize_progress(name=YEAR_PROGRESS, dims=batch_dim + ("time",), progress=year_progress))
  if DAY_PROGRESS not in data.data_vars:
    longitude_coord = data.coords["lon"]
    day_progress = get_day_progress(seconds_since_epoch, np.asarray(longitude_coord.data))
    data.update(featurize_progress(name=DAY_PROGRESS, dims=batch_dim + ("time",) + tuple(longitude_coord.dims),
                                   progress=day_progress))


def add_tisr_var(data) -> None:
  """Adds `toa_incident_solar_radiation` in place if missing (reference :184-212)."""
  if TISR in data.data_vars:
    return
  for coord in ("datetime", "lat", "lon"):
    if coord not in data.coords:
      raise ValueError(f"'{coord}' must be in `data` coordinates.")
  data_no_batch = data.squeeze("batch") if "batch" in data.dims else data   # batch > 1 raises
  tisr = solar_radiation.get_toa_incident_solar_radiation_for_xarray(data_no_batch, use_jit=True)
  if "batch" in data.dims:
    tisr = tisr.expand_dims("batch", axis=0)
  data.update({TISR: tisr})


def _td64(x) -> np.timedelta64:
  return np.timedelta64(pd.Timedelta(x).value, "ns")


def extract_input_target_times(dataset, input_duration: TimedeltaLike,
                               target_lead_times: TargetLeadTimes) -> Tuple[Any, Any]:
  """Inputs (a contiguous period ending at lead time 0) and targets (requested lead times), with
  the time coordinate shifted to forecast lead times (reference :215-293)."""
  target_lead_times, target_duration = _process_target_lead_times_and_get_duration(target_lead_times)
  time = np.asarray(dataset.coords["time"].data).astype("timedelta64[ns]")
  dataset = dataset.assign_coords(time=time + _td64(target_duration) - time[-1])
  targets = dataset.sel({"time": target_lead_times})
  input_duration = pd.Timedelta(input_duration)
  zero = pd.Timedelta(0)
  epsilon = pd.Timedelta(1, "ns")      # label slices include both ends: open the lower one
  inputs = dataset.sel({"time": slice(-input_duration + epsilon, zero)})
  return inputs, targets


def _process_target_lead_times_and_get_duration(target_lead_times: TargetLeadTimes):
  """(normalised lead times, latest lead time) (reference :296-319)."""
  if isinstance(target_lead_times, slice):
    if target_lead_times.start is None:
      target_lead_times = slice(pd.Timedelta(1, "ns"), target_lead_times.stop, target_lead_times.step)
    target_duration = pd.Timedelta(target_lead_times.stop)
  else:
    if not isinstance(target_lead_times, (list, tuple, set)):
      target_lead_times = [target_lead_times]
    target_lead_times = [pd.Timedelta(x) for x in target_lead_times]
    target_lead_times.sort()
    target_duration = target_lead_times[-1]
  return target_lead_times, target_duration


def extract_inputs_targets_forcings(dataset, *, input_variables: Tuple[str, ...],
                                    target_variables: Tuple[str, ...], forcing_variables: Tuple[str, ...],
                                    pressure_levels: Tuple[int, ...], input_duration: TimedeltaLike,
                                    target_lead_times: TargetLeadTimes) -> Tuple[Any, Any, Any]:
  """Inputs, targets and forcings as the predictors take them (reference :322-362)."""
  dataset = dataset.sel(level=list(pressure_levels))
  if set(forcing_variables) & _DERIVED_VARS:
    add_derived_vars(dataset)
  if set(forcing_variables) & {TISR}:
    add_tisr_var(dataset)
  dataset = dataset.drop_vars("datetime")     # needed above, breaks autoregressive rollouts
  inputs, targets = extract_input_target_times(dataset, input_duration=input_duration,
                                               target_lead_times=target_lead_times)
  if set(forcing_variables) & set(target_variables):
    raise ValueError(f"Forcing variables {forcing_variables} should not "
                     f"overlap with target variables {target_variables}.")
  inputs = inputs[list(input_variables)]
  forcings = targets[list(forcing_variables)]      # forcings share the targets' time coordinates
  targets = targets[list(target_variables)]
  return inputs, targets, forcings
