"""From a raw (ERA5-like) sample to the three Datasets the Predictor takes.

Host-side data ingestion for ``GraphCast.__call__`` / ``rollout.chunked_prediction`` (SURVEY.md 8
f3).  Public names and call signatures follow ``weathernext/utils/data_utils.py`` (the demo
notebook calls ``extract_inputs_targets_forcings(batch, target_lead_times=..., **asdict(task))``);
the implementation is written from the behaviour pinned by the reference's known-answer tests and
by ``tests/golden/data_utils_ref.npz`` (outputs of the reference module executed here):

  * clock features: fraction of the tropical year / of the local solar day elapsed, and their
    sine / cosine, derived from the ``datetime`` coordinate (reference :51-181);
  * the lead-time split: the sample's time axis is re-labelled so that the LAST input frame sits
    at lead time 0; inputs are the frames in (-input_duration, 0], targets / forcings the frames
    at the requested lead times (reference :215-362).

  * ``toa_incident_solar_radiation``, when the sample lacks it, is derived from solar geometry by
    ``graphcast_amd.solar_radiation`` (the reference's float32 arithmetic, pinned to the reference
    module's own output) exactly where the reference derives it (reference :184-212).
"""
from typing import Any, Dict, Sequence, Tuple

import numpy as np
import pandas as pd

from graphcast_amd import xarray_lite as xarray

SEC_PER_DAY = 86400
AVG_SEC_PER_YEAR = SEC_PER_DAY * 365.24219          # mean tropical year

DAY_PROGRESS = "day_progress"
YEAR_PROGRESS = "year_progress"
TISR = "toa_incident_solar_radiation"
_DERIVED_VARS = frozenset(f"{stem}{suffix}" for stem in (DAY_PROGRESS, YEAR_PROGRESS)
                          for suffix in ("", "_sin", "_cos"))
_NS = np.timedelta64(1, "ns")


# ---------------------------------------------------------------------------- clock features
def get_year_progress(seconds_since_epoch: np.ndarray) -> np.ndarray:
  """Fraction of the mean tropical year elapsed, in [0, 1), float32 (float64 inside: the day
  count of a modern timestamp does not fit float32)."""
  days = np.asarray(seconds_since_epoch, dtype=np.float64) / SEC_PER_DAY
  return np.mod(days / 365.24219, 1.0).astype(np.float32)


def get_day_progress(seconds_since_epoch: np.ndarray, longitude: np.ndarray) -> np.ndarray:
  """[..., lon] fraction of the LOCAL solar day elapsed, in [0, 1), float32: the Greenwich
  fraction advanced by longitude / 360 deg."""
  utc = np.mod(np.asarray(seconds_since_epoch), SEC_PER_DAY) / SEC_PER_DAY
  shift = np.deg2rad(np.asarray(longitude)) / (2.0 * np.pi)
  return np.mod(utc[..., None] + shift, 1.0).astype(np.float32)


def featurize_progress(name: str, dims: Sequence[str], progress: np.ndarray) -> Dict[str, xarray.Variable]:
  """{name, name_sin, name_cos}: the progress and the two coordinates of its phase angle."""
  if len(dims) != np.ndim(progress):
    raise ValueError(f"Number of feature dimensions ({len(dims)}) must be equal to the"
                     f" number of data dimensions: {np.ndim(progress)}.")
  angle = 2.0 * np.pi * progress
  columns = {"": progress, "_sin": np.sin(angle), "_cos": np.cos(angle)}
  return {name + suffix: xarray.Variable(dims, values) for suffix, values in columns.items()}


def get_seconds_since_epoch(datetime_sequence) -> np.ndarray:
  return np.asarray(datetime_sequence.data).astype("datetime64[s]").astype(np.int64)


def _require_coords(data, names):
  for name in names:
    if name not in data.coords:
      raise ValueError(f"'{name}' must be in `data` coordinates.")


def add_derived_vars(data) -> None:
  """Adds (in place) whichever of the year / day progress feature triples the dataset lacks."""
  _require_coords(data, ("datetime", "lon"))
  seconds = get_seconds_since_epoch(data.coords["datetime"])
  lead = (("batch",) if "batch" in data.dims else ()) + ("time",)
  if YEAR_PROGRESS not in data.data_vars:
    data.update(featurize_progress(YEAR_PROGRESS, lead, get_year_progress(seconds)))
  if DAY_PROGRESS not in data.data_vars:
    lon = data.coords["lon"]
    data.update(featurize_progress(DAY_PROGRESS, lead + tuple(lon.dims),
                                   get_day_progress(seconds, np.asarray(lon.data))))


def add_tisr_var(data) -> None:
  """Adds ``toa_incident_solar_radiation`` in place when the sample lacks it, derived from the
  ``datetime`` / ``lat`` / ``lon`` coordinates (reference :184-212).  A batch axis must have length
  1 (the derivation is per timestamp): longer ones fail in ``squeeze`` like the reference's."""
  if TISR in data.data_vars:
    return
  _require_coords(data, ("datetime", "lat", "lon"))
  from graphcast_amd import solar_radiation
  batched = "batch" in data.dims
  tisr = solar_radiation.get_toa_incident_solar_radiation_for_xarray(
      data.squeeze("batch") if batched else data, use_jit=True)
  data.update({TISR: tisr.expand_dims("batch", axis=0) if batched else tisr})


# ---------------------------------------------------------------------------- lead-time split
def _lead_times(spec) -> Tuple[Any, pd.Timedelta]:
  """Normalises a lead-time request -> (selector for Dataset.sel, latest lead time).
  A slice selects by label range (an open start means "everything after lead time 0"); a single
  value or a collection selects exactly those lead times, in ascending order."""
  if isinstance(spec, slice):
    start = pd.Timedelta(1, "ns") if spec.start is None else spec.start
    return slice(start, spec.stop, spec.step), pd.Timedelta(spec.stop)
  wanted = sorted(pd.Timedelta(t) for t in (spec if isinstance(spec, (list, tuple, set)) else [spec]))
  return wanted, wanted[-1]


def extract_input_target_times(dataset, input_duration, target_lead_times):
  """(inputs, targets): `dataset` re-labelled in lead time -- its last frame is given the latest
  requested lead time, so lead time 0 is the newest input frame -- then cut into the input window
  (-input_duration, 0] and the requested target lead times."""
  selector, horizon = _lead_times(target_lead_times)
  stamps = np.asarray(dataset.coords["time"].data).astype("timedelta64[ns]")
  shifted = dataset.assign_coords(time=stamps - stamps[-1] + np.timedelta64(horizon.value, "ns"))
  window_start = pd.Timedelta(1, "ns") - pd.Timedelta(input_duration)     # label slices are closed: open it
  inputs = shifted.sel({"time": slice(window_start, pd.Timedelta(0))})
  targets = shifted.sel({"time": selector})
  return inputs, targets


def extract_inputs_targets_forcings(dataset, *, input_variables: Tuple[str, ...],
                                    target_variables: Tuple[str, ...],
                                    forcing_variables: Tuple[str, ...],
                                    pressure_levels: Tuple[int, ...], input_duration,
                                    target_lead_times):
  """(inputs, targets, forcings) for a task: the task's pressure levels, derived clock features
  where the task forces with them, the lead-time split, and the per-role variable subsets
  (forcings are read at the TARGET times).  The ``datetime`` coordinate is dropped: it would
  change from chunk to chunk in an autoregressive rollout."""
  clash = set(forcing_variables) & set(target_variables)
  dataset = dataset.sel(level=list(pressure_levels))
  forced = set(forcing_variables)
  if forced & _DERIVED_VARS:
    add_derived_vars(dataset)
  if TISR in forced:
    add_tisr_var(dataset)
  inputs, at_targets = extract_input_target_times(dataset.drop_vars("datetime"),
                                                  input_duration=input_duration,
                                                  target_lead_times=target_lead_times)
  if clash:
    raise ValueError(f"Forcing variables {forcing_variables} should not "
                     f"overlap with target variables {target_variables}.")
  pick = lambda ds, names: ds[list(names)]
  return pick(inputs, input_variables), pick(at_targets, target_variables), pick(at_targets, forcing_variables)
