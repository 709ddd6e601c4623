"""Normalisation wrappers: drop-in for ``weathernext/utils/normalization.py``
(``normalize`` :29-48, ``unnormalize`` :51-69, ``InputsAndResiduals`` :72-196).

The inner predictor sees inputs / forcings normalised with per-variable (and
per-level) ``mean`` / ``stddev``; for a target that is also an input it predicts
the *residual* to the last input frame in units of ``diffs_stddev``; otherwise the
target itself in units of ``stddev`` around ``mean``.

Works on ``xarray_lite`` containers with numpy or torch (HBM-resident) data: the
by-name broadcasting arithmetic of ``DataArray`` moves the small statistics
vectors to the data's device.  ``DeviceRollout`` (rollout_device.py) folds the
same algebra into one state-advance kernel for long rollouts.
"""
import logging
from typing import Optional

from graphcast_amd import predictor_base
from graphcast_amd import xarray_lite as xarray

log = logging.getLogger(__name__)


def _each_variable(values, fn):
  """fn over the DataArrays of a Dataset (coordinates kept), or over a single DataArray."""
  if not isinstance(values, xarray.Dataset):
    return fn(values)
  return xarray.Dataset({name: fn(values[name]) for name in values.keys()}, coords=dict(values._coords))


def _statistic(stats, array, kind):
  """The named variable's entry of `stats` in the array's dtype; None (with the reference's warning)
  if there is none."""
  if stats is None:
    return None
  if array.name not in stats:
    log.warning("No normalization %s found for %s", kind, array.name)
    return None
  value = stats[array.name]
  return value.astype(array.dtype) if hasattr(value, "astype") else value


def _affine(values, scales, locations, inverse):
  def one(array):
    if array.name is None:
      raise ValueError("Can't look up normalization constants because array has no name.")
    if inverse:                                    # x * scale + location
      scale = _statistic(scales, array, "scale")
      array = array if scale is None else array * scale
      location = _statistic(locations, array, "location")
      return array if location is None else array + location
    location = _statistic(locations, array, "location")     # (x - location) / scale
    array = array if location is None else array - location
    scale = _statistic(scales, array, "scale")
    return array if scale is None else array / scale
  return _each_variable(values, one)


def normalize(values, scales, locations: Optional[xarray.Dataset]):
  """(x - location) / scale per named variable; a variable without a statistic passes through that
  step with a warning (reference :29-48)."""
  return _affine(values, scales, locations, inverse=False)


def unnormalize(values, scales, locations: Optional[xarray.Dataset]):
  """x * scale + location (reference :51-69)."""
  return _affine(values, scales, locations, inverse=True)


class InputsAndResiduals(predictor_base.Predictor):
  """Residual connection + input / residual normalisation around a one-step predictor
  (reference :72-160): a target that is also an input is predicted as the difference to the last
  input frame in units of `diffs_stddev`, any other target in units of `stddev` around `mean`."""

  def __init__(self, predictor: predictor_base.Predictor, stddev_by_level: xarray.Dataset,
               mean_by_level: xarray.Dataset, diffs_stddev_by_level: xarray.Dataset):
    self._predictor = predictor
    stddev_by_level, mean_by_level, diffs_stddev_by_level = (
        xarray.from_xarray(stddev_by_level), xarray.from_xarray(mean_by_level), xarray.from_xarray(diffs_stddev_by_level))
    self._state_stats = (stddev_by_level, mean_by_level)          # (scales, locations)
    self._residual_stats = (diffs_stddev_by_level, None)

  @staticmethod
  def _single_step(array, message):
    if array.sizes.get("time") != 1:
      raise ValueError(message)

  def _physical(self, inputs, norm_prediction):
    """One predicted variable back in physical units (the residual ones added to the last input
    frame, which broadcasts over the prediction's length-1 time axis)."""
    self._single_step(norm_prediction,
                      "normalization.InputsAndResiduals only supports predicting a single timestep.")
    if norm_prediction.name not in inputs:
      return unnormalize(norm_prediction, *self._state_stats)
    return unnormalize(norm_prediction, *self._residual_stats) + inputs[norm_prediction.name].isel(time=-1)

  def _normalised_target(self, inputs, target):
    self._single_step(target, "normalization.InputsAndResiduals only supports wrapping predictors"
                              "that predict a single timestep.")
    if target.name not in inputs:
      return normalize(target, *self._state_stats)
    return normalize(target - inputs[target.name].isel(time=-1), *self._residual_stats)

  def _normalised_io(self, inputs, forcings):
    return normalize(inputs, *self._state_stats), normalize(forcings, *self._state_stats)

  @predictor_base.host_datasets_on_device
  def __call__(self, inputs, targets_template, forcings, **kwargs):
    inputs, targets_template, forcings = (xarray.from_xarray(inputs), xarray.from_xarray(targets_template),
                                          xarray.from_xarray(forcings))
    norm_inputs, norm_forcings = self._normalised_io(inputs, forcings)
    norm_predictions = self._predictor(norm_inputs, targets_template, forcings=norm_forcings, **kwargs)
    return _each_variable(norm_predictions, lambda p: self._physical(inputs, p))

  def loss(self, inputs, targets, forcings, **kwargs):
    """Loss of the wrapped predictor on normalised residual targets (reference :162-176)."""
    norm_inputs, norm_forcings = self._normalised_io(inputs, forcings)
    norm_targets = _each_variable(targets, lambda t: self._normalised_target(inputs, t))
    return self._predictor.loss(norm_inputs, norm_targets, forcings=norm_forcings, **kwargs)

  def loss_and_predictions(self, inputs, targets, forcings, **kwargs):
    """reference :178-196."""
    norm_inputs, norm_forcings = self._normalised_io(inputs, forcings)
    norm_targets = _each_variable(targets, lambda t: self._normalised_target(inputs, t))
    (loss, scalars), norm_predictions = self._predictor.loss_and_predictions(
        norm_inputs, norm_targets, forcings=norm_forcings, **kwargs)
    return (loss, scalars), _each_variable(norm_predictions, lambda p: self._physical(inputs, p))
