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


def _map(fn, dataset_or_array):
  if isinstance(dataset_or_array, xarray.Dataset):
    return xarray.Dataset({k: fn(dataset_or_array[k]) for k in dataset_or_array.keys()},
                          coords=dict(dataset_or_array._coords))
  return fn(dataset_or_array)


def _stat(stats, array):
  s = stats[array.name]
  return s.astype(array.dtype) if hasattr(s, "astype") else s


def normalize(values, scales, locations: Optional[xarray.Dataset]):
  """(x - location) / scale per named variable; variables without statistics pass through
  with a warning (reference :29-48)."""
  def one(array):
    if array.name is None:
      raise ValueError("Can't look up normalization constants because array has no name.")
    if locations is not None:
      if array.name in locations:
        array = array - _stat(locations, array)
      else:
        log.warning("No normalization location found for %s", array.name)
    if array.name in scales:
      array = array / _stat(scales, array)
    else:
      log.warning("No normalization scale found for %s", array.name)
    return array
  return _map(one, values)


def unnormalize(values, scales, locations: Optional[xarray.Dataset]):
  """x * scale + location (reference :51-69)."""
  def one(array):
    if array.name is None:
      raise ValueError("Can't look up normalization constants because array has no name.")
    if array.name in scales:
      array = array * _stat(scales, array)
    else:
      log.warning("No normalization scale found for %s", array.name)
    if locations is not None:
      if array.name in locations:
        array = array + _stat(locations, array)
      else:
        log.warning("No normalization location found for %s", array.name)
    return array
  return _map(one, values)


class InputsAndResiduals(predictor_base.Predictor):
  """Residual connection + input / residual normalisation around a one-step predictor
  (reference :72-160)."""

  def __init__(self, predictor: predictor_base.Predictor, stddev_by_level: xarray.Dataset,
               mean_by_level: xarray.Dataset, diffs_stddev_by_level: xarray.Dataset):
    self._predictor = predictor
    self._scales = stddev_by_level
    self._locations = mean_by_level
    self._residual_scales = diffs_stddev_by_level
    self._residual_locations = None

  def _unnormalize_prediction_and_add_input(self, inputs, norm_prediction):
    if norm_prediction.sizes.get("time") != 1:
      raise ValueError(
          "normalization.InputsAndResiduals only supports predicting a single timestep.")
    if norm_prediction.name in inputs:
      prediction = unnormalize(norm_prediction, self._residual_scales, self._residual_locations)
      last_input = inputs[norm_prediction.name].isel(time=-1)
      # `prediction` keeps its length-1 time axis; the last input frame broadcasts over it
      return prediction + last_input
    return unnormalize(norm_prediction, self._scales, self._locations)

  def _subtract_input_and_normalize_target(self, inputs, target):
    if target.sizes.get("time") != 1:
      raise ValueError(
          "normalization.InputsAndResiduals only supports wrapping predictors"
          "that predict a single timestep.")
    if target.name in inputs:
      last_input = inputs[target.name].isel(time=-1)
      return normalize(target - last_input, self._residual_scales, self._residual_locations)
    return normalize(target, self._scales, self._locations)

  def __call__(self, inputs, targets_template, forcings, **kwargs):
    norm_inputs = normalize(inputs, self._scales, self._locations)
    norm_forcings = normalize(forcings, self._scales, self._locations)
    norm_predictions = self._predictor(norm_inputs, targets_template, forcings=norm_forcings,
                                       **kwargs)
    return _map(lambda pred: self._unnormalize_prediction_and_add_input(inputs, pred),
                norm_predictions)

  def loss(self, inputs, targets, forcings, **kwargs):
    """Loss of the wrapped predictor on normalised residual targets (reference :162-176)."""
    norm_inputs = normalize(inputs, self._scales, self._locations)
    norm_forcings = normalize(forcings, self._scales, self._locations)
    norm_target_residuals = _map(
        lambda t: self._subtract_input_and_normalize_target(inputs, t), targets)
    return self._predictor.loss(norm_inputs, norm_target_residuals, forcings=norm_forcings,
                                **kwargs)

  def loss_and_predictions(self, inputs, targets, forcings, **kwargs):
    """reference :178-196."""
    norm_inputs = normalize(inputs, self._scales, self._locations)
    norm_forcings = normalize(forcings, self._scales, self._locations)
    norm_target_residuals = _map(
        lambda t: self._subtract_input_and_normalize_target(inputs, t), targets)
    (loss, scalars), norm_predictions = self._predictor.loss_and_predictions(
        norm_inputs, norm_target_residuals, forcings=norm_forcings, **kwargs)
    predictions = _map(lambda pred: self._unnormalize_prediction_and_add_input(inputs, pred),
                       norm_predictions)
    return (loss, scalars), predictions
