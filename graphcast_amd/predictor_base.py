"""Abstract Predictor API, same shape as the reference's
``weathernext/utils/predictor_base.py:27-169``: ``__call__(inputs,
targets_template, forcings, **kw) -> Dataset``, ``loss``,
``loss_and_predictions``.  This build is inference-only: the default ``loss``
of the reference (a zero per-batch loss) is kept, trainable behaviour is not.
"""
import abc
import os
from typing import Any, Mapping, Tuple

import numpy as np

LossAndDiagnostics = Tuple[Any, Mapping[str, Any]]


def device_of(predictor):
  """The device of the innermost predictor of a wrapper chain (``GraphCast._device``), or None."""
  for _ in range(16):
    if predictor is None:
      return None
    device = getattr(predictor, "_device", None)
    if device is not None:
      return device
    predictor = getattr(predictor, "_predictor", None)
  return None


def host_datasets_on_device(call):
  """Decorator for a wrapper's ``__call__(self, inputs, targets_template, forcings, **kw)``: called on HOST
  (numpy-backed) Datasets around a predictor that lives on a GPU, the wrapper's own Dataset arithmetic --
  normalisation, bfloat16 rounding, the autoregressive feedback -- would run in numpy on the host, seconds per
  0.25 deg step.  Instead the OUTERMOST wrapper uploads inputs and forcings once (per variable, xarray_lite
  .to_device), the whole chain runs on device-resident Datasets (wrappers further in see torch-backed data and pass
  through), and the predictions come back as host Datasets -- what the reference does by construction: its
  Datasets hold jax device arrays (xarray_jax) and ``rollout`` ends with ``jax.device_get`` (rollout.py:362).
  Device-resident inputs, a CPU predictor, or a chain without a device: the call is passed through untouched."""
  import functools

  @functools.wraps(call)
  def wrapped(self, inputs, targets_template, forcings=None, **kwargs):
    from graphcast_amd import xarray_lite as xl
    device = device_of(self)
    inputs, targets_template, forcings = xl.from_xarray(inputs), xl.from_xarray(targets_template), xl.from_xarray(forcings)
    if (device is None or not str(device).startswith("cuda") or not xl.is_host(inputs)
        or os.environ.get("GCAST_WRAPPERS_ON_HOST") == "1"):          # (the switch: A/B runs of scripts/host_boundary_bench.py)
      return call(self, inputs, targets_template, forcings, **kwargs)
    out = call(self, xl.to_device(inputs, device), targets_template,
               None if forcings is None else xl.to_device(forcings, device), **kwargs)
    return xl.to_host(out) if isinstance(out, xl.Dataset) else out
  return wrapped


class Predictor(abc.ABC):
  """A predictor of weather exposing a Dataset-based API."""

  @abc.abstractmethod
  def __call__(self, inputs, targets_template, forcings, **optional_kwargs):
    """Returns predictions shaped like ``targets_template``."""

  def loss(self, inputs, targets, forcings, **optional_kwargs) -> LossAndDiagnostics:
    """Default of the reference (:131-135): a zero loss per batch element, no diagnostics."""
    del targets, forcings, optional_kwargs
    from graphcast_amd import xarray_lite as xl
    return xl.DataArray(np.zeros(inputs.sizes["batch"]), dims=("batch",)), {}

  def loss_and_predictions(self, inputs, targets, forcings, **optional_kwargs):
    """Reference :137-169: not implemented in the base class."""
    raise NotImplementedError
