"""Abstract Predictor API, same shape as the reference's
``weathernext/utils/predictor_base.py:27-169``: ``__call__(inputs,
targets_template, forcings, **kw) -> Dataset``, ``loss``,
``loss_and_predictions``.  This build is inference-only: the default ``loss``
of the reference (a zero per-batch loss) is kept, trainable behaviour is not.
"""
import abc
from typing import Any, Mapping, Tuple

import numpy as np

LossAndDiagnostics = Tuple[Any, Mapping[str, Any]]


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
