"""Reduced-precision tiers next to the fp32-grade step.

The reference's ``utils/casting.py`` (``Bfloat16Cast`` :31-110, ``bfloat16_variable_view``
:155-205) casts inputs / forcings to bfloat16, reads the fp32-stored parameters as bfloat16 and
runs the WHOLE inner predictor in bfloat16: activations between the GEMMs, LayerNorm, residual
streams and every aggregation except grid2mesh's (``graphcast.py:215``) are bfloat16 values.

  * ``Bfloat16Cast`` -- the reference's wrapper, same name, signature and role -- runs the inner
    ``GraphCast`` in its ``"bf16"`` arithmetic (include/gcast.h ``GC_PREC_BF16``,
    csrc/rowmlp_bf16.inc): bfloat16 parameters, every row tensor of the step bfloat16 in HBM, one
    bf16 MFMA per product with fp32 accumulation, values rounded to bfloat16 wherever the
    reference's program materialises an array; LayerNorm's internals and every segment-sum run
    in fp32.  That is never less accurate than the reference's run but NOT bit-identical to it
    (XLA's fusion choices and its bfloat16 scatter order cannot be reproduced offline): the tier has
    its own op-by-op oracle (``oracle/gnn.py``, ``ACTIVATIONS = "bf16"``), marked parity-unpinned.
  (Rounds 1-4 also carried a ``Bf16GemmTier`` -- only the GEMM operands rounded to bfloat16 -- that the reference does
  not have; retired in round 5.)

numpy has no bfloat16: on host datasets the inputs are rounded *to bfloat16-representable
float32 values*; torch-backed (HBM-resident) datasets are rounded through ``torch.bfloat16``.
"""
import contextlib

import numpy as np

from graphcast_amd import packing
from graphcast_amd import predictor_base
from graphcast_amd import xarray_lite as xarray

TIER = "bf16"


def _round_bf16(data):
  if xarray._is_torch(data):
    import torch
    return data.to(torch.bfloat16).to(torch.float32) if data.is_floating_point() else data
  data = np.asarray(data)
  if data.dtype.kind != "f":
    return data
  return packing.bf16_round(data.astype(np.float32))


def to_bfloat16_values(ds: xarray.Dataset) -> xarray.Dataset:
  """Every floating data variable rounded to the nearest bfloat16 (container stays float32)."""
  return xarray.Dataset._construct(
      {k: xarray.Variable(v.dims, _round_bf16(v.data)) for k, v in ds._vars.items()},
      dict(ds._coords))


@contextlib.contextmanager
def precision_view(predictor, tier):
  """Runs the innermost engine-backed predictor in arithmetic mode `tier` for the duration of the
  block (the analogue of the reference's ``bfloat16_variable_view``, casting.py:155-178)."""
  inner = predictor
  while not hasattr(inner, "set_precision") and hasattr(inner, "_predictor"):
    inner = inner._predictor
  if not hasattr(inner, "set_precision"):
    raise TypeError("the bfloat16 tiers need a predictor built on graphcast_amd.graphcast.GraphCast")
  prev = inner.set_precision(tier)
  try:
    yield
  finally:
    inner.set_precision(prev)


class _Bf16Wrapper(predictor_base.Predictor):
  """Inputs / forcings / predictions rounded to bfloat16 values (reference ``_all_inputs_to_bfloat16``
  :126-134 and the cast back to the targets' dtype :61-65), the wrapped predictor run in `_tier`."""
  _tier = TIER

  def __init__(self, predictor: predictor_base.Predictor, enabled: bool = True):
    self._enabled = enabled
    self._predictor = predictor

  @predictor_base.host_datasets_on_device
  def __call__(self, inputs, targets_template, forcings, **kwargs):
    if not self._enabled:
      return self._predictor(inputs, targets_template, forcings, **kwargs)
    with precision_view(self._predictor, self._tier):
      predictions = self._predictor(to_bfloat16_values(inputs), targets_template,
                                    to_bfloat16_values(forcings), **kwargs)
    return to_bfloat16_values(predictions)

  def loss(self, inputs, targets, forcings, **kwargs):
    if not self._enabled:
      return self._predictor.loss(inputs, targets, forcings, **kwargs)
    raise NotImplementedError("inference build: training losses are out of scope")

  def loss_and_predictions(self, inputs, targets, forcings, **kwargs):
    if not self._enabled:
      return self._predictor.loss_and_predictions(inputs, targets, forcings, **kwargs)
    raise NotImplementedError("inference build: training losses are out of scope")


class Bfloat16Cast(_Bf16Wrapper):
  """The reference's wrapper (``utils/casting.py:31-65``): the inner predictor in bfloat16
  (``GC_PREC_BF16``; see the module docstring for where this run rounds and where it keeps fp32)."""
  _tier = "bf16"
