"""Reduced-precision tier next to the fp32-grade step -- and what it is NOT.

The reference's ``utils/casting.py`` (``Bfloat16Cast`` :31-110, ``bfloat16_variable_view``
:155-205) casts inputs / forcings to bfloat16, reads the fp32-stored parameters as bfloat16 and
runs the WHOLE inner predictor in bfloat16: activations between the GEMMs, LayerNorm, residual
streams and every aggregation except grid2mesh's (``graphcast.py:215``) are bfloat16 values.

That is NOT what is built here, and nothing in this module claims the reference's bf16 numerics:

  * ``Bf16GemmTier`` runs the inner ``GraphCast`` in its ``"bf16gemm"`` arithmetic mode
    (include/gcast.h ``GC_PREC_BF16_GEMM``): only the GEMM OPERANDS -- weights and the activations
    entering a matrix product -- are rounded to bfloat16 (nearest even) and multiplied on the bf16
    matrix cores with fp32 accumulation; bias, gathers, swish, LayerNorm, residuals and all
    aggregations stay fp32.  Results sit between the reference's bf16 run and its fp32 run.  It is
    checked against an oracle that rounds the same operands (``oracle.gnn.gemm_operands("bf16")``),
    not against the fp32 tolerance of the path and not against the reference's bf16 run (which
    cannot be reproduced here: jax/haiku bf16 semantics are not available offline).
  * ``Bfloat16Cast`` keeps the reference's NAME and signature so that code written against the
    reference fails loudly instead of silently getting different numerics: with ``enabled=True``
    it raises; with ``enabled=False`` it is the reference's pass-through.

numpy has no bfloat16: on host datasets the inputs are rounded *to bfloat16-representable
float32 values*; torch-backed (HBM-resident) datasets are rounded through ``torch.bfloat16``.
"""
import contextlib

import numpy as np

from graphcast_amd import packing
from graphcast_amd import predictor_base
from graphcast_amd import xarray_lite as xarray

TIER = "bf16gemm"


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
def bf16_gemm_view(predictor):
  """Runs the innermost engine-backed predictor in the "bf16gemm" arithmetic mode for the
  duration of the block."""
  inner = predictor
  while not hasattr(inner, "set_precision") and hasattr(inner, "_predictor"):
    inner = inner._predictor
  if not hasattr(inner, "set_precision"):
    raise TypeError("Bf16GemmTier needs a predictor built on graphcast_amd.graphcast.GraphCast")
  prev = inner.set_precision(TIER)
  try:
    yield
  finally:
    inner.set_precision(prev)


class Bf16GemmTier(predictor_base.Predictor):
  """Wrapper: inputs / forcings / predictions rounded to bfloat16 values, the wrapped predictor
  run with bfloat16 GEMM operands (see the module docstring for what stays fp32)."""

  def __init__(self, predictor: predictor_base.Predictor, enabled: bool = True):
    self._enabled = enabled
    self._predictor = predictor

  def __call__(self, inputs, targets_template, forcings, **kwargs):
    if not self._enabled:
      return self._predictor(inputs, targets_template, forcings, **kwargs)
    with bf16_gemm_view(self._predictor):
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


class Bfloat16Cast(predictor_base.Predictor):
  """The reference's wrapper name (``utils/casting.py:31-65``).  Its semantics -- the whole inner
  predictor in bfloat16 -- are NOT built: ``enabled=True`` raises and points to ``Bf16GemmTier``;
  ``enabled=False`` is the reference's pass-through."""

  def __init__(self, predictor: predictor_base.Predictor, enabled: bool = True):
    if enabled:
      raise NotImplementedError(
          "Bfloat16Cast (all activations in bfloat16, reference utils/casting.py:45-65) is not built "
          "on MI355X: the default path is fp32-grade at 16-bit matrix-core speed.  For bfloat16 GEMM "
          "operands with fp32 everywhere else use casting.Bf16GemmTier -- its numerics are NOT the "
          "reference's bf16 run.")
    self._predictor = predictor

  def __call__(self, inputs, targets_template, forcings, **kwargs):
    return self._predictor(inputs, targets_template, forcings, **kwargs)

  def loss(self, inputs, targets, forcings, **kwargs):
    return self._predictor.loss(inputs, targets, forcings, **kwargs)

  def loss_and_predictions(self, inputs, targets, forcings, **kwargs):
    return self._predictor.loss_and_predictions(inputs, targets, forcings, **kwargs)
