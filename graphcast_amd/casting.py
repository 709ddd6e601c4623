"""bfloat16 tier: drop-in for ``weathernext/utils/casting.py`` (``Bfloat16Cast`` :31-110,
``bfloat16_variable_view`` :155-205).

The reference casts inputs / forcings to bfloat16, reads the fp32-stored parameters as bfloat16
and runs the whole inner predictor in bfloat16 (only the grid2mesh aggregation is up-cast,
``graphcast.py:215``), then casts the predictions back to the targets' dtype.  Here the inner
``GraphCast`` runs its ``"bf16"`` arithmetic mode (include/gcast.h ``GC_PREC_BF16``): every GEMM
operand -- weights and activations alike -- is rounded to bfloat16 (nearest even) and multiplied
on the bf16 matrix cores with fp32 accumulation, while everything between the GEMMs (bias,
gathers, swish, LayerNorm, residuals, aggregation) stays fp32.  That is the GEMM-operand part of
the reference's semantics and strictly more precise elsewhere, so results sit between the
reference's bf16 run and its fp32 run; the tier is checked against an oracle that rounds the same
operands (``oracle.gnn.gemm_operands("bf16")``), NOT against the fp32 tolerance of the path.

numpy has no bfloat16: on host datasets the inputs are rounded *to bfloat16-representable
float32 values*; torch-backed (HBM-resident) datasets are rounded through ``torch.bfloat16``.
"""
import contextlib

import numpy as np

from graphcast_amd import packing
from graphcast_amd import predictor_base
from graphcast_amd import xarray_lite as xarray


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
def bfloat16_variable_view(predictor):
  """Runs the innermost engine-backed predictor in the "bf16" arithmetic mode for the duration of
  the block (the role of the reference's haiku getter context, :155-205)."""
  inner = predictor
  while not hasattr(inner, "set_precision") and hasattr(inner, "_predictor"):
    inner = inner._predictor
  if not hasattr(inner, "set_precision"):
    raise TypeError("Bfloat16Cast needs a predictor built on graphcast_amd.graphcast.GraphCast")
  prev = inner.set_precision("bf16")
  try:
    yield
  finally:
    inner.set_precision(prev)


class Bfloat16Cast(predictor_base.Predictor):
  """Wrapper that runs the wrapped predictor in the bfloat16 tier and returns the targets' dtype."""

  def __init__(self, predictor: predictor_base.Predictor, enabled: bool = True):
    self._enabled = enabled
    self._predictor = predictor

  def __call__(self, inputs, targets_template, forcings, **kwargs):
    if not self._enabled:
      return self._predictor(inputs, targets_template, forcings, **kwargs)
    with bfloat16_variable_view(self._predictor):
      predictions = self._predictor(to_bfloat16_values(inputs), targets_template,
                                    to_bfloat16_values(forcings), **kwargs)
    # the reference rounds the predictions to bfloat16 before casting them back (:63-65)
    return to_bfloat16_values(predictions)

  def loss(self, inputs, targets, forcings, **kwargs):
    if not self._enabled:
      return self._predictor.loss(inputs, targets, forcings, **kwargs)
    raise NotImplementedError("inference build: the bfloat16 training loss (reference :67-90) is out of scope")

  def loss_and_predictions(self, inputs, targets, forcings, **kwargs):
    if not self._enabled:
      return self._predictor.loss_and_predictions(inputs, targets, forcings, **kwargs)
    raise NotImplementedError("inference build (reference :92-125)")
