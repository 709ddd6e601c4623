"""Autoregressive wrapper: drop-in for ``weathernext/utils/autoregressive.py:39-222``.

Turns a one-step Predictor into a multi-step one: time-independent inputs are
held aside, each step predicts one frame, and ``predictions + forcings`` of that
step are appended to a rolling window of input frames whose ``time`` coordinate
is reset to the original (relative) input times (:114-125).

The reference runs the loop as ``hk.scan`` inside one jit; here it is a plain
host loop over fused device steps -- with torch-backed datasets the window
stays in HBM and nothing but kernel launches happens per step.  Training-only
features (``noise_level``, gradient checkpointing, ``loss``) are not built.
"""
from typing import Optional

from graphcast_amd import predictor_base
from graphcast_amd import xarray_lite as xarray


class Predictor(predictor_base.Predictor):
  """Wraps a one-step Predictor to make multi-step predictions autoregressively."""

  def __init__(self, predictor: predictor_base.Predictor, noise_level: Optional[float] = None,
               gradient_checkpointing: bool = False):
    if noise_level is not None:
      raise NotImplementedError("noise_level is a training feature (reference :92-95); "
                                "this build is inference-only")
    del gradient_checkpointing        # nothing to checkpoint without a backward pass
    self._predictor = predictor

  def _get_and_validate_constant_inputs(self, inputs, targets, forcings):
    """reference :89-99."""
    drop = [k for k in inputs.keys() if k in targets.keys() or k in forcings.keys()]
    constant_inputs = inputs.drop_vars(drop)
    for name in constant_inputs.keys():
      if "time" in constant_inputs[name].dims:
        raise ValueError(
            f"Time-dependent input variable {name} must either be a forcing "
            "variable, or a target variable to allow for auto-regressive feedback.")
    return constant_inputs

  def _validate_targets_and_forcings(self, targets, forcings):
    """reference :101-113."""
    for name in targets.keys():
      if "time" not in targets[name].dims:
        raise ValueError(f"Target variable {name} must be time-dependent.")
    for name in forcings.keys():
      if "time" not in forcings[name].dims:
        raise ValueError(f"Forcing variable {name} must be time-dependent.")
    overlap = set(forcings.keys()) & set(targets.keys())
    if overlap:
      raise ValueError("The following were specified as both targets and "
                       f"forcings, which isn't allowed: {overlap}")

  def _update_inputs(self, inputs, next_frame):
    """reference :114-125."""
    num_inputs = inputs.sizes["time"]
    predicted_or_forced_inputs = next_frame[list(inputs.keys())]
    return (xarray.concat([inputs, predicted_or_forced_inputs], dim="time")
            .tail(time=num_inputs)
            .assign_coords(time=inputs.coords["time"].variable))

  @predictor_base.host_datasets_on_device
  def __call__(self, inputs, targets_template, forcings, **kwargs):
    constant_inputs = self._get_and_validate_constant_inputs(inputs, targets_template, forcings)
    self._validate_targets_and_forcings(targets_template, forcings)
    inputs = inputs.drop_vars(list(constant_inputs.keys()))
    one_step_template = targets_template.isel(time=slice(0, 1))
    step_time = one_step_template.coords["time"].variable
    per_step = []
    for t in range(targets_template.sizes["time"]):
      step_forcings = forcings.isel(time=slice(t, t + 1)).assign_coords(time=step_time)
      all_inputs = constant_inputs.assign(inputs)
      predictions = self._predictor(all_inputs, one_step_template, forcings=step_forcings,
                                    **kwargs)
      next_frame = predictions.assign(step_forcings)
      inputs = self._update_inputs(inputs, next_frame)
      per_step.append(predictions)
    out = xarray.concat(per_step, dim="time")
    # the reference's scan stacks the per-step predictions along a NEW LEADING axis (:210-221):
    # every predicted variable comes back as (time, batch, ...)
    out = xarray.Dataset._construct({k: v.transpose("time", ...) for k, v in out._vars.items()},
                                    out._coords)
    return out.assign_coords({k: v.variable for k, v in targets_template.coords.items()
                              if "time" in v.dims})

  def loss(self, inputs, targets, forcings, **kwargs):
    raise NotImplementedError("inference build: the multi-step training loss (reference :224-312) "
                              "is out of scope")
