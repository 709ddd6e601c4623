"""Chunked autoregressive rollouts: drop-in for ``weathernext/utils/rollout.py``.

Same public names, arguments and error behaviour as the reference
(``chunked_prediction`` :326-364, ``chunked_prediction_generator`` :367-565,
``chunked_prediction_generator_multiple_runs`` :158-307, ``_get_next_inputs``
:581-604, ``extend_targets_template`` :618-658): the predictor is called **by
keyword** with ``rng, inputs, targets_template, forcings``; every chunk sees the
time coordinates of the first chunk (relative lead times); ``datetime``
coordinates are stripped before the loop and re-attached to the predictions.

What differs, MI355X-first:

* There is no jit / pmap.  The rolling-window update (``_get_next_inputs``) is a
  handful of container operations on ``xarray_lite`` objects whose ``data`` may
  be torch tensors resident in HBM -- the 2-frame state never has to leave the
  device between steps (``device_put_fn`` moves it there once).
* ``pmap_devices`` (one ensemble member per local device driven from one
  process) becomes *one process per GPU*: ``chunked_prediction_generator_multiple_runs``
  takes ``rank`` / ``world_size`` and rolls out only the members this rank owns
  (member ``i`` -> rank ``i % world_size``, see ``ensemble.py``); there is no
  collective inside a step.  Passing ``pmap_devices`` raises with that message.
* ``rng`` is opaque to this module (GraphCast is deterministic): it is split with
  ``split_rng`` -- numpy ``SeedSequence`` spawning for ints / SeedSequences, pass
  through for ``None`` -- and handed to the predictor unchanged otherwise.
"""
import logging
from typing import Any, Callable, Iterator, Optional, Protocol, Sequence

import numpy as np

from graphcast_amd import xarray_lite as xarray

log = logging.getLogger(__name__)


class PredictorFn(Protocol):
  """Functional version of ``Predictor.__call__`` with an explicit rng (reference :78-87)."""

  def __call__(self, rng: Any, inputs: xarray.Dataset, targets_template: xarray.Dataset,
               forcings: Optional[xarray.Dataset], **optional_kwargs) -> xarray.Dataset:
    ...


# ----------------------------------------------------------------------------- rng
def split_rng(rng, split_fn: Optional[Callable[[Any], tuple]] = None):
  """(carry, key for this chunk) -- the role of ``_split_rng_fn`` (reference :568-578).

  ``None`` stays ``None`` (deterministic predictors); ints, integer key arrays and
  ``SeedSequence`` spawn two child SeedSequences; a ``numpy.random.Generator`` spawns two child
  generators; any other key type needs the caller's
  own ``split_fn(key) -> (carry, this)`` (e.g. ``lambda k: tuple(jax.random.split(k))``) -- handing
  the SAME opaque key to every chunk would correlate the noise across lead times."""
  if rng is None:
    return None, None
  if split_fn is not None:
    carry, this = split_fn(rng)
    return carry, this
  if isinstance(rng, (int, np.integer)):
    rng = np.random.SeedSequence(int(rng))
  elif isinstance(rng, np.ndarray) and rng.dtype.kind in "ui":
    # a raw key array (the layout of a jax PRNGKey): its words seed a SeedSequence
    rng = np.random.SeedSequence([int(w) for w in rng.reshape(-1)])
  if isinstance(rng, (np.random.SeedSequence, np.random.Generator)):
    carry, this = rng.spawn(2)
    return carry, this
  raise TypeError(f"cannot split an rng of type {type(rng).__name__}: pass None, an int, an integer "
                  "key array, a numpy SeedSequence / Generator, or rng_split_fn=")


# ----------------------------------------------------------------------------- next inputs
def _get_next_inputs(prev_inputs: xarray.Dataset, next_frame: xarray.Dataset) -> xarray.Dataset:
  """The autoregressive window: same number of frames as before, advanced by the frames of
  ``next_frame`` (predictions merged with the forcings of the times just predicted).  Inputs
  without a time axis (statics) are carried over; an input WITH a time axis that is neither
  predicted nor forced cannot be advanced (reference :581-604)."""
  fed_back = set(next_frame.keys())
  for name in prev_inputs.keys():
    if name not in fed_back and "time" in prev_inputs[name].dims:
      raise ValueError("Found an input with a time index that is not predicted or forced.")
  window = prev_inputs.sizes["time"]
  newest = next_frame[[name for name in next_frame.keys() if name in prev_inputs.keys()]]
  joined = xarray.concat([prev_inputs, newest], dim="time", data_vars="different", compat="equals")
  return joined.tail(time=window)


# ----------------------------------------------------------------------------- generator
class _ChunkSchedule:
  """What is fixed before the first step of a chunked rollout: validated chunking, the datasets
  stripped of absolute ``datetime`` (the predictor sees lead times only), and the RELATIVE time
  axes every chunk is presented with -- chunk k of the rollout looks to the predictor exactly
  like chunk 0 (reference :421-463)."""

  def __init__(self, inputs, targets_template, forcings, steps_per_chunk):
    if forcings is None:
      # (the reference iterates `.coords` keys here and cannot unpack them)
      forcings = xarray.Dataset({}, coords={
          n: c for n, c in targets_template.coords.items() if "time" in c.dims})
    self.template = self._without_datetime(targets_template)
    self.forcings = self._without_datetime(forcings)
    self.first_inputs = self._without_datetime(inputs)
    self.datetime = targets_template.coords["datetime"] if "datetime" in targets_template.coords else None
    total = self.template.sizes["time"]
    self.steps_per_chunk = steps_per_chunk
    self.num_chunks, leftover = divmod(total, steps_per_chunk)
    if leftover != 0:
      raise ValueError(
          f"The number of steps per chunk {steps_per_chunk} must "
          f"evenly divide the number of target steps {total} ")
    lead = np.asarray(self.template.coords["time"].values)
    if len(np.unique(np.diff(lead))) > 1:
      raise ValueError("The targets time coordinates must be evenly spaced")
    self.inputs_time = self.first_inputs.coords["time"].values
    self.chunk_time = lead[:steps_per_chunk]

  @staticmethod
  def _without_datetime(ds):
    ds = ds.copy()                         # never mutate the caller's datasets
    if "datetime" in ds.coords:
      del ds.coords["datetime"]
    return ds

  def rows(self, k):
    return slice(k * self.steps_per_chunk, (k + 1) * self.steps_per_chunk)

  def chunk(self, k, stage):
    """(template, forcings) of chunk k, staged (replicated / moved to the device) and relabelled
    with the first chunk's lead times, plus the true time-indexed coordinates of that chunk."""
    template = self.template.isel(time=self.rows(k)).compute()
    forcings = self.forcings.isel(time=self.rows(k)).compute()
    true_coords = {n: c for n, c in template.coords.items() if "time" in c.dims}
    relabel = lambda ds: stage(ds).assign_coords(time=self.chunk_time)
    return relabel(template), relabel(forcings), true_coords

  def stamp(self, predictions, k, true_coords):
    predictions = predictions.assign_coords(true_coords)
    if self.datetime is not None:
      predictions.coords["datetime"] = self.datetime.isel(time=self.rows(k))
    return predictions


def chunked_prediction_generator(
    predictor_fn: PredictorFn,
    rng: Any,
    inputs: xarray.Dataset,
    targets_template: xarray.Dataset,
    num_steps_per_chunk: int,
    forcings: Optional[xarray.Dataset] = None,
    verbose: bool = False,
    pmap_devices: Optional[Sequence[Any]] = None,
    replica_axis: Optional[str] = None,
    device_put_fn: Optional[Callable[[xarray.Dataset], xarray.Dataset]] = None,
    replicate_fn: Optional[Callable[[xarray.Dataset], xarray.Dataset]] = None,
    rng_split_fn: Optional[Callable[[Any], tuple]] = None,
) -> Iterator[xarray.Dataset]:
  """Yields the predictions of each chunk of a chunked rollout (reference :367-565).

  ``rng_split_fn`` (not in the reference): how to split an rng key this module does not know
  (see ``split_rng``)."""
  if pmap_devices is not None:
    raise ValueError(
        "pmap_devices is a single-process multi-device feature of the reference; this build runs "
        "one process per GPU: use chunked_prediction_generator_multiple_runs(rank=, world_size=).")
  if (replicate_fn is None) ^ (replica_axis is None):
    raise ValueError("Must provide replicate_fn when replica_axis is provided.")

  def stage(ds):
    if replicate_fn is not None:
      ds = replicate_fn(ds)
    return device_put_fn(ds) if device_put_fn is not None else ds

  # (a host that has xarray may pass its own Datasets: adapted once, here; predictions come back as xarray_lite
  #  Datasets -- xarray_lite.to_xarray(predictions, xarray) converts them for such a host)
  inputs, targets_template, forcings = (xarray.from_xarray(inputs), xarray.from_xarray(targets_template),
                                        xarray.from_xarray(forcings))
  schedule = _ChunkSchedule(inputs, targets_template, forcings, num_steps_per_chunk)
  state = stage(schedule.first_inputs)          # the rolling input window; stays where `stage` put it
  del inputs
  for k in range(schedule.num_chunks):
    if verbose:
      log.info("Chunk %d/%d", k, schedule.num_chunks)
    template_k, forcings_k, true_coords = schedule.chunk(k, stage)
    rng, key = split_rng(rng, rng_split_fn)
    state = state.assign_coords(time=schedule.inputs_time)
    predictions = predictor_fn(rng=key, inputs=state, targets_template=template_k, forcings=forcings_k)
    # feed back: what was just predicted plus the forcings valid at those times
    state = (_get_next_inputs(state, predictions.assign(forcings_k))
             if k + 1 < schedule.num_chunks else None)
    yield schedule.stamp(predictions, k, true_coords)


def _to_host(ds: xarray.Dataset) -> xarray.Dataset:
  """``jax.device_get`` of the reference (:362): torch-backed variables -> numpy."""
  return xarray.to_host(ds)


def chunked_prediction(
    predictor_fn: PredictorFn,
    rng: Any,
    inputs: xarray.Dataset,
    targets_template: xarray.Dataset,
    forcings: Optional[xarray.Dataset] = None,
    num_steps_per_chunk: int = 1,
    **kwargs,
) -> xarray.Dataset:
  """Long trajectory by concatenating chunked predictions in time (reference :326-364)."""
  chunks_list = []
  for prediction_chunk in chunked_prediction_generator(
      predictor_fn=predictor_fn, rng=rng, inputs=inputs, targets_template=targets_template,
      forcings=forcings, num_steps_per_chunk=num_steps_per_chunk, **kwargs):
    chunks_list.append(_to_host(prediction_chunk))
    del prediction_chunk
  return xarray.concat(chunks_list, dim="time")


# ----------------------------------------------------------------------------- ensembles
def _slice_sample_if_present(inputs, forcings, sample_idx):
  """reference :313-323."""
  if "sample" in inputs.dims:
    inputs = inputs.isel(sample=sample_idx)
  if forcings is not None and "sample" in forcings.dims:
    forcings = forcings.isel(sample=sample_idx)
  return inputs, forcings


def chunked_prediction_generator_multiple_runs(
    predictor_fn: PredictorFn,
    rngs: Sequence[Any],
    inputs: xarray.Dataset,
    targets_template: xarray.Dataset,
    forcings: Optional[xarray.Dataset],
    num_samples: Optional[int],
    pmap_devices: Optional[Sequence[Any]] = None,
    rank: int = 0,
    world_size: int = 1,
    **chunked_prediction_kwargs,
) -> Iterator[xarray.Dataset]:
  """Rolls out several ensemble members (reference :158-307).

  All lead-time chunks of one member are yielded before the next member starts, each
  carrying the scalar coordinate ``sample`` = member index, exactly as the reference's
  un-pmapped branch (:286-307).  With ``world_size`` > 1 (one process per GPU) this rank
  only rolls out members ``rank, rank + world_size, ...``: members never interact, so
  there is no collective (``ensemble.gather_member_chunks`` collects results if wanted).
  """
  if pmap_devices is not None:
    raise ValueError("pmap_devices is not supported: launch one process per GPU and pass "
                     "rank= / world_size= instead.")
  if num_samples is None:
    if "sample" not in inputs.dims:
      raise ValueError(
          "The number of samples must be passed when `inputs` don't have a `sample` dim.")
    num_samples = inputs.sizes["sample"]
  if "sample" in inputs.dims and num_samples != inputs.sizes["sample"]:
    raise ValueError(
        f"Inconsistent number of samples requested for inputs{num_samples} != "
        f"{inputs.sizes['sample']}.")
  if num_samples != len(rngs):
    raise ValueError(f"Inconsistent number of rngs passed. {num_samples} != {len(rngs)}.")
  if forcings:
    if "sample" in forcings.dims and num_samples != forcings.sizes["sample"]:
      raise ValueError(
          f"Inconsistent number of samples requested for forcings{num_samples} != "
          f"{forcings.sizes['sample']}.")
  if not 0 <= rank < world_size:
    raise ValueError(f"rank {rank} outside world of size {world_size}")

  for i in range(rank, num_samples, world_size):
    log.info("Sample %d/%d", i, num_samples)
    sample_inputs, sample_forcings = _slice_sample_if_present(inputs, forcings, sample_idx=i)
    for prediction_chunk in chunked_prediction_generator(
        predictor_fn, rngs[i], inputs=sample_inputs, targets_template=targets_template,
        forcings=sample_forcings, **chunked_prediction_kwargs):
      prediction_chunk.coords["sample"] = xarray.Variable((), np.asarray(i))
      yield prediction_chunk
    log.info("Completed sample %d/%d", i, num_samples)


def extend_targets_template(targets_template: xarray.Dataset, required_num_steps: int,
                            value: Optional[float] = None) -> xarray.Dataset:
  """Template of ``required_num_steps`` equispaced lead times (reference :618-690).

  The reference fills it with lazy dask arrays; here the data are zero-stride broadcasts of
  one scalar (``value`` or 0), so nothing of the extended size is ever allocated."""
  time = np.asarray(targets_template.coords["time"].values)
  timestep = time[0]
  if time.shape[0] > 1:
    assert np.all(timestep == time[1:] - time[:-1])
  extended_time = (np.arange(required_num_steps) + 1) * timestep
  coords = {k: v for k, v in targets_template._coords.items() if "time" not in v.dims}
  coords["time"] = xarray.Variable(("time",), extended_time)
  if "datetime" in targets_template.coords:
    datetime = np.asarray(targets_template.coords["datetime"].values)
    coords["datetime"] = xarray.Variable(("time",), (datetime[0] - timestep) + extended_time)

  def extend_time(v: xarray.Variable) -> xarray.Variable:
    if "time" not in v.dims:
      return v
    shape = tuple(required_num_steps if d == "time" else n for d, n in zip(v.dims, v.shape))
    if xarray._is_torch(v.data):
      import torch
      fill = torch.full((), 0.0 if value is None else value, dtype=v.data.dtype,
                        device=v.data.device)
      return xarray.Variable(v.dims, fill.expand(*shape))
    fill = np.full((), 0 if value is None else value, dtype=v.dtype)
    return xarray.Variable(v.dims, np.broadcast_to(fill, shape))

  return xarray.Dataset._construct(
      {k: extend_time(v) for k, v in targets_template._vars.items()}, coords)
