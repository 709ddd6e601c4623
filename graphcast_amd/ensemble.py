"""Ensemble members across GPUs: one process per GPU, member ``i`` on rank ``i % N``.

Members of a GraphCast ensemble never interact in the forward pass (the batch /
sample axis is a pure broadcast axis, reference ``graphcast.py:726-730``; the
reference ``pmap``s over ``sample``, ``rollout.py:220-283``).  So the path shards
with **no data-path collective**: every rank holds a replicated plan (weights +
packed graphs, ~0.25 GB + folded constants) and rolls out its own members;
the only communication is an optional gather of results and the barrier /
max-over-ranks timing of the benchmark.  ``torch.distributed`` backend ``nccl``
is RCCL on ROCm; the CPU tests run the same code over ``gloo``.

``WithSampleDim`` mirrors ``weathernext/utils/ensemble.py:22-55``.
"""
from typing import Any, Callable, Dict, List, Optional


from graphcast_amd import predictor_base
from graphcast_amd import xarray_lite as xarray


def members_of_rank(num_members: int, rank: int, world_size: int) -> List[int]:
  """Round-robin ownership: contiguous load balance for any N (|diff| <= 1 member)."""
  if not 0 <= rank < world_size:
    raise ValueError(f"rank {rank} outside world of size {world_size}")
  return list(range(rank, num_members, world_size))


def owner_of_member(member: int, world_size: int) -> int:
  return member % world_size


def run_members(step_fn: Callable[[int], Any], num_members: int, rank: int,
                world_size: int) -> Dict[int, Any]:
  """Runs ``step_fn(member)`` for the members this rank owns -> {member: result}."""
  return {m: step_fn(m) for m in members_of_rank(num_members, rank, world_size)}


def gather_member_arrays(local: Dict[int, Any], num_members: int, *, dst: Optional[int] = 0,
                         group=None):
  """Collects per-member tensors on ``dst`` (or on every rank if ``dst`` is None).

  ``local`` maps member -> torch tensor (same shape/dtype for all members).  Uses ONE
  collective (all_gather of a [members_per_rank, ...] stack, padded to equal length), not one
  per member: over xGMI a single large ring step beats many small ones.  Returns a list
  indexed by member (None on ranks other than ``dst``)."""
  import torch
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    return [local[m] for m in range(num_members)]
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  per_rank = (num_members + world - 1) // world
  mine = members_of_rank(num_members, rank, world)
  if not local:
    raise ValueError("every rank must own at least one member to define the result shape")
  ref = next(iter(local.values()))
  stack = torch.zeros((per_rank,) + tuple(ref.shape), dtype=ref.dtype, device=ref.device)
  for j, m in enumerate(mine):
    stack[j].copy_(local[m])
  parts = [torch.empty_like(stack) for _ in range(world)]
  dist.all_gather(parts, stack, group=group)
  if dst is not None and rank != dst:
    return None
  out = [None] * num_members
  for r in range(world):
    for j, m in enumerate(members_of_rank(num_members, r, world)):
      out[m] = parts[r][j]
  return out


def max_over_ranks(seconds: float, device=None, group=None) -> float:
  """The benchmark's clock: the slowest rank defines the time of the step."""
  import torch
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    return seconds
  t = torch.tensor([seconds], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
  return float(t.item())


class WithSampleDim(predictor_base.Predictor):
  """Adds a leading ``sample`` dimension by broadcasting (reference ensemble.py:22-55)."""

  def __init__(self, underlying: predictor_base.Predictor, num_samples: int):
    self._underlying = underlying
    self._num_samples = num_samples

  def _add_sample_axis(self, data_array):
    v = data_array.variable
    dims = ("sample",) + tuple(v.dims)
    sizes = (self._num_samples,) + tuple(v.shape)
    return xarray.DataArray(v.set_dims(dims, sizes), coords=dict(data_array._coords),
                            name=data_array.name)

  def __call__(self, inputs, targets_template, forcings=None, **kwargs):
    inputs = inputs.map(self._add_sample_axis)
    targets_template = targets_template.map(self._add_sample_axis)
    if forcings is not None:
      forcings = forcings.map(self._add_sample_axis)
    return self._underlying(inputs, targets_template, forcings=forcings)

  def loss(self, inputs, targets, forcings=None, **kwargs):
    inputs = inputs.map(self._add_sample_axis)
    targets = targets.map(self._add_sample_axis)
    if forcings is not None:
      forcings = forcings.map(self._add_sample_axis)
    return self._underlying.loss(inputs, targets, forcings=forcings, **kwargs)
