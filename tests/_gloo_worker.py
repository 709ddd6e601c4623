"""Worker of tests/test_ensemble_gloo.py: one process per 'GPU' (here: CPU + gloo)."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from graphcast_amd import ensemble            # noqa: E402


def member_state(m, n=64, c=5):
  return torch.from_numpy(np.random.default_rng(m).standard_normal((n, 1, c)).astype(np.float32))


def step(x):
  return torch.tanh(x * 1.5 + 0.25)


# ---- "rollout" mode (round 6): the PRODUCT's multi-process ensemble path instead of a toy step --
#      rollout.chunked_prediction_generator_multiple_runs(rank=, world_size=) around a stub Predictor (a fixed linear map +
#      tanh on the stacked channels: predictor_base.Predictor, Datasets in, Datasets out, autoregressive feedback through
#      rollout._get_next_inputs), the members' trajectories collected with ensemble.gather_member_arrays.
import dataclasses                                 # noqa: E402

from graphcast_amd import graphcast as gc          # noqa: E402
from graphcast_amd import model_utils              # noqa: E402
from graphcast_amd import predictor_base           # noqa: E402
from graphcast_amd import rollout                  # noqa: E402
from graphcast_amd import synthetic                # noqa: E402
from graphcast_amd import xarray_lite as xarray    # noqa: E402

R_LAT, R_LON, R_STEPS = np.arange(-90, 91, 30.0), np.arange(0, 360, 45.0), 3
R_TASK = dataclasses.replace(gc.TASK_13, pressure_levels=(500, 850, 1000))


class StubPredictor(predictor_base.Predictor):
  def __init__(self, c_in, c_out, seed=3):
    self.a = (np.random.default_rng(seed).standard_normal((c_in, c_out)) / np.sqrt(c_in)).astype(np.float32)

  def __call__(self, inputs, targets_template, forcings, **kw):
    x = xarray.concat([model_utils.dataset_to_stacked(inputs), model_utils.dataset_to_stacked(forcings)], dim="channels")
    y = np.tanh(np.asarray(x.data, np.float32) @ self.a)
    return model_utils.stacked_to_dataset(xarray.Variable(("batch", "lat", "lon", "channels"), y), targets_template)


def rollout_case(num_members):
  """(predictor_fn, rngs, inputs with a "sample" axis, template, forcings): every member its own initial state."""
  per = [synthetic.make_example(R_TASK, R_LAT, R_LON, num_target_steps=R_STEPS, seed=70 + m) for m in range(num_members)]
  i0, template, f0 = per[0]
  inputs = xarray.Dataset({k: ((("sample",) + i0[k].dims), np.stack([p[0][k].values for p in per])) if "time" in i0[k].dims
                           else (i0[k].dims, i0[k].values) for k in i0.keys()}, coords=dict(i0._coords))
  one_f = f0.isel(time=slice(0, 1))
  c_in = (model_utils.dataset_to_stacked(i0).sizes["channels"] + model_utils.dataset_to_stacked(one_f).sizes["channels"])
  c_out = model_utils.dataset_to_stacked(template.isel(time=slice(0, 1))).sizes["channels"]
  predictor = StubPredictor(c_in, c_out)
  return (lambda rng, **kw: predictor(**kw)), list(range(num_members)), inputs, template, f0


def member_trajectories(num_members, rank, world):
  """{member: [T, channels-stacked prediction] tensor} of the members `rank` owns, through the product's generator."""
  fn, rngs, inputs, template, forcings = rollout_case(num_members)
  by_member = {}
  for chunk in rollout.chunked_prediction_generator_multiple_runs(
      fn, rngs, inputs, template, forcings, num_samples=None, num_steps_per_chunk=1, rank=rank, world_size=world):
    m = int(chunk.coords["sample"].values)
    by_member.setdefault(m, []).append(np.asarray(model_utils.dataset_to_stacked(chunk).values, np.float32))
  return {m: torch.from_numpy(np.concatenate(v, axis=0)) for m, v in by_member.items()}


def main():
  out_path, num_members = sys.argv[1], int(sys.argv[2])
  mode = sys.argv[3] if len(sys.argv) > 3 else "toy"
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  dist.init_process_group("gloo", rank=rank, world_size=world)
  if mode == "rollout":
    local = member_trajectories(num_members, rank, world)
    everyone = ensemble.gather_member_arrays(local, num_members, dst=None)
    np.savez(out_path + f".rank{rank}.npz", owned=np.array(sorted(local)), gathered=torch.stack(everyone).numpy())
    dist.barrier()
    dist.destroy_process_group()
    return
  t0 = time.perf_counter()
  local = ensemble.run_members(lambda m: step(step(member_state(m))), num_members, rank, world)
  elapsed = time.perf_counter() - t0 + 0.01 * rank          # rank-dependent: max must win
  slowest = ensemble.max_over_ranks(elapsed)
  everyone = ensemble.gather_member_arrays(local, num_members, dst=None)
  on_root = ensemble.gather_member_arrays(local, num_members, dst=0)
  assert (on_root is None) == (rank != 0)
  assert slowest >= elapsed - 1e-12
  np.savez(out_path + f".rank{rank}.npz", owned=np.array(sorted(local)), slowest=slowest,
           elapsed=elapsed, gathered=torch.stack(everyone).numpy())
  dist.barrier()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
