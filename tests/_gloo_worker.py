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


def main():
  out_path, num_members = sys.argv[1], int(sys.argv[2])
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  dist.init_process_group("gloo", rank=rank, world_size=world)
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
