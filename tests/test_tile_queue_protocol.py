"""The dynamic tile queue's protocol (csrc/rowmlp_half.inc: tile_queue_fetch / tile_queue_leave; include/gcast.h:
gc_rowmlp_desc.tile_queue), restated as a tiny interleaving model and checked exhaustively-at-random: whatever the
order in which the workgroups' atomic operations land,

  * every tile of the launch is run exactly once,
  * both words are zero when the launch is over (so the next launch on the stream can reuse them), and
  * the reset cannot overtake a fetch (the last workgroup to leave has seen every other workgroup's last fetch).

The kernels themselves are compared bit for bit with and without the queue on the GPU (tests/test_rowmlp_gpu.py);
this file pins the reasoning the kernel comment gives, on the CPU."""
import random

import pytest


def run_launch(q, n_tiles, grid, rng, min_rounds=1):
  """One launch over the shared words q = [next, left].  Each workgroup is a little state machine whose atomic steps
  are interleaved in random order; returns the tiles each workgroup ran."""
  queued = n_tiles >= min_rounds * grid and n_tiles > grid
  ran = [[] for _ in range(grid)]
  # state per workgroup: ("tile", t) about to start tile t; ("leave",) about to leave; None = gone
  state = [("tile", b) if b < n_tiles else ("leave",) for b in range(grid)]
  live = [b for b in range(grid) if state[b] is not None]
  while live:
    b = rng.choice(live)
    kind = state[b][0]
    if kind == "tile":
      t = state[b][1]
      if queued:
        nxt = grid + q[0]; q[0] += 1                      # atomicAdd(q, 1): fetched at the TOP of the tile
      else:
        nxt = t + grid
      ran[b].append(t)                                    # ... the tile itself ...
      state[b] = ("tile", nxt) if nxt < n_tiles else ("leave",)
    else:
      if queued:
        left = q[1]; q[1] += 1                            # atomicAdd(q + 1, 1), after the workgroup's last fetch returned
        if left == grid - 1:
          q[0], q[1] = 0, 0                               # the last one out clears the pair
      state[b] = None
      live.remove(b)
  return ran


@pytest.mark.parametrize("n_tiles,grid", [(5120, 512), (641, 512), (2048, 512), (2049, 512), (16223, 256), (7, 512), (513, 512)])
def test_every_tile_once_and_words_zero_after(n_tiles, grid):
  rng = random.Random(n_tiles * 1000 + grid)
  q = [0, 0]
  for launch in range(3):                                 # launches one after another share the words
    g = min(grid, n_tiles)
    ran = run_launch(q, n_tiles, g, rng)
    tiles = sorted(t for r in ran for t in r)
    assert tiles == list(range(n_tiles)), f"launch {launch}: tiles lost or run twice"
    assert q == [0, 0], f"launch {launch}: queue words left at {q}"
    assert all(r and r[0] == b for b, r in enumerate(ran)), "a workgroup's first tile is its own index"


def test_the_host_rule_keeps_short_launches_static():
  """gcast.hip: tile_queue_pays -- below GC_TILE_QUEUE_MIN_ROUNDS tiles per workgroup the walk stays b, b + grid."""
  rng = random.Random(1)
  q = [0, 0]
  ran = run_launch(q, 641, 512, rng, min_rounds=4)
  assert all(r == list(range(b, 641, 512)) for b, r in enumerate(ran)) and q == [0, 0]
  ran = run_launch(q, 5120, 512, rng, min_rounds=4)
  assert sorted(t for r in ran for t in r) == list(range(5120))
  assert any(r != list(range(b, 5120, 512)) for b, r in enumerate(ran)), "ten rounds: handed out dynamically"
