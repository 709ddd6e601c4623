"""The seeded case of tests/golden/gnn_deepgnn512.npz -- sizes, DenseLayer kwargs, inputs, sampled rows -- in a
module WITHOUT reference imports: the generator (make_golden_deepgnn.py, build container only) and the tests (CPU
and GPU box, where /root/reference does not exist) both take it from here."""
import numpy as np

LATENT, STEPS, SEED = 512, 2, 9
DENSE = dict(hidden_size=LATENT, output_size=LATENT, num_hidden_layers=1, activation="swish",
             activation_normalization="layer_norm", activate_final=False, with_bias=True,
             one_less_layer_when_activate_final=False, w_init=None, b_init=None,
             activation_normalization_kwargs=None)


def inputs():
  """A mesh-like typed graph: one node set, one receiver-SORTED edge set (what utils/padding_utils.py
  hands DeepGNN) with in-degrees 0 .. 150 (empty segments, runs across 64-row tile borders)."""
  rng = np.random.default_rng(31)
  n, batch = 260, 2
  deg = np.concatenate([[0, 150, 0, 70], rng.integers(1, 14, n - 4)])
  recv = np.repeat(np.arange(n), deg)
  snd = rng.integers(0, n, len(recv))
  f32 = lambda a: a.astype(np.float32).astype(np.float64)
  return dict(n=n, batch=batch, senders=snd, receivers=recv, h=f32(rng.standard_normal((n, batch, LATENT))),
              e=f32(rng.standard_normal((len(recv), batch, LATENT))))


def sample_rows(x):
  rng = np.random.default_rng(5)
  return (np.sort(rng.choice(x["n"], 32, replace=False)), np.sort(rng.choice(len(x["receivers"]), 64, replace=False)))
