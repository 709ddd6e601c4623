"""Oracle: the WN2 ``DeepGNN`` processor (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``/root/reference/weathernext/utils/deep_gnn.py`` for the configuration the HIP kernels build
(``dense.DenseLayer`` = MLP with one hidden layer + "layer_norm", swish; ``utils/dense.py:56-141,
233-357``):
  * ``_process`` / ``_process_step`` (:317-400): ``num_processor_repetitions`` x the
    ``num_message_passing_steps`` unshared InteractionNetwork steps, node residuals always, edge
    residuals when ``use_edge_residuals``;
  * one step (``typed_graph_net.py:272-350,369-546``): every edge set
    e' = f_e([e | h_send[senders] | h_recv[receivers]]), then every node set
    h' = f_n([h | segment_sum(e', receivers) per incoming edge set]), sent messages dropped;
  * ``pre_gather_matmul`` (:224-262 + ``dense.summed_args``): the edge MLP's first matrix split into
    three bias-free Linear layers applied BEFORE the gathers, their outputs summed, then the MLP with
    ``drop_first_matmul`` (linear_0 = bias only) -- the same function as the concat form with
    W1 = [W_edge; W_sender; W_receiver].
Pinned by tests/golden/gnn_deepgnn512.npz = the reference source executed on the stand-ins
(tests/golden/make_golden_deepgnn.py); the arithmetic primitives stay parity-unpinned like everywhere."""
import numpy as np

from oracle import gnn


def _dense(params, stem, x, dtype, first_matrix=True):
  p = lambda m, l: np.asarray(params[f"{stem}/{m}"][l], dtype=dtype)
  z = gnn.linear(x, p("mlp/linear_0", "w"), p("mlp/linear_0", "b")) if first_matrix else x + p("mlp/linear_0", "b")
  y = gnn.linear(gnn.swish(z), p("mlp/linear_1", "w"), p("mlp/linear_1", "b"))
  return gnn.layer_norm(y, p("normalization/layer_norm", "scale"), p("normalization/layer_norm", "offset"))


def forward(params, nodes, edges, *, num_message_passing_steps, num_processor_repetitions=1,
            use_edge_residuals=True, pre_gather_matmul=False, name="DeepGNN", dtype=np.float64):
  """nodes {set: [N, B, D]}; edges {set: dict(senders_set, receivers_set, senders, receivers, features [E, B, D])}
  -> (nodes, {set: features})."""
  nodes = {k: np.asarray(v, dtype=dtype) for k, v in nodes.items()}
  feats = {k: np.asarray(e["features"], dtype=dtype) for k, e in edges.items()}
  for _ in range(num_processor_repetitions):
    for i in range(num_message_passing_steps):
      new_e = {}
      for k, e in edges.items():
        hs, hr = nodes[e["senders_set"]][e["senders"]], nodes[e["receivers_set"]][e["receivers"]]
        stem = f"{name}/processor_edges_{i}_{k}"
        if pre_gather_matmul:
          w = lambda part: np.asarray(params[f"{name}/processor_edges_{i}_{part}_{k}"]["w"], dtype=dtype)
          lin = lambda x, m: (x.reshape(-1, x.shape[-1]) @ m).reshape(x.shape[:-1] + (m.shape[1],))
          # (the reference multiplies the node tables and gathers afterwards: the same numbers)
          summed = lin(feats[k], w("edge")) + lin(hs, w("sender")) + lin(hr, w("receiver"))
          new_e[k] = _dense(params, stem, summed, dtype, first_matrix=False)
        else:
          new_e[k] = _dense(params, stem, np.concatenate([feats[k], hs, hr], axis=-1), dtype)
      new_n = {}
      for n, h in nodes.items():
        received = [gnn.segment_sum(new_e[k], e["receivers"], h.shape[0])
                    for k, e in sorted(edges.items()) if e["receivers_set"] == n]
        new_n[n] = _dense(params, f"{name}/processor_nodes_{i}_{n}", np.concatenate([h] + received, axis=-1), dtype)
      nodes = {n: nodes[n] + new_n[n] for n in nodes}
      feats = {k: (feats[k] + new_e[k]) if use_edge_residuals else new_e[k] for k in feats}
  return nodes, feats
