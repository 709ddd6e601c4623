"""Generates tests/golden/gnn_conditioned.npz: the reference's DeepTypedGraphNet with
``use_norm_conditioning=True`` (the GenCast encoder / decoder configuration,
weathernext1_gen/denoiser.py:303-363) executed UNMODIFIED --
utils/legacy/deep_typed_graph_net.py, utils/typed_graph_net.py, utils/typed_graph.py,
utils/dense.py (LinearNormConditioning) -- on the haiku / jraph / jax stand-ins of
tests/golden/ref_shims, on a small bipartite typed graph.  Pins the conditional-LayerNorm wiring
and its parameter names for oracle/gnn.py.

    python tests/golden/make_golden_conditioned.py
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, REF)

import typing                                                # noqa: E402
import typing_extensions                                     # noqa: E402
for n in ("Required", "NotRequired"):
  if not hasattr(typing, n):
    setattr(typing, n, getattr(typing_extensions, n))

import haiku as hk                                           # noqa: E402  (numpy stand-in)
from weathernext.utils import typed_graph                    # noqa: E402
from weathernext.utils.legacy import deep_typed_graph_net    # noqa: E402


def main():
  small()
  latent512()


def _graph(n_grid, n_mesh):
  def graph(edge_name, senders, receivers, feats, send_set, recv_set, grid, mesh):
    return typed_graph.TypedGraph(
        context=typed_graph.Context(n_graph=np.array([1]), features=()),
        nodes={"grid_nodes": typed_graph.NodeSet(n_node=np.array([n_grid]), features=grid),
               "mesh_nodes": typed_graph.NodeSet(n_node=np.array([n_mesh]), features=mesh)},
        edges={typed_graph.EdgeSetKey(edge_name, (send_set, recv_set)): typed_graph.EdgeSet(
            n_edge=np.array([len(senders)]),
            indices=typed_graph.EdgesIndices(senders=senders, receivers=receivers), features=feats)})
  return graph


def latent512():
  """The same reference code at the width the HIP kernels are built for (latent 512), on a graph
  with receivers of very different in-degree (0 .. > 64 edges: empty segments, runs across tile
  borders).  Parameters are regenerated from a seed on both sides (oracle.params.
  init_conditioned_params); their digest is in the fixture."""
  from oracle import params as oparams
  rng = np.random.default_rng(23)
  n_grid, n_mesh, batch, latent, c_cond = 300, 40, 2, 512, 16
  c_grid, c_mesh, c_edge, c_out, seed = 41, 7, 4, 29, 5
  deg = np.concatenate([[0, 0, 150, 70], rng.integers(1, 12, n_mesh - 4)])      # in-degree per mesh node
  g2m_r = np.repeat(np.arange(n_mesh), deg)
  g2m_s = rng.integers(0, n_grid, len(g2m_r))
  perm = rng.permutation(len(g2m_r))                                            # construction order != sorted
  g2m_s, g2m_r = g2m_s[perm], g2m_r[perm]
  m2g_s, m2g_r = rng.integers(0, n_mesh, 3 * n_grid), np.repeat(np.arange(n_grid), 3)
  f32 = lambda a: a.astype(np.float32).astype(np.float64)      # inputs exactly representable in fp32
  grid_x = f32(rng.standard_normal((n_grid, batch, c_grid)))
  mesh_x = f32(rng.standard_normal((n_mesh, batch, c_mesh)))
  g2m_e = np.broadcast_to(f32(rng.standard_normal((len(g2m_r), 1, c_edge))), (len(g2m_r), batch, c_edge)).copy()
  m2g_e = np.broadcast_to(f32(rng.standard_normal((3 * n_grid, 1, c_edge))), (3 * n_grid, batch, c_edge)).copy()
  cond = f32(rng.standard_normal((batch, c_cond)))
  params = {k: {l: np.asarray(v, np.float64) for l, v in m.items()}
            for k, m in oparams.init_conditioned_params(c_grid, c_mesh, c_edge, c_cond, c_out, latent, seed=seed).items()}
  graph = _graph(n_grid, n_mesh)

  def encoder(g, c):
    return deep_typed_graph_net.DeepTypedGraphNet(
        activation="swish", aggregate_normalization=None, edge_latent_size=dict(grid2mesh=latent),
        embed_edges=True, embed_nodes=True, f32_aggregation=True,
        include_sent_messages_in_node_update=False, mlp_hidden_size=latent, mlp_num_hidden_layers=1,
        name="grid2mesh_gnn", node_latent_size=dict(grid_nodes=latent, mesh_nodes=latent),
        node_output_size=None, num_message_passing_steps=1, use_layer_norm=True,
        use_norm_conditioning=True)(g, global_norm_conditioning=c)

  def decoder(g, c):
    return deep_typed_graph_net.DeepTypedGraphNet(
        activation="swish", edge_latent_size=dict(mesh2grid=latent), embed_nodes=False,
        f32_aggregation=False, include_sent_messages_in_node_update=False, mlp_hidden_size=latent,
        mlp_num_hidden_layers=1, name="mesh2grid_gnn",
        node_latent_size=dict(grid_nodes=latent, mesh_nodes=latent),
        node_output_size={"grid_nodes": c_out}, num_message_passing_steps=1, use_layer_norm=True,
        use_norm_conditioning=True)(g, global_norm_conditioning=c)

  n_before = len(params)
  with hk.running(params):
    enc = encoder(graph("grid2mesh", g2m_s, g2m_r, g2m_e, "grid_nodes", "mesh_nodes", grid_x, mesh_x), cond)
    dec = decoder(graph("mesh2grid", m2g_s, m2g_r, m2g_e, "mesh_nodes", "grid_nodes",
                        enc.nodes["grid_nodes"].features, enc.nodes["mesh_nodes"].features), cond)
  assert len(params) == n_before, "the reference asked for a parameter the seeded tree does not have"
  out = dict(config=np.array([c_grid, c_mesh, c_edge, c_cond, c_out, latent, seed]),
             params_sha256=np.array(oparams.digest(params)),
             grid_x=grid_x.astype(np.float32), mesh_x=mesh_x.astype(np.float32),
             g2m_e=g2m_e[:, 0].astype(np.float32), m2g_e=m2g_e[:, 0].astype(np.float32), cond=cond.astype(np.float32),
             g2m_senders=g2m_s, g2m_receivers=g2m_r, m2g_senders=m2g_s, m2g_receivers=m2g_r,
             enc_grid=np.asarray(enc.nodes["grid_nodes"].features), enc_mesh=np.asarray(enc.nodes["mesh_nodes"].features),
             dec_grid=np.asarray(dec.nodes["grid_nodes"].features))
  path = os.path.join(HERE, "gnn_conditioned512.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith(("enc", "dec"))})


def small():
  rng = np.random.default_rng(11)
  n_grid, n_mesh, batch, latent, c_cond = 40, 9, 2, 16, 6
  n_g2m, n_m2g = 70, 3 * n_grid
  g2m_s, g2m_r = rng.integers(0, n_grid, n_g2m), rng.integers(0, n_mesh, n_g2m)
  m2g_s, m2g_r = rng.integers(0, n_mesh, n_m2g), np.repeat(np.arange(n_grid), 3)
  grid_x = rng.standard_normal((n_grid, batch, 5))
  mesh_x = rng.standard_normal((n_mesh, batch, 5))
  g2m_e = np.broadcast_to(rng.standard_normal((n_g2m, 1, 4)), (n_g2m, batch, 4)).copy()
  m2g_e = np.broadcast_to(rng.standard_normal((n_m2g, 1, 4)), (n_m2g, batch, 4)).copy()
  cond = rng.standard_normal((batch, c_cond))

  def graph(edge_name, senders, receivers, feats, send_set, recv_set, grid, mesh):
    return typed_graph.TypedGraph(
        context=typed_graph.Context(n_graph=np.array([1]), features=()),
        nodes={"grid_nodes": typed_graph.NodeSet(n_node=np.array([n_grid]), features=grid),
               "mesh_nodes": typed_graph.NodeSet(n_node=np.array([n_mesh]), features=mesh)},
        edges={typed_graph.EdgeSetKey(edge_name, (send_set, recv_set)): typed_graph.EdgeSet(
            n_edge=np.array([len(senders)]),
            indices=typed_graph.EdgesIndices(senders=senders, receivers=receivers), features=feats)})

  def encoder(g, c):                                   # denoiser.py:303-330
    net = deep_typed_graph_net.DeepTypedGraphNet(
        activation="swish", aggregate_normalization=None, edge_latent_size=dict(grid2mesh=latent),
        embed_edges=True, embed_nodes=True, f32_aggregation=True,
        include_sent_messages_in_node_update=False, mlp_hidden_size=latent, mlp_num_hidden_layers=1,
        name="grid2mesh_gnn", node_latent_size=dict(grid_nodes=latent, mesh_nodes=latent),
        node_output_size=None, num_message_passing_steps=1, use_layer_norm=True,
        use_norm_conditioning=True)
    return net(g, global_norm_conditioning=c)

  def decoder(g, c):                                   # denoiser.py:340-363
    net = deep_typed_graph_net.DeepTypedGraphNet(
        activation="swish", edge_latent_size=dict(mesh2grid=latent), embed_nodes=False,
        f32_aggregation=False, include_sent_messages_in_node_update=False, mlp_hidden_size=latent,
        mlp_num_hidden_layers=1, name="mesh2grid_gnn",
        node_latent_size=dict(grid_nodes=latent, mesh_nodes=latent),
        node_output_size={"grid_nodes": 7}, num_message_passing_steps=1, use_layer_norm=True,
        use_norm_conditioning=True)
    return net(g, global_norm_conditioning=c)

  params = {}
  g_enc = graph("grid2mesh", g2m_s, g2m_r, g2m_e, "grid_nodes", "mesh_nodes", grid_x, mesh_x)
  with hk.running(params, init_rng=rng):
    enc = encoder(g_enc, cond)
    g_dec = graph("mesh2grid", m2g_s, m2g_r, m2g_e, "mesh_nodes", "grid_nodes",
                  enc.nodes["grid_nodes"].features, enc.nodes["mesh_nodes"].features)
    decoder(g_dec, cond)
  # the conditioning layers are initialised at ~1e-8 (dense.py:381,385): make them matter
  for mod, leaves in params.items():
    for k in leaves:
      if mod.endswith("_norm_conditioning/linear") or k == "b":
        leaves[k] = (0.3 * rng.standard_normal(leaves[k].shape)).astype(np.float64)
  with hk.running(params):
    enc = encoder(g_enc, cond)
    g_dec = graph("mesh2grid", m2g_s, m2g_r, m2g_e, "mesh_nodes", "grid_nodes",
                  enc.nodes["grid_nodes"].features, enc.nodes["mesh_nodes"].features)
    dec = decoder(g_dec, cond)
  out = {f"params:{mod}:{leaf}": np.asarray(v) for mod, leaves in params.items() for leaf, v in leaves.items()}
  out.update(grid_x=grid_x, mesh_x=mesh_x, g2m_e=g2m_e, m2g_e=m2g_e, cond=cond,
             g2m_senders=g2m_s, g2m_receivers=g2m_r, m2g_senders=m2g_s, m2g_receivers=m2g_r,
             enc_grid=np.asarray(enc.nodes["grid_nodes"].features), enc_mesh=np.asarray(enc.nodes["mesh_nodes"].features),
             dec_grid=np.asarray(dec.nodes["grid_nodes"].features))
  path = os.path.join(HERE, "gnn_conditioned.npz")
  np.savez_compressed(path, **out)
  print("wrote", path)
  print(sorted(k for k in params if "norm_conditioning" in k)[:4], "...",
        sum("layer_norm" in k for k in params), "layer_norm modules with parameters")


if __name__ == "__main__":
  main()
