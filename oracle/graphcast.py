"""Oracle: GraphCast encode-process-decode at the tensor boundary
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``/root/reference/weathernext/weathernext1_graph/graphcast.py``:
  * ``build_graphs``  <- ``__init__`` mesh/radius part (:196-198,266-270),
                         ``_init_mesh_properties`` (:380-394), ``_init_grid_properties``
                         (:396-406), ``_init_grid2mesh_graph`` (:408-458),
                         ``_init_mesh_graph`` (:460-497), ``_init_mesh2grid_graph`` (:499-548)
  * ``forward``       <- ``_run_grid2mesh_gnn`` (:550-604), ``_run_mesh_gnn`` (:606-639),
                         ``_run_mesh2grid_gnn`` (:641-678) with the three
                         ``DeepTypedGraphNet`` configurations of ``__init__`` (:202-262)
"""
import numpy as np

from oracle import connectivity, features, gnn, mesh


def build_graphs(lat, lon, mesh_size, radius_fraction=0.6, m2g_norm=None, m2g_face_indices=None):
  """Static structure: indices + structural features of the three graphs.
  `m2g_face_indices`: a given answer of the containing-face query (:114-119) instead of the
  restated rule (one finest-mesh face id per grid point)."""
  lat = np.asarray(lat).astype(np.float32)
  lon = np.asarray(lon).astype(np.float32)
  levels = mesh.mesh_hierarchy(mesh_size)
  verts, faces = levels[-1]
  radius = mesh.max_edge_length(verts, faces) * radius_fraction
  m_lat, m_lon = features.cartesian_to_lat_lon(verts)
  glon, glat = np.meshgrid(lon, lat)
  g_lat = glat.reshape([-1]).astype(np.float32)
  g_lon = glon.reshape([-1]).astype(np.float32)

  g2m_grid, g2m_mesh = connectivity.radius_query(lat, lon, verts, radius)
  g2m_feat, _ = features.edge_features(g_lat, g_lon, m_lat, m_lon, g2m_grid, g2m_mesh)
  ms, mr = mesh.faces_to_edges(mesh.merged_faces(levels))
  mesh_feat, _ = features.edge_features(m_lat, m_lon, m_lat, m_lon, ms, mr)
  if m2g_face_indices is None:
    m2g_grid, m2g_mesh = connectivity.containing_triangle_query(lat, lon, verts, faces)
  else:
    m2g_mesh = np.asarray(faces)[np.asarray(m2g_face_indices)].reshape([-1])
    m2g_grid = np.repeat(np.arange(len(g_lat)), 3)
  m2g_feat, _ = features.edge_features(m_lat, m_lon, g_lat, g_lon, m2g_mesh, m2g_grid,
                                       normalization=m2g_norm)
  return dict(
      n_grid=len(g_lat), n_mesh=len(verts), radius=radius,
      mesh_vertices=verts, mesh_faces=faces, mesh_lat=m_lat, mesh_lon=m_lon,
      grid_node_feat=features.node_features(g_lat, g_lon),
      mesh_node_feat=features.node_features(m_lat, m_lon),
      g2m=dict(senders=g2m_grid, receivers=g2m_mesh, feat=g2m_feat),
      mesh=dict(senders=ms, receivers=mr, feat=mesh_feat),
      m2g=dict(senders=m2g_mesh, receivers=m2g_grid, feat=m2g_feat))


def _batch(x, b, dtype):
  return np.repeat(np.asarray(x, dtype=dtype)[:, None, :], b, axis=1)


def forward(params, graphs, x_grid, steps, dtype=np.float64, chunk=1 << 16,
            return_latents=False, f32_aggregation=False):
  """x_grid [N_grid, B, C_in] -> [N_grid, B, C_out].

  ``f32_aggregation=True`` reproduces the reference's float32 cast around the grid2mesh
  segment-sum (``graphcast.py:215``); it only changes results when ``dtype`` is float64 and is
  used to match the golden vectors bit-for-bit.  The float64 "truth" for GPU parity leaves it off."""
  x = np.asarray(x_grid, dtype=dtype)
  b = x.shape[1]
  n_mesh = graphs["n_mesh"]
  # grid2mesh (:550-604): grid <- [x | struct], mesh <- [0 | struct]
  enc = gnn.deep_typed_graph_net(
      params, "grid2mesh_gnn",
      {"nodes": {
          "grid_nodes": np.concatenate([x, _batch(graphs["grid_node_feat"], b, dtype)], -1),
          "mesh_nodes": np.concatenate([np.zeros((n_mesh,) + x.shape[1:], dtype),
                                        _batch(graphs["mesh_node_feat"], b, dtype)], -1)},
       "edges": {"grid2mesh": dict(
           senders_set="grid_nodes", receivers_set="mesh_nodes",
           senders=graphs["g2m"]["senders"], receivers=graphs["g2m"]["receivers"],
           features=_batch(graphs["g2m"]["feat"], b, dtype))}},
      num_steps=1, embed_nodes=True, embed_edges=True, dtype=dtype, chunk=chunk,
      live_edges=(), f32_aggregation=f32_aggregation)
  lat_mesh, lat_grid = enc["nodes"]["mesh_nodes"], enc["nodes"]["grid_nodes"]
  # mesh (:606-639)
  proc = gnn.deep_typed_graph_net(
      params, "mesh_gnn",
      {"nodes": {"mesh_nodes": lat_mesh},
       "edges": {"mesh": dict(
           senders_set="mesh_nodes", receivers_set="mesh_nodes",
           senders=graphs["mesh"]["senders"], receivers=graphs["mesh"]["receivers"],
           features=_batch(graphs["mesh"]["feat"], b, dtype))}},
      num_steps=steps, embed_nodes=False, embed_edges=True, dtype=dtype, chunk=chunk,
      live_edges=())
  upd_mesh = proc["nodes"]["mesh_nodes"]
  # mesh2grid (:641-678)
  dec = gnn.deep_typed_graph_net(
      params, "mesh2grid_gnn",
      {"nodes": {"mesh_nodes": upd_mesh, "grid_nodes": lat_grid},
       "edges": {"mesh2grid": dict(
           senders_set="mesh_nodes", receivers_set="grid_nodes",
           senders=graphs["m2g"]["senders"], receivers=graphs["m2g"]["receivers"],
           features=_batch(graphs["m2g"]["feat"], b, dtype))}},
      num_steps=1, embed_nodes=False, embed_edges=True, node_output=("grid_nodes",),
      dtype=dtype, chunk=chunk, live_nodes=("grid_nodes",), live_edges=())
  out = dec["nodes"]["grid_nodes"]
  if return_latents:
    return out, dict(latent_mesh=lat_mesh, latent_grid=lat_grid, updated_mesh=upd_mesh)
  return out
