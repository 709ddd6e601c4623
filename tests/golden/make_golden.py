"""Generates tests/golden/*.npz|json by executing the REFERENCE's own modules.

Run in the build container only (needs /root/reference; it does not exist on
the GPU box, which is why the outputs are committed):

    python tests/golden/make_golden.py

What runs unmodified from /root/reference/weathernext (numpy/scipy only):
  utils/icosahedral_mesh.py, utils/legacy/grid_mesh_connectivity.py
  (radius_query_indices, _grid_lat_lon_to_coordinates), utils/model_utils.py
  (structural-feature functions).  ``jax``, ``xarray``, ``trimesh`` ... are
  absent from this image; inert stand-ins are placed in ``sys.modules`` so the
  ``import`` lines succeed -- none of the executed functions touch them.

Part 2 executes the reference's GNN sources UNMODIFIED --
  weathernext1_graph/graphcast.py (GraphCast.__init__, _init_*_graph, _run_grid2mesh_gnn,
  _run_mesh_gnn, _run_mesh2grid_gnn), utils/legacy/deep_typed_graph_net.py,
  utils/typed_graph_net.py, utils/typed_graph.py -- on the numpy stand-ins for
  haiku / jraph / jax / chex in tests/golden/ref_shims (see its README: this pins the
  reference's WIRING and parameter tree; the primitives are restated from the libraries'
  published behaviour).  trimesh is absent, so grid_mesh_connectivity.in_mesh_triangle_indices
  is replaced by the oracle's containing-triangle query for this run (its output is part
  of the fixture).

Outputs
  structure_tiny.npz   every structure array for a 10 deg grid / M2 mesh
  structure_hashes.json  sha256 fingerprints (+ counts) for 1 deg/M5 and 0.25 deg/M6
  gnn_tiny.npz         10 deg / M2, latent 32, 2 processor steps, batch 2: parameter tree
                       ("params:<module>:<leaf>"), input x, the three stage outputs, float64
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _install_inert_modules():
  class _Anything(types.ModuleType):
    def __getattr__(self, name):
      if name.startswith("__"):
        raise AttributeError(name)
      sub = _Anything(f"{self.__name__}.{name}")
      setattr(self, name, sub)
      return sub

    def __call__(self, *a, **k):
      raise RuntimeError(f"inert stand-in {self.__name__} was called")

  for name in ("jax", "jax.numpy", "jax.tree_util", "xarray", "xarray.ufuncs",
               "trimesh", "chex", "haiku", "jraph", "tree", "xarray_jax",
               "dask", "dask.array"):
    if name not in sys.modules:
      sys.modules[name] = _Anything(name)


def sha(a):
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def reference_structure(res, mesh_size, fraction=0.6):
  from weathernext.utils import icosahedral_mesh as im
  from weathernext.utils import model_utils as mu
  from weathernext.utils.legacy import grid_mesh_connectivity as gmc
  kw = dict(add_node_positions=False, add_node_latitude=True, add_node_longitude=True,
            add_relative_positions=True, relative_longitude_local_coordinates=True,
            relative_latitude_local_coordinates=True)           # graphcast.py:186-193
  lat = np.arange(-90, 90 + res / 2, res).astype(np.float32)
  lon = np.arange(0, 360, res).astype(np.float32)
  meshes = im.get_hierarchy_of_triangular_meshes_for_sphere(splits=mesh_size)
  finest = meshes[-1]
  s, r = im.faces_to_edges(finest.faces)
  radius = np.linalg.norm(finest.vertices[s] - finest.vertices[r], axis=-1).max() * fraction
  phi, theta = mu.cartesian_to_spherical(finest.vertices[:, 0], finest.vertices[:, 1],
                                         finest.vertices[:, 2])
  mlat, mlon = mu.spherical_to_lat_lon(phi=phi, theta=theta)
  mlat, mlon = mlat.astype(np.float32), mlon.astype(np.float32)
  glon, glat = np.meshgrid(lon, lat)
  glon = glon.reshape([-1]).astype(np.float32)
  glat = glat.reshape([-1]).astype(np.float32)
  gi, mi = gmc.radius_query_indices(grid_latitude=lat, grid_longitude=lon,
                                    mesh=finest, radius=radius)
  gnf, mnf, g2m_ef = mu.get_bipartite_graph_spatial_features(
      senders_node_lat=glat, senders_node_lon=glon, receivers_node_lat=mlat,
      receivers_node_lon=mlon, senders=gi, receivers=mi,
      edge_normalization_factor=None, **kw)
  merged = im.merge_meshes(meshes)
  ms, mr = im.faces_to_edges(merged.faces)
  mesh_nf, mesh_ef = mu.get_graph_spatial_features(
      node_lat=mlat, node_lon=mlon, senders=ms, receivers=mr, **kw)
  return dict(lat=lat, lon=lon, radius=np.asarray(radius),
              mesh_vertices=finest.vertices, mesh_faces=finest.faces,
              mesh_lat=mlat, mlon=mlon,
              g2m_grid_idx=gi, g2m_mesh_idx=mi, grid_node_feat=gnf,
              mesh_node_feat_bipartite=mnf, g2m_edge_feat=g2m_ef,
              mesh_senders=ms, mesh_receivers=mr, mesh_node_feat=mesh_nf,
              mesh_edge_feat=mesh_ef,
              grid_xyz=gmc._grid_lat_lon_to_coordinates(lat, lon))


def params_digest(params):
  """sha256 over the sorted parameter leaves (detects drift of a seed-regenerated tree)."""
  h = hashlib.sha256()
  for mod in sorted(params):
    for leaf in sorted(params[mod]):
      h.update(f"{mod}:{leaf}".encode())
      h.update(np.ascontiguousarray(params[mod][leaf], dtype=np.float32).tobytes())
  return h.hexdigest()


def reference_gnn(res=10.0, mesh_size=2, latent=32, steps=2, batch=2, c_in=17, seed=5,
                  params=None):
  """Runs the reference GraphCast's three DeepTypedGraphNets on the haiku/jraph/jax stand-ins."""
  import typing
  import typing_extensions
  for n in ("Required", "NotRequired"):          # reference targets python >= 3.11
    if not hasattr(typing, n):
      setattr(typing, n, getattr(typing_extensions, n))
  if not getattr(reference_gnn, "_shims_installed", False):
    for name in [m for m in sys.modules if m.split(".")[0] in ("jax", "haiku", "jraph", "chex")]:
      del sys.modules[name]                      # drop part 1's inert stand-ins
    sys.path.insert(0, os.path.join(HERE, "ref_shims"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root, for `oracle`
    for name in [m for m in sys.modules if m.startswith("weathernext.") and
                 m.split(".")[-1] in ("typed_graph_net", "deep_typed_graph_net", "dense",
                                      "model_utils")]:
      del sys.modules[name]                      # re-import on top of the numpy stand-ins
    reference_gnn._shims_installed = True
  import haiku as hk
  from weathernext.utils.legacy import grid_mesh_connectivity as gmc
  from weathernext.weathernext1_graph import graphcast as rgc
  from oracle import connectivity as oconn

  def m2g(*, grid_latitude, grid_longitude, mesh):
    return oconn.containing_triangle_query(grid_latitude, grid_longitude, mesh.vertices, mesh.faces)
  gmc.in_mesh_triangle_indices = m2g
  rgc.grid_mesh_connectivity.in_mesh_triangle_indices = m2g

  lat = np.arange(-90, 90 + res / 2, res).astype(np.float32)
  lon = np.arange(0, 360, res).astype(np.float32)
  cfg = rgc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=latent,
                        gnn_msg_steps=steps, hidden_layers=1,
                        radius_query_fraction_edge_length=0.6)
  rng = np.random.default_rng(seed)
  x = rng.standard_normal((len(lat) * len(lon), batch, c_in)).astype(np.float32).astype(np.float64)
  given = params is not None
  params = params if given else {}

  def run(model):
    lm, lg = model._run_grid2mesh_gnn(x)
    um = model._run_mesh_gnn(lm)
    return lm, lg, um, model._run_mesh2grid_gnn(um, lg)

  def build():
    model = rgc.GraphCast(cfg, rgc.TASK_13)
    model._init_mesh_properties()
    model._init_grid_properties(grid_lat=lat, grid_lon=lon)
    model._grid2mesh_graph_structure = model._init_grid2mesh_graph()
    model._mesh_graph_structure = model._init_mesh_graph()
    model._mesh2grid_graph_structure = model._init_mesh2grid_graph()
    return model

  if not given:
    with hk.running(params, init_rng=rng):       # creates the parameter tree (haiku-default init)
      run(build())
    for leaves in params.values():               # make b / scale / offset non-trivial
      for k in leaves:
        if k in ("b", "offset"):
          leaves[k] = (0.1 * rng.standard_normal(leaves[k].shape)).astype(np.float32)
        elif k == "scale":
          leaves[k] = (1.0 + 0.1 * rng.standard_normal(leaves[k].shape)).astype(np.float32)
  with hk.running(params):                       # no rng: every parameter must already exist
    model = build()
    lm, lg, um, out = run(model)
  g2m = model._grid2mesh_graph_structure.edge_by_name("grid2mesh")
  m2g_e = model._mesh2grid_graph_structure.edge_by_name("mesh2grid")
  fx = ({} if given else
        {f"params:{mod}:{leaf}": v for mod, leaves in params.items() for leaf, v in leaves.items()})
  fx.update(lat=lat, lon=lon, x=x, latent_mesh=lm, latent_grid=lg, updated_mesh=um, out=out,
            g2m_senders=g2m.indices.senders, g2m_receivers=g2m.indices.receivers,
            m2g_senders=m2g_e.indices.senders, m2g_receivers=m2g_e.indices.receivers,
            config=np.array([res, mesh_size, latent, steps, batch, c_in], dtype=np.float64))
  return fx


def main():
  _install_inert_modules()
  sys.path.insert(0, REF)
  tiny = reference_structure(10.0, 2)
  np.savez_compressed(os.path.join(HERE, "structure_tiny.npz"), **tiny)
  hashes = {}
  for tag, res, m in (("1deg_M5", 1.0, 5), ("0p25deg_M6", 0.25, 6)):
    st = reference_structure(res, m)
    entry = {"radius_repr": repr(float(st["radius"])),
             "radius_dtype": str(st["radius"].dtype)}
    for k, v in st.items():
      entry[k] = {"shape": list(v.shape), "dtype": str(v.dtype)}
      if v.dtype.kind in "iu":
        entry[k]["sha256_16"] = sha(v)           # bit-exact targets
      else:
        # libm-dependent: keep float32-rounded hash + robust statistics.
        entry[k]["sha256_16_as_f32"] = sha(v.astype(np.float32))
        entry[k]["sum_f64"] = float(np.sum(v, dtype=np.float64))
        entry[k]["abs_sum_f64"] = float(np.sum(np.abs(v), dtype=np.float64))
    hashes[tag] = entry
    print(tag, {k: entry[k]["shape"] for k in st})
  with open(os.path.join(HERE, "structure_hashes.json"), "w") as f:
    json.dump(hashes, f, indent=1, sort_keys=True)
  if "--skip-gnn" not in sys.argv:
    fx = reference_gnn()
    np.savez_compressed(os.path.join(HERE, "gnn_tiny.npz"), **fx)
    print("gnn_tiny:", fx["out"].shape, len([k for k in fx if k.startswith("params:")]), "leaves")
    # latent 512 (what the HIP kernels are built for): the 8.7 M parameters are regenerated
    # from a seed by oracle.params.init_params on both sides; their digest is in the fixture.
    from oracle import params as oparams
    c_in, c_out, steps, seed = 183, 83, 2, 1
    p512 = oparams.init_params(c_in, c_out, 512, steps, seed=seed, nontrivial=True)
    fx = reference_gnn(latent=512, steps=steps, batch=1, c_in=c_in, seed=11, params=p512)
    keep = {k: fx[k] for k in ("lat", "lon", "out", "config")}
    keep["x"] = fx["x"].astype(np.float32)
    keep["latent_mesh_checksum"] = np.array([fx["latent_mesh"].sum(), np.abs(fx["updated_mesh"]).sum()])
    keep["params_seed"] = np.array([c_in, c_out, 512, steps, seed])
    keep["params_sha256"] = np.array(params_digest(p512))
    np.savez_compressed(os.path.join(HERE, "gnn_latent512.npz"), **keep)
    print("gnn_latent512:", keep["out"].shape, str(keep["params_sha256"])[:16])


if __name__ == "__main__":
  main()
