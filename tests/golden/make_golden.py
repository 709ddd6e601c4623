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

Outputs
  structure_tiny.npz   every structure array for a 10 deg grid / M2 mesh
  structure_hashes.json  sha256 fingerprints (+ counts) for 1 deg/M5 and 0.25 deg/M6
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


if __name__ == "__main__":
  main()
