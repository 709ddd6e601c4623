"""Pins the oracle's STRUCTURE half against fixtures produced by the reference's
own code (tests/golden/make_golden.py) and against the reference's known-answer
tests (icosahedral_mesh_test.py, grid_mesh_connectivity_test.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import connectivity, features, graphcast as ogc, mesh


def sha(a):
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.fixture(scope="module")
def tiny(golden_dir):
  return dict(np.load(os.path.join(golden_dir, "structure_tiny.npz")))


@pytest.fixture(scope="module")
def hashes(golden_dir):
  with open(os.path.join(golden_dir, "structure_hashes.json")) as f:
    return json.load(f)


def test_icosahedron_counts():
  # icosahedral_mesh_test.py:36-39
  v, f = mesh.icosahedron()
  assert v.shape == (12, 3) and f.shape == (20, 3)
  assert v.dtype == np.float32 and f.dtype == np.int32


def test_hierarchy_properties():
  # icosahedral_mesh_test.py:41-59,94-126
  levels = mesh.mesh_hierarchy(4)
  nv, nf = 12, 20
  for i, (v, f) in enumerate(levels):
    assert v.shape == (nv, 3) and f.shape == (nf, 3)
    np.testing.assert_allclose(np.linalg.norm(v, axis=1), 1.0, rtol=1e-6)
    tri = v[f].astype(np.float64)
    normal = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 1])
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    centre = tri.mean(1)
    centre /= np.linalg.norm(centre, axis=1, keepdims=True)
    np.testing.assert_allclose((normal * centre).sum(1), 1.0, atol=6e-4)
    if i:
      np.testing.assert_array_equal(v[:len(levels[i - 1][0])], levels[i - 1][0])
    nv, nf = nv + 3 * nf // 2, 4 * nf


def test_faces_to_edges_known_answer():
  # icosahedral_mesh_test.py:72-91
  s, r = mesh.faces_to_edges(np.array([[0, 1, 2], [3, 4, 5]]))
  np.testing.assert_array_equal(np.stack([s, r], -1),
                                [[0, 1], [3, 4], [1, 2], [4, 5], [2, 0], [5, 3]])


def test_merge_meshes():
  # icosahedral_mesh_test.py:61-70
  levels = mesh.mesh_hierarchy(2)
  merged = mesh.merged_faces(levels)
  assert merged.shape[0] == sum(f.shape[0] for _, f in levels)
  np.testing.assert_array_equal(merged[:20], levels[0][1])


def test_grid_xyz_known_answer():
  # grid_mesh_connectivity_test.py:23-47
  lat = np.array([-45.0, 0.0, 45.0])
  lon = np.array([0.0, 90.0, 180.0, 270.0])
  inv = 1 / np.sqrt(2)
  want = np.array([
      [[inv, 0, -inv], [0, inv, -inv], [-inv, 0, -inv], [0, -inv, -inv]],
      [[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]],
      [[inv, 0, inv], [0, inv, inv], [-inv, 0, inv], [0, -inv, inv]]])
  np.testing.assert_allclose(connectivity.grid_xyz(lat, lon), want, atol=1e-15)


def test_tiny_structure_matches_reference(tiny):
  g = ogc.build_graphs(tiny["lat"], tiny["lon"], mesh_size=2)
  np.testing.assert_array_equal(g["mesh_vertices"], tiny["mesh_vertices"])
  np.testing.assert_array_equal(g["mesh_faces"], tiny["mesh_faces"])
  assert repr(float(g["radius"])) == repr(float(tiny["radius"]))
  np.testing.assert_array_equal(g["g2m"]["senders"], tiny["g2m_grid_idx"])
  np.testing.assert_array_equal(g["g2m"]["receivers"], tiny["g2m_mesh_idx"])
  np.testing.assert_array_equal(g["mesh"]["senders"], tiny["mesh_senders"])
  np.testing.assert_array_equal(g["mesh"]["receivers"], tiny["mesh_receivers"])
  np.testing.assert_allclose(g["grid_node_feat"], tiny["grid_node_feat"], rtol=0, atol=1e-7)
  np.testing.assert_allclose(g["mesh_node_feat"], tiny["mesh_node_feat"], rtol=0, atol=1e-7)
  np.testing.assert_allclose(g["mesh_node_feat"], tiny["mesh_node_feat_bipartite"], atol=1e-7)
  np.testing.assert_allclose(g["g2m"]["feat"], tiny["g2m_edge_feat"], rtol=0, atol=1e-12)
  np.testing.assert_allclose(g["mesh"]["feat"], tiny["mesh_edge_feat"], rtol=0, atol=1e-12)
  np.testing.assert_allclose(connectivity.grid_xyz(tiny["lat"], tiny["lon"]),
                             tiny["grid_xyz"], atol=1e-7)


def _check_against_hashes(entry, res, mesh_size):
  lat = np.arange(-90, 90 + res / 2, res).astype(np.float32)
  lon = np.arange(0, 360, res).astype(np.float32)
  levels = mesh.mesh_hierarchy(mesh_size)
  v, f = levels[-1]
  assert sha(f) == entry["mesh_faces"]["sha256_16"]
  assert sha(v) == entry["mesh_vertices"]["sha256_16_as_f32"]
  radius = mesh.max_edge_length(v, f) * 0.6
  assert repr(float(radius)) == entry["radius_repr"]
  ms, mr = mesh.faces_to_edges(mesh.merged_faces(levels))
  assert sha(ms) == entry["mesh_senders"]["sha256_16"]
  assert sha(mr) == entry["mesh_receivers"]["sha256_16"]
  gi, mi = connectivity.radius_query(lat, lon, v, radius)
  assert gi.dtype == np.dtype(entry["g2m_grid_idx"]["dtype"])
  assert sha(gi) == entry["g2m_grid_idx"]["sha256_16"]
  assert sha(mi) == entry["g2m_mesh_idx"]["sha256_16"]
  mlat, mlon = features.cartesian_to_lat_lon(v)
  glon, glat = np.meshgrid(lon, lat)
  glat = glat.reshape(-1).astype(np.float32)
  glon = glon.reshape(-1).astype(np.float32)
  ef, _ = features.edge_features(glat, glon, mlat, mlon, gi, mi)
  np.testing.assert_allclose(np.abs(ef).sum(dtype=np.float64),
                             entry["g2m_edge_feat"]["abs_sum_f64"], rtol=1e-9)
  ef, _ = features.edge_features(mlat, mlon, mlat, mlon, ms, mr)
  np.testing.assert_allclose(np.abs(ef).sum(dtype=np.float64),
                             entry["mesh_edge_feat"]["abs_sum_f64"], rtol=1e-9)
  nf = features.node_features(glat, glon)
  np.testing.assert_allclose(np.abs(nf).sum(dtype=np.float64),
                             entry["grid_node_feat"]["abs_sum_f64"], rtol=1e-6)


def test_1deg_structure_fingerprints(hashes):
  _check_against_hashes(hashes["1deg_M5"], 1.0, 5)


@pytest.mark.slow
def test_0p25deg_structure_fingerprints(hashes):
  _check_against_hashes(hashes["0p25deg_M6"], 0.25, 6)


def test_local_coordinate_helpers_match_reference_functions(golden_dir):
  """get_rotation_matrices_to_local_coordinates, rotate_with_matrices,
  get_relative_position_in_receiver_local_coordinates (+ bipartite) against the outputs of the
  reference's own functions (tests/golden/make_golden_rotations.py -> rotation_ref.npz)."""
  import os
  from graphcast_amd import model_utils as mu
  z = np.load(os.path.join(golden_dir, "rotation_ref.npz"))
  for lat, lon in ((True, True), (False, True), (True, False)):
    tag = f"lat{int(lat)}lon{int(lon)}"
    m = mu.get_rotation_matrices_to_local_coordinates(z["phi"], z["theta"], rotate_latitude=lat, rotate_longitude=lon)
    assert m.shape == (9, 3, 3) and m.dtype == np.float64
    np.testing.assert_allclose(m, z[f"mat_{tag}"], atol=1e-15)
    np.testing.assert_allclose(mu.rotate_with_matrices(m, z["pos"]), z[f"rot_{tag}"], atol=1e-14)
    rel = mu.get_relative_position_in_receiver_local_coordinates(
        z["phi"], z["theta"], z["senders"], z["receivers"], latitude_local_coordinates=lat, longitude_local_coordinates=lon)
    np.testing.assert_allclose(rel, z[f"rel_{tag}"], atol=1e-14)
    brel = mu.get_bipartite_relative_position_in_receiver_local_coordinates(
        z["phi"], z["theta"], z["b_send"], z["phi2"], z["theta2"], z["b_recv"],
        latitude_local_coordinates=lat, longitude_local_coordinates=lon)
    np.testing.assert_allclose(brel, z[f"brel_{tag}"], atol=1e-14)
  # the rotated receiver sits at longitude 0 / polar angle pi/2
  m = mu.get_rotation_matrices_to_local_coordinates(z["phi"], z["theta"], rotate_latitude=True, rotate_longitude=True)
  p = np.stack(mu.spherical_to_cartesian(z["phi"].astype(np.float64), z["theta"].astype(np.float64)), axis=-1)
  np.testing.assert_allclose(mu.rotate_with_matrices(m, p), np.tile([1.0, 0.0, 0.0], (9, 1)), atol=1e-6)
  with pytest.raises(ValueError):
    mu.get_rotation_matrices_to_local_coordinates(z["phi"], z["theta"], rotate_latitude=False, rotate_longitude=False)


def test_mesh2grid_product_equals_restated_oracle_hash_1deg(golden_dir):
  """The product's mesh2grid indices at 1 deg / M5 vs the fingerprint of the oracle's
  (tests/golden/make_m2g_hashes.py): two independent restatements of trimesh's rule agree,
  tie points included (63 grid points lie exactly on a mesh edge at this size)."""
  import hashlib
  import json
  import os
  from graphcast_amd import grid_mesh_connectivity as gmc
  from graphcast_amd import icosahedral_mesh as im
  want = json.load(open(os.path.join(golden_dir, "m2g_restated_hashes.json")))["1deg_M5"]
  lat = np.arange(-90, 90.5, 1.0).astype(np.float32)
  lon = np.arange(0, 360, 1.0).astype(np.float32)
  mesh = im.get_hierarchy_of_triangular_meshes_for_sphere(splits=5)[-1]
  grid_idx, mesh_idx = gmc.in_mesh_triangle_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh)
  h16 = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
  assert h16(np.asarray(grid_idx, np.int64)) == want["m2g_grid_idx"]["sha256_16"]
  assert h16(np.asarray(mesh_idx, np.int64)) == want["m2g_mesh_idx"]["sha256_16"]


def test_mesh2grid_face_indices_can_be_injected():
  """Escape hatch for hosts that have trimesh (reference grid_mesh_connectivity.py:114-119):
  a precomputed face per grid point replaces the restated rule, in the product and the oracle."""
  from graphcast_amd import graphcast as gc
  from graphcast_amd import grid_mesh_connectivity as gmc
  from graphcast_amd import icosahedral_mesh as im
  from oracle import graphcast as ogc
  res, mesh_size = 10.0, 2
  lat = np.arange(-90, 90 + res / 2, res).astype(np.float32)     # (the Predictor casts to float32)
  lon = np.arange(0, 360, res).astype(np.float32)
  mesh = im.get_hierarchy_of_triangular_meshes_for_sphere(splits=mesh_size)[-1]
  own = gmc._nearest_face_on_surface(gmc._grid_lat_lon_to_coordinates(lat, lon).reshape([-1, 3]), mesh)
  # same faces injected -> same graph; a different (valid) choice shows up in the edges
  g0, m0 = gmc.in_mesh_triangle_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh)
  g1, m1 = gmc.in_mesh_triangle_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh,
                                        query_face_indices=own)
  np.testing.assert_array_equal(m0, m1)
  np.testing.assert_array_equal(g0, g1)
  other = own.copy()
  other[5] = (own[5] + 1) % len(mesh.faces)
  _, m2 = gmc.in_mesh_triangle_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh,
                                       query_face_indices=other)
  np.testing.assert_array_equal(m2[15:18], mesh.faces[other[5]])
  np.testing.assert_array_equal(np.delete(m2, [15, 16, 17]), np.delete(m0, [15, 16, 17]))
  with pytest.raises(ValueError):
    gmc.in_mesh_triangle_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh,
                                 query_face_indices=own[:-1])
  with pytest.raises(ValueError):
    gmc.in_mesh_triangle_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh,
                                 query_face_indices=np.full_like(own, len(mesh.faces)))
  # the Predictor and the oracle take the same array
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=1,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  model = gc.GraphCast(cfg, gc.TASK_13, mesh2grid_face_indices=other).init_from_coordinates(lat, lon)
  want = ogc.build_graphs(lat, lon, mesh_size, m2g_face_indices=other)
  got = model.graph_arrays()
  np.testing.assert_array_equal(got["m2g"]["senders"], want["m2g"]["senders"])
  np.testing.assert_array_equal(got["m2g"]["receivers"], want["m2g"]["receivers"])
  np.testing.assert_allclose(got["m2g"]["feat"], want["m2g"]["feat"], atol=1e-12)
