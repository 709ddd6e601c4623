"""Oracle: structural node/edge features (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``/root/reference/weathernext/utils/model_utils.py`` for the one
configuration GraphCast uses (``graphcast.py:186-193``: no absolute positions,
cos(lat) + cos/sin(lon) node features, receiver-local relative positions with
both latitude and longitude rotations):
  * ``cartesian_to_lat_lon``    <- ``cartesian_to_spherical`` + ``spherical_to_lat_lon`` (:189-206)
  * ``node_features``           <- ``get_graph_spatial_features`` node half (:83-104)
  * ``receiver_local_rotation`` <- ``get_rotation_matrices_to_local_coordinates`` (:322-398)
  * ``edge_features``           <- ``get_bipartite_graph_spatial_features`` edge half
                                   (:511-542) with
                                   ``get_bipartite_relative_position_in_receiver_local_coordinates``
                                   (:547-642); the homogeneous-graph variant (:106-138,237-319)
                                   is the same computation with senders == receivers node set.
"""
import numpy as np
from scipy.spatial import transform


def cartesian_to_lat_lon(xyz):
  """float32 (lat, lon) in degrees of unit vectors, as graphcast.py:383-394."""
  phi = np.arctan2(xyz[:, 1], xyz[:, 0])
  with np.errstate(invalid="ignore"):
    theta = np.arccos(xyz[:, 2])
  lon = np.mod(np.rad2deg(phi), 360)
  lat = 90 - np.rad2deg(theta)
  return lat.astype(np.float32), lon.astype(np.float32)


def _phi_theta(lat, lon):
  return np.deg2rad(lon), np.deg2rad(90 - lat)     # model_utils.py:180-186


def _unit_xyz(phi, theta):
  return np.stack([np.cos(phi) * np.sin(theta),    # model_utils.py:209-216
                   np.sin(phi) * np.sin(theta),
                   np.cos(theta)], axis=-1)


def node_features(lat, lon):
  """[N,3] = [cos(theta), cos(phi), sin(phi)] in the dtype of lat/lon."""
  phi, theta = _phi_theta(lat, lon)
  return np.stack([np.cos(theta), np.cos(phi), np.sin(phi)], axis=-1)


def receiver_local_rotation(phi, theta):
  """[N,3,3] float64: Rz(-phi) then Ry(pi/2 - theta) (scipy euler "zy")."""
  return transform.Rotation.from_euler(
      "zy", np.stack([-phi, -theta + np.pi / 2], axis=1)).as_matrix()


def edge_features(s_lat, s_lon, r_lat, r_lon, senders, receivers,
                  normalization=None):
  """[E,4] float64 = [|d|, dx, dy, dz] / norm; returns (features, norm used)."""
  s_phi, s_theta = _phi_theta(s_lat, s_lon)
  r_phi, r_theta = _phi_theta(r_lat, r_lon)
  s_pos, r_pos = _unit_xyz(s_phi, s_theta), _unit_xyz(r_phi, r_theta)
  rot = receiver_local_rotation(r_phi, r_theta)[receivers]
  apply = lambda m, p: np.einsum("...ji,...i->...j", m, p)   # model_utils.py:401-403
  rel = apply(rot, s_pos[senders]) - apply(rot, r_pos[receivers])
  dist = np.linalg.norm(rel, axis=-1, keepdims=True)
  if normalization is None:
    normalization = dist.max()
  feats = np.concatenate([dist / normalization, rel / normalization], axis=-1)
  return feats, normalization
