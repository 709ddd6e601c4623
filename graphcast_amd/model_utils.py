"""Structural (lat/lon-derived) node and edge features + Dataset <-> [node, batch, channel] stacking.

Host side.  Same public names, argument meaning and results as the parts of the
reference's ``weathernext/utils/model_utils.py`` that GraphCast uses
(``get_graph_spatial_features`` :29-152, ``get_bipartite_graph_spatial_features``
:406-544, the spherical helpers :180-234, the stacking helpers :155-177,645-776).
The receiver-local rotation is written in closed form (R = Ry(pi/2 - theta) .
Rz(-phi)) instead of going through scipy Rotation objects and per-edge 3x3
matrix gathers: only the rotated difference is ever formed.
"""
from typing import Mapping, Optional, Tuple

import numpy as np


def lat_lon_deg_to_spherical(node_lat, node_lon):
  return np.deg2rad(node_lon), np.deg2rad(90 - node_lat)


def spherical_to_lat_lon(phi, theta):
  return 90 - np.rad2deg(theta), np.mod(np.rad2deg(phi), 360)


def cartesian_to_spherical(x, y, z):
  with np.errstate(invalid="ignore"):
    return np.arctan2(y, x), np.arccos(z)


def spherical_to_cartesian(phi, theta):
  return np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta)


def lat_lon_to_cartesian(lat, lon):
  return spherical_to_cartesian(*lat_lon_deg_to_spherical(lat, lon))


def cartesian_to_lat_lon(x, y, z):
  return spherical_to_lat_lon(*cartesian_to_spherical(x, y, z))


def _node_features(phi, theta, num_nodes, dtype, add_node_positions, add_node_latitude,
                   add_node_longitude):
  feats = []
  if add_node_positions:
    feats.extend(spherical_to_cartesian(phi, theta))
  if add_node_latitude:
    feats.append(np.cos(theta))
  if add_node_longitude:
    feats.append(np.cos(phi))
    feats.append(np.sin(phi))
  if not feats:
    return np.zeros([num_nodes, 0], dtype=dtype)
  return np.stack(feats, axis=-1)


def get_rotation_matrices_to_local_coordinates(reference_phi: np.ndarray, reference_theta: np.ndarray,
                                               rotate_latitude: bool, rotate_longitude: bool) -> np.ndarray:
  """[N, 3, 3] float64 matrices M such that ``rotate_with_matrices(M, p)`` takes the reference
  point to longitude 0 (``rotate_longitude``) and / or latitude 0, i.e. polar angle pi/2
  (``rotate_latitude``) -- reference model_utils.py:322-395.  With a = -phi and b = pi/2 - theta:
    longitude only       p -> Rz(a) p
    both                 p -> Ry(b) Rz(a) p
    latitude only        p -> Rz(-a) Ry(b) Rz(a) p   (keeps the point's own longitude)
  Built from the closed forms of Rz / Ry (the reference asks scipy for the same rotations; the
  angles are promoted to float64 exactly as scipy does)."""
  if not (rotate_latitude or rotate_longitude):
    raise ValueError("At least one of longitude and latitude should be rotated.")
  a = (-np.asarray(reference_phi)).astype(np.float64)
  b = (-np.asarray(reference_theta) + np.pi / 2).astype(np.float64)
  zero, one = np.zeros_like(a), np.ones_like(a)

  def rz(t):
    c, s = np.cos(t), np.sin(t)
    return np.stack([np.stack([c, -s, zero], -1), np.stack([s, c, zero], -1), np.stack([zero, zero, one], -1)], -2)

  def ry(t):
    c, s = np.cos(t), np.sin(t)
    return np.stack([np.stack([c, zero, s], -1), np.stack([zero, one, zero], -1), np.stack([-s, zero, c], -1)], -2)

  if rotate_longitude and rotate_latitude:
    forward = ry(b) @ rz(a)
  elif rotate_longitude:
    forward = rz(a)
  else:
    forward = rz(-a) @ ry(b) @ rz(a)
  return forward


def rotate_with_matrices(rotation_matrices: np.ndarray, positions: np.ndarray) -> np.ndarray:
  """Batched M p: out[..., j] = sum_i M[..., j, i] p[..., i] (reference model_utils.py:401-403)."""
  return np.einsum("...ji,...i->...j", rotation_matrices, positions)


def get_relative_position_in_receiver_local_coordinates(
    node_phi: np.ndarray, node_theta: np.ndarray, senders: np.ndarray, receivers: np.ndarray,
    latitude_local_coordinates: bool, longitude_local_coordinates: bool) -> np.ndarray:
  """[E, 3] sender - receiver positions, each edge in its receiver's rotated frame
  (reference model_utils.py:237-319)."""
  return get_bipartite_relative_position_in_receiver_local_coordinates(
      node_phi, node_theta, senders, node_phi, node_theta, receivers,
      latitude_local_coordinates, longitude_local_coordinates)


def get_bipartite_relative_position_in_receiver_local_coordinates(
    senders_node_phi: np.ndarray, senders_node_theta: np.ndarray, senders: np.ndarray,
    receivers_node_phi: np.ndarray, receivers_node_theta: np.ndarray, receivers: np.ndarray,
    latitude_local_coordinates: bool, longitude_local_coordinates: bool) -> np.ndarray:
  """Bipartite form (reference model_utils.py:547-642): sender and receiver nodes come from
  different sets; the frame is always the receiver's."""
  s_pos = np.stack(spherical_to_cartesian(senders_node_phi, senders_node_theta), axis=-1)
  r_pos = np.stack(spherical_to_cartesian(receivers_node_phi, receivers_node_theta), axis=-1)
  if not (latitude_local_coordinates or longitude_local_coordinates):
    return s_pos[senders] - r_pos[receivers]
  # one matrix per RECEIVER node, gathered per edge
  m = get_rotation_matrices_to_local_coordinates(
      receivers_node_phi, receivers_node_theta, rotate_latitude=latitude_local_coordinates,
      rotate_longitude=longitude_local_coordinates)[receivers]
  return (rotate_with_matrices(m, s_pos[senders].astype(np.float64))
          - rotate_with_matrices(m, r_pos[receivers].astype(np.float64)))


def _relative_positions(s_phi, s_theta, r_phi, r_theta, senders, receivers,
                        latitude_local_coordinates, longitude_local_coordinates):
  return get_bipartite_relative_position_in_receiver_local_coordinates(
      s_phi, s_theta, senders, r_phi, r_theta, receivers,
      latitude_local_coordinates, longitude_local_coordinates)


def _edge_features(relative_position, num_edges, dtype, edge_normalization_factor):
  distances = np.linalg.norm(relative_position, axis=-1, keepdims=True)
  if edge_normalization_factor is None:
    edge_normalization_factor = distances.max()
  return np.concatenate([distances / edge_normalization_factor,
                         relative_position / edge_normalization_factor], axis=-1)


def get_graph_spatial_features(
    *, node_lat: np.ndarray, node_lon: np.ndarray, senders: np.ndarray, receivers: np.ndarray,
    add_node_positions: bool, add_node_latitude: bool, add_node_longitude: bool,
    add_relative_positions: bool, edge_normalization_factor: Optional[float] = None,
    relative_longitude_local_coordinates: bool, relative_latitude_local_coordinates: bool,
    ) -> Tuple[np.ndarray, np.ndarray]:
  """Node features [N, F_n] (node dtype) and edge features [E, F_e] (float64)."""
  s_feat, _, e_feat = get_bipartite_graph_spatial_features(
      senders_node_lat=node_lat, senders_node_lon=node_lon, senders=senders,
      receivers_node_lat=node_lat, receivers_node_lon=node_lon, receivers=receivers,
      add_node_positions=add_node_positions, add_node_latitude=add_node_latitude,
      add_node_longitude=add_node_longitude, add_relative_positions=add_relative_positions,
      edge_normalization_factor=edge_normalization_factor,
      relative_longitude_local_coordinates=relative_longitude_local_coordinates,
      relative_latitude_local_coordinates=relative_latitude_local_coordinates)
  return s_feat, e_feat


def get_bipartite_graph_spatial_features(
    *, senders_node_lat: np.ndarray, senders_node_lon: np.ndarray, senders: np.ndarray,
    receivers_node_lat: np.ndarray, receivers_node_lon: np.ndarray, receivers: np.ndarray,
    add_node_positions: bool, add_node_latitude: bool, add_node_longitude: bool,
    add_relative_positions: bool, edge_normalization_factor: Optional[float] = None,
    relative_longitude_local_coordinates: bool, relative_latitude_local_coordinates: bool,
    ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
  """(sender node features, receiver node features, edge features)."""
  dtype = senders_node_lat.dtype
  assert receivers_node_lat.dtype == dtype
  s_phi, s_theta = lat_lon_deg_to_spherical(senders_node_lat, senders_node_lon)
  r_phi, r_theta = lat_lon_deg_to_spherical(receivers_node_lat, receivers_node_lon)
  flags = (add_node_positions, add_node_latitude, add_node_longitude)
  s_feat = _node_features(s_phi, s_theta, senders_node_lat.shape[0], dtype, *flags)
  r_feat = _node_features(r_phi, r_theta, receivers_node_lat.shape[0], dtype, *flags)
  if add_relative_positions:
    rel = _relative_positions(s_phi, s_theta, r_phi, r_theta, senders, receivers,
                              relative_latitude_local_coordinates,
                              relative_longitude_local_coordinates)
    e_feat = _edge_features(rel, senders.shape[0], dtype, edge_normalization_factor)
  else:
    e_feat = np.zeros([senders.shape[0], 0], dtype=dtype)
  return s_feat, r_feat, e_feat


# ----------------------------------------------------------------------------- Dataset <-> stacked
_PRESERVED = ("batch", "lat", "lon")


def lat_lon_to_leading_axes(grid_xarray):
  """[...] + (lat, lon) + [...] -> (lat, lon, ...)  (reference :155-161)."""
  return grid_xarray.transpose("lat", "lon", ...)


def restore_leading_axes(grid_xarray):
  """(lat, lon, [batch, time, level], ...) -> ([batch, time, level], lat, lon, ...)  (reference :164-177)."""
  dims = list(grid_xarray.dims)
  front = [d for d in ("batch", "time", "level") if d in dims]
  return grid_xarray.transpose(*(front + [d for d in dims if d not in front]))


def variable_to_stacked(variable, sizes: Mapping[str, int],
                        preserved_dims: Tuple[str, ...] = _PRESERVED):
  """Variable -> preserved_dims + ("channels",)  (reference :645-674).

  Every dim outside ``preserved_dims`` is folded (C order, i.e. time-major then level)
  into a trailing ``channels`` dim; missing preserved dims are broadcast to ``sizes``."""
  folded = [d for d in variable.dims if d not in preserved_dims]
  if folded:
    variable = variable.stack(channels=folded)
  out_sizes = {d: variable.sizes.get(d) or sizes[d] for d in preserved_dims}
  out_sizes["channels"] = variable.sizes.get("channels", 1)
  return variable.set_dims(out_sizes)


def dataset_to_stacked(dataset, sizes: Optional[Mapping[str, int]] = None,
                       preserved_dims: Tuple[str, ...] = _PRESERVED):
  """Dataset -> one DataArray preserved_dims + ("channels",), variables in sorted-name
  order (reference :677-710)."""
  from graphcast_amd import xarray_lite as xr
  sizes = sizes or dataset.sizes
  variables = dataset.variables
  stacked = [variable_to_stacked(variables[name], sizes, preserved_dims)
             for name in sorted(dataset.data_vars.keys())]
  coords = {d: c for d, c in dataset.coords.items() if d in preserved_dims}
  return xr.DataArray(data=xr.Variable.concat(stacked, dim="channels"), coords=coords)


def stacked_to_dataset(stacked_array, template_dataset,
                       preserved_dims: Tuple[str, ...] = _PRESERVED):
  """Inverse of ``dataset_to_stacked`` given a template (reference :713-776)."""
  from graphcast_amd import xarray_lite as xr
  names = sorted(template_dataset.keys())
  unstack_sizes = {}
  for name in names:
    tv = template_dataset[name]
    if not all(d in tv.dims for d in preserved_dims):
      raise ValueError(
          f"stacked_to_dataset requires all Variables to have {preserved_dims} "
          f"dimensions, but found only {tv.dims}.")
    unstack_sizes[name] = {d: n for d, n in tv.sizes.items() if d not in preserved_dims}
  channels = {name: int(np.prod(list(s.values()), dtype=np.int64))
              for name, s in unstack_sizes.items()}
  expected, found = sum(channels.values()), stacked_array.sizes["channels"]
  if expected != found:
    raise ValueError(
        f"Expected {expected} channels but found {found}, when trying to convert a stacked "
        f"array of shape {stacked_array.sizes} to a dataset of shape {template_dataset}.")
  data_vars, start = {}, 0
  for name in names:
    tv = template_dataset[name]
    piece = stacked_array.isel({"channels": slice(start, start + channels[name])})
    start += channels[name]
    piece = piece.unstack({"channels": unstack_sizes[name]}).transpose(*tv.dims)
    data_vars[name] = xr.DataArray(data=piece, coords=dict(tv._coords), name=tv.name)
  return type(template_dataset)(data_vars)
