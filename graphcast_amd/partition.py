"""Spatial partition of the three GraphCast graphs across GPUs (BASELINE.json config 5).

Nodes (grid nodes and mesh nodes) are split into P compact regions (hemispheres / quadrants /
OCTANTS by the signs of the unit-sphere coordinates for P = 2 / 4 / 8, equal-count longitude bands
otherwise); every edge belongs
to the **owner of its receiver**, so the receiver aggregation (``jraph.segment_sum``,
``typed_graph_net.py:532-538``) is local to a rank -- the "shard-local => drop the collective"
case of the reference's own sharded ops (``gather_scatter_ops.py:102-144``).  What a rank needs
from others are *sender* rows only:

  encoder   : ``(h_grid . W_s)`` rows of grid nodes that send into its mesh nodes     (1 exchange)
  processor : ``(h_mesh . W_s)`` rows of remote mesh senders, every message-passing step (16)
  decoder   : ``(h_mesh . W_s)`` rows of remote mesh senders of its grid nodes          (1)

i.e. 18 halo exchanges per 6-h step of 512-float rows.  Measured at 0.25 deg / 8 parts (this plan,
round 3): octants 393-466 remote sender rows per rank and processor step (mean 419 = 0.86 MB; the
figures SURVEY.md 8e predicted), 770-5,540 grid rows once for the encoder, 200-278 mesh rows once
for the decoder; longitude bands 522-592 / 3,483-7,768 / 266-339.  Latency-bound, so each exchange
is ONE ``all_to_all_single`` with precomputed split sizes.

Local index space of a rank, per node set: ``[owned nodes (global order) | halo nodes (grouped by
owner rank, global order within a group)]``.  Kernels run over the owned prefix; an exchange
fills the halo suffix.  This module is pure numpy (plan) + a small exchanger over
``torch.distributed`` (nccl = RCCL on GPUs, gloo on CPU in the tests) or, for single-process
emulation of P ranks on one GPU, plain tensor copies.
"""
from typing import List, NamedTuple, Sequence

import numpy as np


class HaloPlan(NamedTuple):
  """Remote sender rows of one edge set for one rank."""
  halo_global: np.ndarray        # [H] global ids of the halo nodes, grouped by owner rank
  recv_counts: np.ndarray        # [P] rows received from each rank (sum = H)
  send_local: List[np.ndarray]   # per destination rank: LOCAL owned indices to send (ascending global)


class RankGraphs(NamedTuple):
  """What one rank builds its engine from: local graphs + exchange plans."""
  rank: int
  graphs: dict                   # same layout as GraphCast.graph_arrays(), in local indices
  grid_owned: np.ndarray         # global grid ids owned (ascending) -> rows of x / y it handles
  mesh_owned: np.ndarray
  n_grid_owned: int
  n_mesh_owned: int
  halo_g2m: HaloPlan             # grid rows needed by the encoder
  halo_mesh: HaloPlan            # mesh rows needed by every processor step
  halo_m2g: HaloPlan             # mesh rows needed by the decoder


def owner_by_longitude(lon_deg: np.ndarray, n_parts: int) -> np.ndarray:
  """Equal-count longitude bands (ties broken by index): rank of each node."""
  lon = np.mod(np.asarray(lon_deg, dtype=np.float64), 360.0)
  order = np.lexsort((np.arange(len(lon)), lon))
  owner = np.empty(len(lon), dtype=np.int32)
  bounds = np.linspace(0, len(lon), n_parts + 1).astype(np.int64)
  for p in range(n_parts):
    owner[order[bounds[p]:bounds[p + 1]]] = p
  return owner


def owner_by_octant(lat_deg: np.ndarray, lon_deg: np.ndarray, n_parts: int) -> np.ndarray:
  """Compact equal-area regions for n_parts in {1, 2, 4, 8}: the part of a node is read off the signs
  of its unit-sphere coordinates -- (x) hemispheres, (x, y) quadrants (lunes), (x, y, z) octants --
  the partition SURVEY.md section 8e measured (octants: 2.65 % of the multi-mesh edges cross parts,
  393-466 remote sender rows per rank and processor step at 0.25 deg, against 520-590 for the eight
  longitude bands, whose slivers all meet at both poles).  Nodes exactly on a dividing plane go to the
  non-negative side, so grid and mesh nodes at the same place agree."""
  if n_parts not in (1, 2, 4, 8):
    raise ValueError("sign-based regions exist for 1, 2, 4 or 8 parts")
  lat = np.radians(np.asarray(lat_deg, dtype=np.float64))
  lon = np.radians(np.asarray(lon_deg, dtype=np.float64))
  x, y, z = np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)
  tiny = 1e-12                                          # (cos(90 deg) is 6e-17, not 0: snap it)
  bits = [(v < -tiny).astype(np.int32) for v in (x, y, z)]
  owner = np.zeros(len(lat), dtype=np.int32)
  for k in range({1: 0, 2: 1, 4: 2, 8: 3}[n_parts]):
    owner |= bits[k] << k
  return owner


def _halo(senders, receivers, send_owner, recv_owner, rank, n_parts):
  """Remote senders of the edges whose receiver `rank` owns."""
  mine = recv_owner[receivers] == rank
  snd = np.unique(senders[mine])
  remote = snd[send_owner[snd] != rank]
  groups = [remote[send_owner[remote] == q] for q in range(n_parts)]      # ascending within a group
  halo_global = np.concatenate(groups) if len(remote) else np.zeros(0, dtype=np.int64)
  return halo_global.astype(np.int64), np.array([len(g) for g in groups], dtype=np.int64)


def plan(graphs: dict, grid_lon: np.ndarray, mesh_lon: np.ndarray, n_parts: int, *,
         grid_lat: np.ndarray = None, mesh_lat: np.ndarray = None) -> List[RankGraphs]:
  """Splits ``GraphCast.graph_arrays()`` into per-rank local graphs + halo plans.

  ``grid_lon`` / ``mesh_lon``: longitude (degrees) of every grid / mesh node.  With the latitudes
  given as well and 2, 4 or 8 parts the regions are hemispheres / quadrants / OCTANTS
  (``owner_by_octant``); otherwise equal-count longitude bands."""
  n_grid, n_mesh = int(graphs["n_grid"]), int(graphs["n_mesh"])
  if grid_lat is not None and mesh_lat is not None and n_parts in (2, 4, 8):
    g_owner = owner_by_octant(grid_lat, grid_lon, n_parts)
    m_owner = owner_by_octant(mesh_lat, mesh_lon, n_parts)
  else:
    g_owner = owner_by_longitude(grid_lon, n_parts)
    m_owner = owner_by_longitude(mesh_lon, n_parts)
  g_owned = [np.flatnonzero(g_owner == p) for p in range(n_parts)]
  m_owned = [np.flatnonzero(m_owner == p) for p in range(n_parts)]
  g2m, mesh, m2g = graphs["g2m"], graphs["mesh"], graphs["m2g"]
  as64 = lambda a: np.asarray(a).astype(np.int64)
  es = dict(g2m=(as64(g2m["senders"]), as64(g2m["receivers"]), g_owner, m_owner),
            mesh=(as64(mesh["senders"]), as64(mesh["receivers"]), m_owner, m_owner),
            m2g=(as64(m2g["senders"]), as64(m2g["receivers"]), m_owner, g_owner))
  halos = {k: [_halo(s, r, so, ro, p, n_parts) for p in range(n_parts)]
           for k, (s, r, so, ro) in es.items()}

  def local_map(owned, halo_global, n):
    m = np.full(n, -1, dtype=np.int64)
    m[owned] = np.arange(len(owned))
    m[halo_global] = len(owned) + np.arange(len(halo_global))
    return m

  def send_lists(key, owned_by_rank, src_rank):
    """For source rank `src_rank`: local owned indices to send to every destination."""
    out = []
    pos = np.full(max(n_grid, n_mesh), -1, dtype=np.int64)
    pos[owned_by_rank[src_rank]] = np.arange(len(owned_by_rank[src_rank]))
    for dst in range(n_parts):
      hg, counts = halos[key][dst]
      start = int(counts[:src_rank].sum())
      ids = hg[start:start + int(counts[src_rank])]
      out.append(pos[ids])
    return out

  ranks = []
  for p in range(n_parts):
    # node index spaces differ per edge set (the halo of the encoder's grid senders is not the
    # halo of anything else): each edge set gets its own sender-side local map
    def edge_set(key, feat, n_send, n_recv, send_owned, recv_owned):
      s, r, so, ro = es[key]
      mine = ro[r] == p
      hg, _ = halos[key][p]
      smap = local_map(send_owned[p], hg, n_send)
      rmap = np.full(n_recv, -1, dtype=np.int64)
      rmap[recv_owned[p]] = np.arange(len(recv_owned[p]))
      ls, lr = smap[s[mine]], rmap[r[mine]]
      assert (ls >= 0).all() and (lr >= 0).all()
      return dict(senders=ls, receivers=lr, feat=np.asarray(feat)[mine], edge_ids=np.flatnonzero(mine))

    local = dict(
        n_grid=len(g_owned[p]), n_mesh=len(m_owned[p]), radius=graphs.get("radius"),
        n_grid_senders=len(g_owned[p]) + len(halos["g2m"][p][0]),
        n_mesh_senders=len(m_owned[p]) + len(halos["mesh"][p][0]),
        n_mesh_senders_dec=len(m_owned[p]) + len(halos["m2g"][p][0]),
        grid_node_feat=np.asarray(graphs["grid_node_feat"])[g_owned[p]],
        mesh_node_feat=np.asarray(graphs["mesh_node_feat"])[m_owned[p]],
        g2m=edge_set("g2m", g2m["feat"], n_grid, n_mesh, g_owned, m_owned),
        mesh=edge_set("mesh", mesh["feat"], n_mesh, n_mesh, m_owned, m_owned),
        m2g=edge_set("m2g", m2g["feat"], n_mesh, n_grid, m_owned, g_owned))
    mk = lambda key, owned_by_rank: HaloPlan(halos[key][p][0], halos[key][p][1],
                                             send_lists(key, owned_by_rank, p))
    ranks.append(RankGraphs(
        rank=p, graphs=local, grid_owned=g_owned[p], mesh_owned=m_owned[p],
        n_grid_owned=len(g_owned[p]), n_mesh_owned=len(m_owned[p]),
        halo_g2m=mk("g2m", g_owned), halo_mesh=mk("mesh", m_owned), halo_m2g=mk("m2g", m_owned)))
  return ranks


# ----------------------------------------------------------------------------- exchangers
class LocalExchanger:
  """All P ranks live in this process (emulation on one GPU / CPU): halo rows are copied
  directly between the ranks' tensors.  ``tensors[q]`` is rank q's row table
  ``[owned_q + halo_q, 512]``; the owned prefix is valid on entry, the halo suffix on exit.

  An exchange is, per rank, what ``DistExchanger`` does on a real rank -- ONE packing ``index_select`` of the rows it
  sends (all destinations, in destination order) and ONE landing of the rows it receives in its contiguous halo
  suffix (here a second ``index_select`` out of the P ranks' packed send buffers, which stand in for the wire): 2 P
  device launches per exchange, index tensors resident on the device.  (Until round 6 this walked the P x P pairs
  with a host-built index tensor each -- 18 exchanges cost 21 ms of mostly host time per emulated 8-way step,
  2.7 ms "per rank" that no rank of a real run spends; ``profiles/r06_s15_partition8_*``.)"""

  def __init__(self, plans: Sequence[HaloPlan], n_owned: Sequence[int]):
    self.plans, self.n_owned = list(plans), list(n_owned)
    n_parts = len(self.plans)
    as64 = lambda a: np.asarray(a, dtype=np.int64).reshape(-1)
    # rank q's packed send buffer = its rows for destination 0 | 1 | ... ; the P buffers back to back form the "wire"
    self._send_index = [np.concatenate([as64(pl.send_local[dst]) for dst in range(n_parts)]) if n_parts else as64([])
                        for pl in self.plans]
    starts = np.concatenate([[0], np.cumsum([len(i) for i in self._send_index])]).astype(np.int64)
    self._send_start, self._wire_rows = starts, int(starts[-1])
    # where on the wire the block (src -> dst) sits; a destination's halo suffix is those blocks in source order
    block = [[int(starts[src] + sum(len(self.plans[src].send_local[d]) for d in range(dst))) for dst in range(n_parts)]
             for src in range(n_parts)]
    self._recv_index = [np.concatenate([block[src][dst] + np.arange(len(self.plans[src].send_local[dst]), dtype=np.int64)
                                        for src in range(n_parts)]) if n_parts else as64([]) for dst in range(n_parts)]
    self._dev = {}            # device -> (send index tensors, receive index tensors)
    self._wire = {}           # (device, dtype, columns) -> the packed send buffers

  def _indices(self, device):
    import torch
    key = str(device)
    if key not in self._dev:
      self._dev[key] = ([torch.as_tensor(i, device=device) for i in self._send_index],
                        [torch.as_tensor(i, device=device) for i in self._recv_index])
    return self._dev[key]

  def exchange(self, tensors):
    import torch
    if self._wire_rows == 0:
      return
    device = tensors[0].device
    if any(t.device != device for t in tensors):          # ranks on several devices: pair by pair
      return self._exchange_pairs(tensors)
    send_idx, recv_idx = self._indices(device)
    shape, dtype = tuple(tensors[0].shape[1:]), tensors[0].dtype
    key = (str(device), dtype, shape)
    if key not in self._wire:
      self._wire[key] = torch.empty((self._wire_rows,) + shape, dtype=dtype, device=device)
    wire = self._wire[key]
    for src, t in enumerate(tensors):
      a, b = int(self._send_start[src]), int(self._send_start[src + 1])
      if b > a:
        torch.index_select(t, 0, send_idx[src], out=wire[a:b])
    for dst, t in enumerate(tensors):
      n = len(self._recv_index[dst])
      if n:
        halo = t[self.n_owned[dst]:self.n_owned[dst] + n]
        if halo.is_contiguous():
          torch.index_select(wire, 0, recv_idx[dst], out=halo)
        else:
          halo.copy_(wire.index_select(0, recv_idx[dst]))

  def _exchange_pairs(self, tensors):
    import torch
    n_parts = len(self.plans)
    for dst in range(n_parts):
      offset = self.n_owned[dst]
      for src in range(n_parts):
        idx = self.plans[src].send_local[dst]
        if len(idx) == 0:
          continue
        rows = tensors[src][torch.as_tensor(idx, device=tensors[src].device)]
        tensors[dst][offset:offset + len(idx)] = rows.to(tensors[dst].device)
        offset += len(idx)


class DistExchanger:
  """One rank per process: ONE ``all_to_all_single`` per exchange (nccl = RCCL over xGMI on GPUs,
  gloo on CPU).  Send rows are packed by one index_select, received rows land directly in the
  contiguous halo suffix of the table."""

  def __init__(self, plan_: HaloPlan, n_owned: int, device, group=None, host_staged=None):
    """`host_staged`: pack on the device, exchange through host buffers, unpack on the device --
    for process groups that cannot move device memory (gloo; e.g. two ranks sharing ONE GPU in
    tests).  Default: staged exactly when the table lives on a GPU and the group's backend is gloo."""
    import torch
    import torch.distributed as dist
    self.n_owned, self.group = n_owned, group
    if host_staged is None:
      host_staged = (torch.device(device).type == "cuda" and dist.is_initialized()
                     and dist.get_backend(group) == "gloo")
    self.host_staged = bool(host_staged)
    self.send_counts = [int(len(i)) for i in plan_.send_local]
    self.recv_counts = [int(c) for c in plan_.recv_counts]
    idx = np.concatenate([np.asarray(i, dtype=np.int64) for i in plan_.send_local]) \
        if sum(self.send_counts) else np.zeros(0, dtype=np.int64)
    self.send_index = torch.as_tensor(idx, device=device)

  def exchange(self, table):
    import torch
    import torch.distributed as dist
    send = table.index_select(0, self.send_index) if self.send_index.numel() else table[:0]
    recv = table[self.n_owned:self.n_owned + sum(self.recv_counts)]
    if self.host_staged:
      send_h = send.contiguous().cpu()                 # (synchronises the launch stream: test path)
      recv_h = torch.empty(recv.shape, dtype=recv.dtype)
      dist.all_to_all_single(recv_h, send_h, output_split_sizes=self.recv_counts,
                             input_split_sizes=self.send_counts, group=self.group)
      recv.copy_(recv_h)
      return table
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=self.recv_counts,
                           input_split_sizes=self.send_counts, group=self.group)
    return table


# ----------------------------------------------------------------------------- runners
def tables_of(rank_graphs: RankGraphs):
  """Exchange name -> (halo plan, owned rows) of one rank."""
  return {"g2m": (rank_graphs.halo_g2m, rank_graphs.n_grid_owned),
          "mesh": (rank_graphs.halo_mesh, rank_graphs.n_mesh_owned),
          "m2g": (rank_graphs.halo_m2g, rank_graphs.n_mesh_owned)}


class EmulatedPartitionedStep:
  """All P ranks of a partitioned step in ONE process on ONE device: P engines over the local
  graphs run their launch segments in lockstep and halo rows are copied between their tables.
  Numerically this is the multi-GPU execution (same kernels, same local graphs, same exchange
  points); it exists to validate config 5 on a single MI355X and in CI."""

  def __init__(self, graphs: dict, params, grid_lon, mesh_lon, n_parts: int, *, num_steps: int,
               c_in: int, c_out: int, device="cuda:0", precision=None, grid_lat=None, mesh_lat=None):
    from graphcast_amd import engine
    self.ranks = plan(graphs, grid_lon, mesh_lon, n_parts, grid_lat=grid_lat, mesh_lat=mesh_lat)
    self.engines = [engine.StepEngine(r.graphs, params, num_steps=num_steps, c_in=c_in, c_out=c_out,
                                      device=device, precision=precision) for r in self.ranks]
    self.exchangers = {
        name: LocalExchanger([tables_of(r)[name][0] for r in self.ranks],
                             [tables_of(r)[name][1] for r in self.ranks])
        for name in ("g2m", "mesh", "m2g")}
    self.n_grid = int(graphs["n_grid"])
    self.c_out = c_out
    self.exchanges_per_call = 0
    self._owned_dev = {}
    # (rounds 3-4 could split every edge update into sender-local and halo-sender launches and run the exchange on a
    #  second stream under the first -- GCAST_OVERLAP=1.  Retired in round 5 on the measurements: the 18 extra small
    #  launches + joins cost 1.6 ms per rank at 8-way (profiles/r03_s9_*) against 0.47 ms for ALL 18 exchanges of a
    #  step on the device (bench.py --mode partition: roofline.exchange, profiles/r05_s1_*) -- there is less to hide
    #  than the hiding costs.  Every edge update is one launch behind a blocking exchange.)

  def forward(self, x):
    import torch
    owned = self._owned_rows(x.device)
    xs = [x.index_select(0, o) for o in owned]
    bound = [e.segments(xl) for e, xl in zip(self.engines, xs)]
    n_seg = len(bound[0][1])
    # (engine.segments: one segment per cut, so the structure does not depend on a rank's own split decisions)
    assert all([a for _, a in segs] == [a for _, a in bound[0][1]] for _, segs in bound), \
        "the ranks' segment / action lists differ"
    self.exchanges_per_call = 0
    for k in range(n_seg):
      for _, segs in bound:
        segs[k][0]()
      for kind, name in bound[0][1][k][1]:
        if kind == "start":
          self.exchanges_per_call += 1
          self.exchangers[name].exchange([e.halo_table(name) for e in self.engines])
    y = torch.empty((self.n_grid, x.shape[1], self.c_out), dtype=torch.float32, device=x.device)
    for o, (yl, _) in zip(owned, bound):
      y.index_copy_(0, o, yl)
    return y

  def _owned_rows(self, device):
    """Every rank's owned grid rows as index tensors resident on `device` (built once: a real rank holds its rows)."""
    import torch
    key = str(device)
    if key not in self._owned_dev:
      self._owned_dev[key] = [torch.as_tensor(r.grid_owned, dtype=torch.int64, device=device) for r in self.ranks]
    return self._owned_dev[key]

  __call__ = forward


class DistributedPartitionedStep:
  """One rank of a partitioned step, one process per GPU: the local engine's segments
  interleaved with ONE all_to_all_single per halo exchange (RCCL over xGMI)."""

  def __init__(self, rank_graphs: RankGraphs, params, *, num_steps: int, c_in: int, c_out: int,
               device, precision=None, group=None):
    from graphcast_amd import engine
    self.rank_graphs = rank_graphs
    self.engine = engine.StepEngine(rank_graphs.graphs, params, num_steps=num_steps, c_in=c_in,
                                    c_out=c_out, device=device, precision=precision)
    self.exchangers = {name: DistExchanger(pl, n_owned, device, group)
                       for name, (pl, n_owned) in tables_of(rank_graphs).items()}

  def forward(self, x_local, y_local=None):
    """x_local = rows ``rank_graphs.grid_owned`` of the global [N_grid, B, C_in] input."""
    import torch
    y, segs = self.engine.segments(x_local, y_local)
    for run, actions in segs:
      run()
      for kind, name in actions:
        if kind == "start":                  # ONE all_to_all_single, in front of the edge update that gathers from the suffix
          self.exchangers[name].exchange(self.engine.halo_table(name))
    return y

  __call__ = forward
