"""Norm-conditioned encoder / decoder on the device.

GenCast's denoiser reuses GraphCast's two bipartite GNNs unchanged except for one switch
(``weathernext1_gen/denoiser.py:303-363``): ``DeepTypedGraphNet(use_norm_conditioning=True)``.
Every LayerNorm then loses its learned scale / offset and is followed by
``dense.LinearNormConditioning`` (``utils/dense.py:360-393``, wired at
``utils/legacy/deep_typed_graph_net.py:210-246``):

    [s | o] = cond @ w + b          (cond: one vector per batch element, e.g. the noise level code)
    y       = LayerNorm(x) * (1 + s) + o

On the MI355X this needs NO new kernel: a ``gc_rowmlp`` launch already takes its LayerNorm scale /
offset as two 512-vectors, and the engine launches once per batch element anyway -- so element b
simply gets ``(1 + s_b, o_b)``.  The vectors themselves come from the same library
(``GC_MODE_LINEAR`` launches over the [B, C_cond] conditioning rows, with the ``+ 1`` folded into the
bias).  Edge embeddings depend on the conditioning here, so nothing is constant-folded at load time
(GraphCast's engine folds them, engine.py); the first edge-MLP layer is still split per node
((x[idx]).W == (x.W)[idx]).

The sparse-transformer processor between the two (``denoiser.py:331-339``) is out of scope.
"""
from typing import Mapping, Optional

import numpy as np
import torch

from graphcast_amd import _native as nat
from graphcast_amd import launch
from graphcast_amd import packing

D = packing.LATENT
_G = "grid2mesh_gnn/~_networks_builder/"
_X = "mesh2grid_gnn/~_networks_builder/"


class _Cond:
  """One ``<stem>_norm_conditioning/linear``: packed halves of w and the (+1)-folded biases."""

  def __init__(self, params, stem, kc, up, pack):
    w = np.asarray(params[f"{stem}_norm_conditioning/linear"]["w"], dtype=np.float32)
    b = np.asarray(params[f"{stem}_norm_conditioning/linear"]["b"], dtype=np.float32)
    if w.shape[1] != 2 * D:
      raise NotImplementedError(f"norm conditioning must produce 2 x {D} values, got {w.shape}")
    self.w_scale, self.w_offset = pack(w[:, :D]), pack(w[:, D:])
    self.b_scale, self.b_offset = up(b[:D] + np.float32(1.0)), up(b[D:])
    self.k = kc


class ConditionedEncoderDecoder(launch.LaunchBase):
  """grid2mesh encoder + mesh2grid decoder with ``global_norm_conditioning``.

  graphs: ``n_grid``, ``n_mesh``, ``g2m`` / ``m2g`` = dict(senders, receivers, feat [E, <=32]).
  params: haiku tree of the two GNNs built with ``use_norm_conditioning=True``
          (``oracle.params.conditioned_module_specs`` lists the modules).
  encode(grid_x [N_g, B, C_g], mesh_x [N_m, B, C_m], cond [B, C_c]) -> (latent_mesh, latent_grid)
  decode(latent_mesh, latent_grid, cond) -> [N_g, B, C_out]            (all fp32 device tensors)
  """

  def __init__(self, graphs: Mapping, params: Mapping, *, c_grid: int, c_mesh: int, c_cond: int,
               c_out: int, device="cuda:0", precision: Optional[str] = None):
    self.dev = torch.device(device)
    self.lib = nat.lib()
    precision = precision or launch.DEFAULT_PRECISION
    self.precision, self.prec = precision, nat.PRECISIONS[precision]
    import os
    self.half = self.prec == nat.PREC_F16X3
    self.scratch = None
    # latents come from the caller: every launch that reads rows carries the f16x3 range flag (launch.LaunchBase.check_range)
    self.check_all_rows = True
    self.range_flag = (torch.zeros((1,), dtype=torch.int32, device=self.dev)
                       if self.half and self.prec == nat.PREC_F16X3 else None)
    self.onepass = False       # (its launches carry per-batch LayerNorm vectors through the two-pass kernels)
    self.n_grid, self.n_mesh = int(graphs["n_grid"]), int(graphs["n_mesh"])
    self.c_grid, self.c_mesh, self.c_cond, self.c_out = c_grid, c_mesh, c_cond, c_out
    if c_out > 240:
      raise NotImplementedError("decoder width above 240 needs a wider output tile")
    self._keep = []
    dev = self.dev
    M = lambda stem, **kw: launch._Mlp(params, stem, dev, prec=self.prec, **kw)
    esr = ("e", "s", "r")
    self.m_enc_grid = M(_G + "encoder_nodes_grid_nodes")
    self.m_enc_mesh = M(_G + "encoder_nodes_mesh_nodes")
    self.m_enc_e_g2m = M(_G + "encoder_edges_grid2mesh")
    self.m_g2m_edge = M(_G + "processor_edges_0_grid2mesh", split=esr)
    self.m_g2m_mesh = M(_G + "processor_nodes_0_mesh_nodes")
    self.m_g2m_grid = M(_G + "processor_nodes_0_grid_nodes")
    self.m_enc_e_m2g = M(_X + "encoder_edges_mesh2grid")
    self.m_m2g_edge = M(_X + "processor_edges_0_mesh2grid", split=esr)
    self.m_m2g_grid = M(_X + "processor_nodes_0_grid_nodes")
    self.m_out = M(_X + "decoder_nodes_grid_nodes", np2=256)
    if self.m_out.n_out != c_out:
      raise ValueError(f"decoder produces {self.m_out.n_out} channels, asked for {c_out}")
    for m, c, what in ((self.m_enc_grid, c_grid, "grid"), (self.m_enc_mesh, c_mesh, "mesh")):
      if m.k_in != c:
        raise ValueError(f"{what} embedder expects {m.k_in} input channels, got {c}")
    self.kc = packing.round_up(c_cond, packing.K_CHUNK)

    def pack_w(w):
      # the same packing _Mlp applies to a first-layer matrix of this precision
      holder = {"x_mlp/~/linear_0": {"w": w, "b": np.zeros(D, np.float32)},
                "x_mlp/~/linear_1": {"w": np.zeros((D, D), np.float32), "b": np.zeros(D, np.float32)}}
      return launch._Mlp(holder, "x", dev, prec=self.prec).w1

    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cond_stems = dict(
        enc_grid=_G + "encoder_nodes_grid_nodes", enc_mesh=_G + "encoder_nodes_mesh_nodes",
        enc_e_g2m=_G + "encoder_edges_grid2mesh", g2m_edge=_G + "processor_edges_0_grid2mesh",
        g2m_mesh=_G + "processor_nodes_0_mesh_nodes", g2m_grid=_G + "processor_nodes_0_grid_nodes",
        enc_e_m2g=_X + "encoder_edges_mesh2grid", m2g_edge=_X + "processor_edges_0_mesh2grid",
        m2g_grid=_X + "processor_nodes_0_grid_nodes")
    self.cond = {k: _Cond(params, stem, self.kc, up, pack_w) for k, stem in cond_stems.items()}
    self._keep += [self.m_enc_grid, self.m_enc_mesh, self.m_enc_e_g2m, self.m_g2m_edge, self.m_g2m_mesh,
                   self.m_g2m_grid, self.m_enc_e_m2g, self.m_m2g_edge, self.m_m2g_grid, self.m_out, self.cond]

    self.e_g2m = launch._Edges(packing.pack_edges(graphs["g2m"]["senders"], graphs["g2m"]["receivers"],
                                                  self.n_mesh), dev)
    self.e_m2g = launch._Edges(packing.pack_edges(graphs["m2g"]["senders"], graphs["m2g"]["receivers"],
                                                  self.n_grid), dev)

    def edge_feat_rows(edges, feat):
      feat = np.asarray(feat, dtype=np.float32)
      if feat.shape[1] > packing.K_CHUNK:
        raise NotImplementedError("more than 32 structural edge features")
      rows = np.zeros((edges.n_rows, packing.K_CHUNK), dtype=np.float32)
      ok = edges.pk.perm >= 0
      rows[ok, :feat.shape[1]] = feat[edges.pk.perm[ok]]
      return up(rows)

    self.ef_g2m = edge_feat_rows(self.e_g2m, graphs["g2m"]["feat"])
    self.ef_m2g = edge_feat_rows(self.e_m2g, graphs["m2g"]["feat"])
    ng, nm = self.n_grid, self.n_mesh
    self.kg, self.km = self.m_enc_grid.k1p, self.m_enc_mesh.k1p
    self.xg, self.xm = self._new(ng, self.kg), self._new(nm, self.km)
    self.xg.zero_()
    self.xm.zero_()
    self.h_grid, self.h_mesh0 = self._new(ng), self._new(nm)
    self.e0_g2m, self.e0_m2g = self._new(self.e_g2m.n_rows), self._new(self.e_m2g.n_rows)
    self.pre_grid, self.pre_mesh = self._new(ng), self._new(nm)
    self.agg_mesh, self.agg_grid = self._new(nm), self._new(ng)
    self.h_dec = self._new(ng)
    self._ln = {}          # module key -> (scale [B, 512], offset [B, 512]) of the current call

  # ---------------------------------------------------------------- conditioning vectors
  def _conditioning(self, cond: torch.Tensor, keys):
    """(1 + s_b, o_b) of every listed module, for all batch elements: two LINEAR launches each."""
    if (cond.dtype != torch.float32 or cond.dim() != 2 or cond.shape[1] != self.c_cond
        or cond.device != self.dev):
      raise ValueError(f"cond must be a float32 [B, {self.c_cond}] tensor on the engine's device")
    b = cond.shape[0]
    rows = torch.zeros((b, self.kc), dtype=torch.float32, device=self.dev)
    rows[:, :self.c_cond] = cond
    ops, out = [], {}
    for k in keys:
      c = self.cond[k]
      scale = torch.empty((b, D), dtype=torch.float32, device=self.dev)
      offset = torch.empty((b, D), dtype=torch.float32, device=self.dev)
      ops.append(self._op_mlp("enc_pre", self._desc(nat.MODE_LINEAR, b, a0=rows, k0=self.kc,
                                                    w1p=c.w_scale, b1=c.b_scale, out=scale)))
      ops.append(self._op_mlp("enc_pre", self._desc(nat.MODE_LINEAR, b, a0=rows, k0=self.kc,
                                                    w1p=c.w_offset, b1=c.b_offset, out=offset)))
      out[k] = (scale, offset)
    self._run(ops)
    self._cond_rows = rows      # alive until the launches have run
    return out

  def _cln(self, n_rows, mlp, key, b, **kw):
    """MLP + conditional LayerNorm of batch element b (LN scale / offset = row b of the tables)."""
    scale, offset = self._ln[key]
    return self._desc(nat.MODE_MLP_LN, n_rows, w2p=mlp.w2, b2=mlp.b2, n2=D,
                      ln=(scale[b], offset[b]), **kw)

  @staticmethod
  def _check(t, rows, cols, what, dev):
    if (t.dtype != torch.float32 or t.dim() != 3 or t.shape[0] != rows or t.shape[2] != cols
        or not t.is_contiguous() or t.device != dev):
      raise ValueError(f"{what} must be a contiguous float32 [{rows}, B, {cols}] tensor on the engine's device")

  # ---------------------------------------------------------------- encoder
  def encode(self, grid_x: torch.Tensor, mesh_x: torch.Tensor, cond: torch.Tensor):
    """reference denoiser.py:303-330 / graphcast.py:550-604 with conditioning."""
    self._check(grid_x, self.n_grid, self.c_grid, "grid_x", self.dev)
    self._check(mesh_x, self.n_mesh, self.c_mesh, "mesh_x", self.dev)
    batch = grid_x.shape[1]
    if mesh_x.shape[1] != batch or cond.shape[0] != batch:
      raise ValueError("grid_x, mesh_x and cond disagree on the batch size")
    ng, nm = self.n_grid, self.n_mesh
    self._ln = self._conditioning(cond, ("enc_grid", "enc_mesh", "enc_e_g2m", "g2m_edge", "g2m_mesh", "g2m_grid"))
    lat_mesh = torch.empty((nm, batch, D), dtype=torch.float32, device=self.dev)
    lat_grid = torch.empty((ng, batch, D), dtype=torch.float32, device=self.dev)
    for b in range(batch):
      self.xg[:, :self.c_grid] = grid_x[:, b]
      self.xm[:, :self.c_mesh] = mesh_x[:, b]
      ops = []
      m = self.m_enc_grid
      ops.append(self._op_mlp("enc_embed_grid", self._cln(ng, m, "enc_grid", b, a0=self.xg, k0=self.kg,
                                                          w1p=m.w1, b1=m.b1, out=self.h_grid)))
      m = self.m_enc_mesh
      ops.append(self._op_mlp("enc_pre", self._cln(nm, m, "enc_mesh", b, a0=self.xm, k0=self.km,
                                                   w1p=m.w1, b1=m.b1, out=self.h_mesh0)))
      m = self.m_enc_e_g2m
      ops.append(self._op_mlp("enc_pre", self._cln(self.e_g2m.n_rows, m, "enc_e_g2m", b, a0=self.ef_g2m,
                                                   k0=packing.K_CHUNK, w1p=m.w1, b1=m.b1, out=self.e0_g2m)))
      m = self.m_g2m_edge
      ops.append(self._op_mlp("enc_pre", self._desc(nat.MODE_LINEAR, ng, a0=self.h_grid, k0=D,
                                                    w1p=m.w1["s"], out=self.pre_grid)))
      ops.append(self._op_mlp("enc_pre", self._desc(nat.MODE_LINEAR, nm, a0=self.h_mesh0, k0=D,
                                                    w1p=m.w1["r"], out=self.pre_mesh)))
      ops.append(self._op_mlp("enc_edge", self._cln(
          self.e_g2m.n_rows, m, "g2m_edge", b, a0=self.e0_g2m, k0=D, w1p=m.w1["e"], b1=m.b1,
          g0=self.pre_grid, idx0=self.e_g2m.snd, g1=self.pre_mesh, idx1=self.e_g2m.rcv,
          edges=self.e_g2m, agg=self.agg_mesh)))
      ops += self._ops_after_segsum(self.e_g2m, self.agg_mesh)
      m = self.m_g2m_mesh
      d = self._cln(nm, m, "g2m_mesh", b, a0=self.h_mesh0, k0=D, a1=self.agg_mesh, k1=D, w1p=m.w1,
                    b1=m.b1, res=self.h_mesh0, out_ptr=lat_mesh.data_ptr() + 4 * b * D, ldo=batch * D)
      ops.append(self._op_mlp("enc_node_mesh", d))
      m = self.m_g2m_grid
      d = self._cln(ng, m, "g2m_grid", b, a0=self.h_grid, k0=D, w1p=m.w1, b1=m.b1, res=self.h_grid,
                    out_ptr=lat_grid.data_ptr() + 4 * b * D, ldo=batch * D)
      ops.append(self._op_mlp("enc_node_grid", d))
      self._run(ops)
    self.check_range()
    return lat_mesh, lat_grid

  # ---------------------------------------------------------------- decoder
  def decode(self, latent_mesh: torch.Tensor, latent_grid: torch.Tensor, cond: torch.Tensor):
    """reference denoiser.py:340-363 / graphcast.py:641-678 with conditioning (the decoder's
    mesh-node update is never read, graphcast.py:676: not computed)."""
    self._check(latent_mesh, self.n_mesh, D, "latent_mesh", self.dev)
    self._check(latent_grid, self.n_grid, D, "latent_grid", self.dev)
    batch = latent_grid.shape[1]
    if latent_mesh.shape[1] != batch or cond.shape[0] != batch:
      raise ValueError("latent_mesh, latent_grid and cond disagree on the batch size")
    ng, nm = self.n_grid, self.n_mesh
    self._ln = self._conditioning(cond, ("enc_e_m2g", "m2g_edge", "m2g_grid"))
    y = torch.empty((ng, batch, self.c_out), dtype=torch.float32, device=self.dev)
    for b in range(batch):
      hm, hg = latent_mesh[:, b], latent_grid[:, b]           # row stride batch * 512
      ld = batch * D
      ops = []
      m = self.m_enc_e_m2g
      ops.append(self._op_mlp("dec_pre", self._cln(self.e_m2g.n_rows, m, "enc_e_m2g", b, a0=self.ef_m2g,
                                                   k0=packing.K_CHUNK, w1p=m.w1, b1=m.b1, out=self.e0_m2g)))
      m = self.m_m2g_edge
      ops.append(self._op_mlp("dec_pre", self._desc(nat.MODE_LINEAR, nm, a0=hm, k0=D, lda0=ld,
                                                    w1p=m.w1["s"], out=self.pre_mesh)))
      ops.append(self._op_mlp("dec_pre", self._desc(nat.MODE_LINEAR, ng, a0=hg, k0=D, lda0=ld,
                                                    w1p=m.w1["r"], out=self.pre_grid)))
      ops.append(self._op_mlp("dec_edge", self._cln(
          self.e_m2g.n_rows, m, "m2g_edge", b, a0=self.e0_m2g, k0=D, w1p=m.w1["e"], b1=m.b1,
          g0=self.pre_mesh, idx0=self.e_m2g.snd, g1=self.pre_grid, idx1=self.e_m2g.rcv,
          edges=self.e_m2g, agg=self.agg_grid)))
      ops += self._ops_after_segsum(self.e_m2g, self.agg_grid)
      m = self.m_m2g_grid
      d = self._cln(ng, m, "m2g_grid", b, a0=hg, k0=D, lda0=ld, a1=self.agg_grid, k1=D, w1p=m.w1,
                    b1=m.b1, res=hg, out=self.h_dec)
      d.ldres = ld
      ops.append(self._op_mlp("dec_node", d))
      m = self.m_out
      ops.append(self._op_mlp("dec_out", self._desc(
          nat.MODE_MLP_OUT, ng, a0=self.h_dec, k0=D, w1p=m.w1, b1=m.b1, w2p=m.w2, b2=m.b2,
          n2=self.c_out, out_ptr=y.data_ptr() + 4 * b * self.c_out, ldo=batch * self.c_out)))
      self._run(ops)
    self.check_range()
    return y

  # the GraphCast step API of the base class does not apply here
  def forward(self, *a, **k):
    raise TypeError("ConditionedEncoderDecoder has encode() / decode(), not a fused step")

  __call__ = forward
