"""`StepEngine`: one GraphCast encode-process-decode step on an MI355X, driven through the C-ABI's plan.

The three static graphs (``graphcast.py`` builds them exactly like the reference's ``_init_*_graph``,
weathernext1_graph/graphcast.py:408-548) and the haiku parameter tree go to ``gc_plan_create``
(csrc/gcast_plan.inc), which owns everything static:
  * packed, receiver-sorted edge sets; packed weights with the first edge-MLP matrix split into its edge / sender /
    receiver row blocks (W1 = [W_e; W_s; W_r] in concat order, deep_typed_graph_net.py:209 + typed_graph_net.py:448-453),
    so the sender / receiver products are taken per NODE before the gather ((x[idx]).W == (x.W)[idx]);
  * the input-independent terms folded once at load time ON THE DEVICE with the same kernels;
  * THE launch program of the step (``gc_plan_program``): since round 5 there is ONE builder.  Rounds 1-4 recorded the
    same program a second time here (``_program_fused`` / ``_program_plain`` / ``_build`` / ``_build_bf16``), held
    together with the C++ one by bit-equality tests.

What this class adds to the plan is what a Python host wants around it: torch-owned workspace, the step as an
exported op array that can be run whole (``forward``), up to a stage boundary (``run_until``: tests compare stage
boundaries with the oracle), launch by launch with HIP events (``time_ops``: bench.py's per-stage roofline) or as
segments between halo-exchange points (``segments``: the spatially partitioned step of partition.py), named views of
the workspace tensors, and the range flag of the f16x3 arithmetic read the torch way.

All arithmetic happens in libgcast_hip.so; torch only owns the memory.
"""
import ctypes
import os
from typing import Mapping, Optional

import numpy as np
import torch

from graphcast_amd import _native as nat
from graphcast_amd import launch
from graphcast_amd import packing
from graphcast_amd.launch import DEFAULT_PRECISION, TAGS, _Edges, _Mlp, _PW, _chained    # noqa: F401  (re-exported)

D = packing.LATENT

_EDGE_TAGS = {TAGS["enc_edge"]: "g2m", TAGS["proc_edge"]: "mesh", TAGS["dec_edge"]: "m2g"}
_WORKSPACE_TENSORS = ("xin", "h_grid", "pre_grid", "h_grid2", "agg_grid", "h_mesh", "agg_mesh", "pre_s_mesh",
                      "pre_r_mesh", "e_mesh_lat")


def _f32(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i32(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def tensor_descs(params: Mapping[str, Mapping[str, np.ndarray]]):
  """haiku tree {"module": {"w": ...}} -> (TensorDesc array, keep-alive list), names "module/leaf"."""
  keep, descs = [], []
  for module, leaves in params.items():
    for leaf, value in leaves.items():
      a = _f32(value)
      a2 = a.reshape(1, -1) if a.ndim == 1 else a
      name = f"{module}/{leaf}".encode()
      keep += [a2, name]
      descs.append(nat.TensorDesc(name, a2.ctypes.data, a2.shape[0], a2.shape[1]))
  return (nat.TensorDesc * len(descs))(*descs), keep


def create_plan(lib, graphs: Mapping, params: Mapping, *, num_steps, c_in, c_out, precision, device):
  """gc_plan_create on the reference-layout arrays; -> the plan handle (a ctypes.c_void_p the caller destroys)."""
  keep = []

  def edge_set(g):
    s, r, f = _i32(g["senders"]), _i32(g["receivers"]), _f32(g["feat"])
    keep.extend([s, r, f])
    return nat.EdgeSet(len(s), s.ctypes.data, r.ctypes.data, f.ctypes.data, f.shape[1])

  gnf, mnf = _f32(graphs["grid_node_feat"]), _f32(graphs["mesh_node_feat"])
  half = precision in ("f16x3", "bf16")         # the half-N kernel family (f32: the chunked exact-fp32 kernel)
  model = nat.ModelDesc(int(graphs["n_grid"]), int(graphs["n_mesh"]), c_in, c_out, gnf.shape[1], num_steps,
                        nat.PRECISIONS[precision], gnf.ctypes.data, mnf.ctypes.data,
                        edge_set(graphs["g2m"]), edge_set(graphs["mesh"]), edge_set(graphs["m2g"]),
                        nat.LAYOUT_HALF if half else nat.LAYOUT_CHUNKED,
                        # spatially partitioned graphs (partition.plan): sender tables with a halo suffix
                        int(graphs.get("n_grid_senders", 0)), int(graphs.get("n_mesh_senders", 0)),
                        int(graphs.get("n_mesh_senders_dec", 0)))
  tensors, keep_t = tensor_descs(params)
  handle = ctypes.c_void_p()
  dev = torch.device(device)
  with torch.cuda.device(dev):
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nat.check(lib.gc_plan_create(ctypes.byref(model), tensors, len(tensors), stream, ctypes.byref(handle)),
              "gc_plan_create")
  del keep, keep_t
  return handle


class StepEngine(launch.LaunchBase):
  """x [N_grid, B, C_in] fp32 (device) -> y [N_grid, B, C_out] fp32 (device)."""

  def __init__(self, graphs: Mapping, params: Mapping, *, num_steps: int, c_in: int, c_out: int,
               device="cuda:0", precision: Optional[str] = None, half: Optional[bool] = None):
    self.dev = torch.device(device)
    self.lib = nat.lib()
    precision = precision or os.environ.get("GCAST_PRECISION", DEFAULT_PRECISION)
    if precision not in nat.PRECISIONS:
      raise ValueError(f"precision must be one of {sorted(nat.PRECISIONS)}, got {precision!r}")
    self.precision = precision
    self.prec = nat.PRECISIONS[precision]
    # f16x3 and bf16 run the half-N formulation (csrc/rowmlp_half.inc, rowmlp_bf16.inc: two workgroups per CU, so one
    # tile's non-GEMM phases run under the other's MFMAs); f32 the chunked round-1 kernel.
    if half is False and self.prec == nat.PREC_F16X3:
      raise ValueError("half=False: the chunked f16x3 kernels were retired in round 5 (f16x3 runs the half-N kernels)")
    self.half = self.prec in (nat.PREC_F16X3, nat.PREC_BF16)
    # The speed-only switches come from the library's ONE tuning surface (include/gcast.h: gc_tuning; the GCAST_*
    # variables only initialise its process default): the plan snapshots it at creation, the engine reads the snapshot
    # back below.  helpers_min_rows: launches without gather / segment-sum from that many rows on run as one eight-wave
    # workgroup per CU (0 = never); the attribute lets a caller (smoke(), tests) change it per engine.
    self.wide_min_rows = None      # (None: the plan's choice.  An int: two-pass MLP launches without gather / segment-sum
    #                                 from that many rows on run in the wide form -- smoke(), tests)
    self.wide_edges = 0            # (bit 0: every one-pass edge update, bit 1: every two-pass one, in the wide form WHATEVER
    #                                 its size -- smoke(), tests; 0: the library's rule, gc_tuning.wide_edges)
    self.n_grid, self.n_mesh = int(graphs["n_grid"]), int(graphs["n_mesh"])
    self.c_in, self.c_out, self.num_steps = c_in, c_out, num_steps
    if c_out > 240:
      raise NotImplementedError("decoder width above 240 needs a wider output tile")
    self._views = {}
    self._plan = create_plan(self.lib, graphs, params, num_steps=num_steps, c_in=c_in, c_out=c_out,
                             precision=precision, device=self.dev)
    self.tuning = nat.Tuning()
    nat.check(self.lib.gc_plan_get_tuning(self._plan, ctypes.byref(self.tuning)), "gc_plan_get_tuning")
    self.fuse = self.half and (bool(self.tuning.fuse) or self.prec == nat.PREC_BF16)
    self.helpers_min_rows = self._plan_helpers_min_rows = int(self.tuning.helpers_min_rows)
    with torch.cuda.device(self.dev):
      self._ws = torch.empty(self.lib.gc_plan_workspace_bytes(self._plan, 1), dtype=torch.uint8, device=self.dev)
    # the f16x3 kernels' range word (include/gcast.h: gc_rowmlp_desc.range_flag) and the persistent kernels' tile queue:
    # both live in the workspace; this class owns them (gc_plan_program clears nothing)
    self.range_flag = self._tensor("range_flag", optional=True)
    self.tile_queue = self._tensor("tile_queue", optional=True)
    if self.range_flag is not None:
      self.range_flag.zero_()
    if self.tile_queue is not None:
      self.tile_queue.zero_()
    # (the exchange of a partitioned step is never split into sender-local / halo-sender launches any more: the
    #  GCAST_OVERLAP mode of rounds 3-4 cost 1.6 ms per rank in extra launches against <= 1.1 ms of exchange it could
    #  hide -- DESIGN.md section 7; the attribute keeps partition.py's callers simple)
    self.halo = dict(g2m=None, mesh=None, m2g=None)
    self._cap = 0

  # ---------------------------------------------------------------- named tensors of the plan's workspace
  def _tensor(self, name, optional=False):
    """A torch VIEW of the workspace tensor `name` (gc_plan_tensor): fp32 [rows, cols], bfloat16 rows in the bf16 tier,
    int32 for the two control words."""
    if name in self._views:
      return self._views[name]
    ptr, rows, cols, eb = ctypes.c_void_p(), ctypes.c_longlong(), ctypes.c_int(), ctypes.c_int()
    rc = self.lib.gc_plan_tensor(self._plan, self._ws.data_ptr(), name.encode(), ctypes.byref(ptr), ctypes.byref(rows),
                                 ctypes.byref(cols), ctypes.byref(eb))
    if rc != 0:
      if optional:
        return None
      nat.check(rc, "gc_plan_tensor")
    off = ptr.value - self._ws.data_ptr()
    n = rows.value * cols.value
    if not 0 <= off <= self._ws.numel() - n * eb.value:
      raise RuntimeError(f"gc_plan_tensor({name}): not inside the workspace")
    control = name in ("range_flag", "tile_queue")
    dtype = torch.int32 if control else torch.bfloat16 if eb.value == 2 else torch.float32
    view = self._ws[off:off + n * eb.value].view(dtype)
    view = view if control else view.view(rows.value, cols.value)
    self._views[name] = view
    return view

  def __getattr__(self, name):
    # (only reached for names that are not set: the workspace row tensors as attributes, e.g. engine.agg_mesh)
    if name in _WORKSPACE_TENSORS and "_views" in self.__dict__:
      return self._tensor(name)
    raise AttributeError(name)

  # ---------------------------------------------------------------- the program
  def bind(self, x: torch.Tensor, y: Optional[torch.Tensor] = None):
    """Validates x / y and returns (the plan's op array for them -- a fresh ctypes array --, y)."""
    if x.dtype != torch.float32 or x.dim() != 3 or not x.is_contiguous() or x.device != self.dev:
      raise ValueError("x must be a contiguous float32 [N_grid, B, C_in] tensor on the engine's device")
    if x.shape[0] != self.n_grid or x.shape[2] != self.c_in:
      raise ValueError(f"x has shape {tuple(x.shape)}, expected [{self.n_grid}, B, {self.c_in}]")
    batch = x.shape[1]
    if y is None:
      y = torch.empty((self.n_grid, batch, self.c_out), dtype=torch.float32, device=self.dev)
    elif (y.shape != (self.n_grid, batch, self.c_out) or y.dtype != torch.float32
          or not y.is_contiguous() or y.device != self.dev):
      raise ValueError("y must be a contiguous float32 [N_grid, B, C_out] tensor on the engine's device")
    cap = max(self._cap, 64 + batch * (16 + 8 * self.num_steps))
    while True:
      arr = (nat.Op * cap)()
      n = ctypes.c_int()
      rc = self.lib.gc_plan_program(self._plan, x.data_ptr(), y.data_ptr(), batch, self._ws.data_ptr(),
                                    self._ws.numel(), arr, cap, ctypes.byref(n))
      if rc != 0 and n.value > cap:
        cap = n.value
        continue
      nat.check(rc, "gc_plan_program")
      break
    self._cap = cap
    ops = (nat.Op * n.value).from_buffer(arr)       # (the first n entries; shares `arr`'s memory and keeps it alive)
    # the engine-level knob on top of the plan's default (which marks the big node-side launches GC_WG_WIDE, or
    # GC_WG_HELPERS where the wide form does not apply): set to something else than the default, it puts every launch
    # without gather / segment-sum from that many rows on into the eight-wave HELPER form (smoke(), tests)
    # -- or, with `wide_min_rows` set, into the WIDE form where the launch has the shape for it.
    if (self.wide_min_rows is not None
        or self.helpers_min_rows != self._plan_helpers_min_rows):
      for k in range(n.value):
        m = ops[k].mlp
        if (ops[k].kind == nat.OP_ROWMLP and m.layout == nat.LAYOUT_HALF and m.prec == nat.PREC_F16X3
            and not m.g0 and not m.seg):
          m.flags &= ~(nat.WG_HELPERS | nat.WG_WIDE)
          if (self.wide_min_rows is not None and m.n_rows >= self.wide_min_rows and m.mode == nat.MODE_MLP_LN
              and m.k0 + m.k1 > 0):
            m.flags |= nat.WG_WIDE
          elif self.helpers_min_rows and m.n_rows >= self.helpers_min_rows and m.k0 + m.k1 > 0:
            # (k0 + k1 > 0: the addend-only launches of a hidden_layers > 1 plan stay in the four-wave form, as in the plan's own rule)
            m.flags |= nat.WG_HELPERS
    if self.wide_edges:
      for k in range(n.value):
        m = ops[k].mlp
        if (ops[k].kind == nat.OP_ROWMLP and m.layout == nat.LAYOUT_HALF and m.prec == nat.PREC_F16X3 and m.seg
            and m.mode == nat.MODE_MLP_LN and (self.wide_edges & (1 if m.flags & nat.W2_NATURAL else 2))):
          m.flags = (m.flags & ~(nat.WG_HELPERS | nat.WG_NO_HELPERS)) | nat.WG_WIDE
    return ops, y

  def _clear_tile_queue(self):
    """The two tile-queue words are left zero by every launch that COMPLETES; an aborted launch, or a caller who
    overlapped two launches of one engine on different streams, would leave them non-zero and every later launch
    would silently skip tiles.  One 8-byte memset at the head of each enqueued program (as gc_step_forward does)."""
    if self.tile_queue is not None:
      self.tile_queue.zero_()

  def _enqueue(self, ops, n=None):
    with torch.cuda.device(self.dev):       # launches go to the engine's device whatever is current
      self._clear_tile_queue()
      nat.check(self.lib.gc_run_program(ops, len(ops) if n is None else n, self._stream_ptr()), "gc_run_program")

  def forward(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    ops, y = self.bind(x, y)
    self._enqueue(ops)
    return y

  __call__ = forward

  def run_until(self, x: torch.Tensor, tag: str, y: Optional[torch.Tensor] = None):
    """Enqueues the step's launches up to (not including) the first launch tagged `tag` (batch
    element 0) and returns how many ran: the workspace then holds that stage boundary -- e.g.
    `run_until(x, "enc_node_mesh")` leaves the encoder's grid2mesh aggregate in `agg_mesh`.
    Verification hook (tests compare stage boundaries with the oracle); not on the product path."""
    ops, _ = self.bind(x, y)
    n = next(k for k in range(len(ops)) if ops[k].tag == TAGS[tag])
    self._enqueue(ops, n)
    return n

  def time_ops(self, x, iters=3):
    """Per-op mean milliseconds measured with HIP events on the launch stream."""
    ops, _ = self.bind(x)
    ms = (ctypes.c_float * len(ops))()
    with torch.cuda.device(self.dev):
      self._clear_tile_queue()
      nat.check(self.lib.gc_time_program(ops, len(ops), iters, ms, self._stream_ptr()), "gc_time_program")
    return [(ops[k].tag, ops[k].kind, ms[k]) for k in range(len(ops))]

  # ---------------------------------------------------------------- partitioned execution
  def segments(self, x: torch.Tensor, y: Optional[torch.Tensor] = None):
    """The step as launch segments separated by halo exchange points (partition.py).

    Returns ``(y, [(run, actions), ...])``: call ``run()`` (enqueues the segment's launches on the current
    stream), then perform ``actions`` in order -- ``("start", table)``: the launch that produced the owned
    rows of ``self.halo_table(table)`` is enqueued, the exchange of its halo suffix may begin; ``("wait", table)``:
    the next segment gathers from that suffix.  The exchange points are read off the plan's program: right in front
    of every edge update (include/gcast.h: gc_plan_program).  18 exchanges per batch element: 1 encoder, 1 per
    processor step, 1 decoder; every rank of a partitioned step has the same segment structure."""
    ops, y = self.bind(x, y)
    cuts = [(k, _EDGE_TAGS[ops[k].tag]) for k in range(len(ops))
            if ops[k].kind == nat.OP_ROWMLP and ops[k].tag in _EDGE_TAGS and ops[k].mlp.g0]
    # (`mlp.g0`: the launch that GATHERS -- every edge update of a one-hidden-layer plan, the first of the n launches
    #  an edge MLP with n hidden layers is, csrc/gcast_plan.inc: push_mlp)
    with torch.cuda.device(self.dev):
      self._clear_tile_queue()
    segs, lo = [], 0
    for hi, actions in [(k, [("start", name), ("wait", name)]) for k, name in cuts] + [(len(ops), [])]:
      sub = ctypes.cast(ctypes.byref(ops, lo * ctypes.sizeof(nat.Op)), ctypes.POINTER(nat.Op))

      def run(sub=sub, n=hi - lo, keep=ops):        # (`keep`: the array lives as long as the closure)
        if n > 0:
          with torch.cuda.device(self.dev):
            nat.check(self.lib.gc_run_program(sub, n, self._stream_ptr()), "gc_run_program")
      segs.append((run, actions))
      lo = hi
    return y, segs

  def halo_table(self, name: str) -> torch.Tensor:
    """The row table whose halo suffix the exchange `name` fills (owned prefix already valid)."""
    return self._tensor("pre_grid" if name == "g2m" else "pre_s_mesh")

  def owned_rows(self, name: str) -> int:
    return self.n_grid if name == "g2m" else self.n_mesh

  def close(self):
    if self.__dict__.get("_plan"):
      self.lib.gc_plan_destroy(self._plan)
      self._plan = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
