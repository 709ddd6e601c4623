"""`NativePlan`: the step driven through the C-ABI's plan API (include/gcast.h:
gc_plan_create / gc_step_forward) instead of the Python plan builder of engine.py.

Same launches, same packed images -- the C++ packers in csrc/gcast_plan.inc mirror packing.py and
StepEngine bit for bit (tests/test_plan_gpu.py) -- so this is what a C / C++ host of the library
gets.  The Python side only flattens the reference-layout arrays into the C descriptors.
"""
import ctypes
from typing import Mapping, Optional

import numpy as np
import torch

from graphcast_amd import _native as nat


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


class NativePlan:
  """x [N_grid, B, C_in] fp32 (device) -> y [N_grid, B, C_out] fp32 (device)."""

  def __init__(self, graphs: Mapping, params: Mapping, *, num_steps: int, c_in: int, c_out: int,
               device="cuda:0", precision: str = "f16x3", half=None):
    self.lib = nat.lib()
    self.dev = torch.device(device)
    self.c_in, self.c_out = c_in, c_out
    self.n_grid = int(graphs["n_grid"])
    if half is False and precision == "f16x3":
      raise ValueError("half=False: the chunked f16x3 kernels were retired in round 5")
    self.half = precision in ("f16x3", "bf16")       # the half-N formulation (f32: the chunked exact-fp32 kernel)
    keep = []

    def edge_set(g):
      s, r, f = _i32(g["senders"]), _i32(g["receivers"]), _f32(g["feat"])
      keep.extend([s, r, f])
      return nat.EdgeSet(len(s), s.ctypes.data, r.ctypes.data, f.ctypes.data, f.shape[1])

    gnf, mnf = _f32(graphs["grid_node_feat"]), _f32(graphs["mesh_node_feat"])
    keep += [gnf, mnf]
    model = nat.ModelDesc(self.n_grid, int(graphs["n_mesh"]), c_in, c_out, gnf.shape[1], num_steps,
                          nat.PRECISIONS[precision], gnf.ctypes.data, mnf.ctypes.data,
                          edge_set(graphs["g2m"]), edge_set(graphs["mesh"]), edge_set(graphs["m2g"]),
                          nat.LAYOUT_HALF if self.half else nat.LAYOUT_CHUNKED)
    tensors, keep_t = tensor_descs(params)
    handle = ctypes.c_void_p()
    with torch.cuda.device(self.dev):
      stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
      nat.check(self.lib.gc_plan_create(ctypes.byref(model), tensors, len(tensors), stream,
                                        ctypes.byref(handle)), "gc_plan_create")
    self._plan = handle
    self._ws: Optional[torch.Tensor] = None

  def forward(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype != torch.float32 or x.dim() != 3 or not x.is_contiguous() or x.device != self.dev:
      raise ValueError("x must be a contiguous float32 [N_grid, B, C_in] tensor on the plan's device")
    if x.shape[0] != self.n_grid or x.shape[2] != self.c_in:
      raise ValueError(f"x has shape {tuple(x.shape)}, expected [{self.n_grid}, B, {self.c_in}]")
    batch = x.shape[1]
    if y is None:
      y = torch.empty((self.n_grid, batch, self.c_out), dtype=torch.float32, device=self.dev)
    elif (tuple(y.shape) != (self.n_grid, batch, self.c_out) or y.dtype != torch.float32
          or not y.is_contiguous() or y.device != self.dev):
      # (goes to the library as a raw pointer: a wrong y would be written out of bounds)
      raise ValueError("y must be a contiguous float32 [N_grid, B, C_out] tensor on the plan's device")
    need = self.lib.gc_plan_workspace_bytes(self._plan, batch)
    with torch.cuda.device(self.dev):       # the launches go to THIS device's stream, whatever is current
      if self._ws is None or self._ws.numel() < need:
        self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
      stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
      nat.check(self.lib.gc_step_forward(self._plan, x.data_ptr(), y.data_ptr(), batch, self._ws.data_ptr(),
                                         self._ws.numel(), stream), "gc_step_forward")
    return y

  __call__ = forward

  def check_range(self):
    """gc_plan_check_range: synchronises the launch stream and raises GcastRangeError if the last step read an
    input value outside the exact range of the f16x3 arithmetic (|x| > 65504)."""
    if self._ws is None:
      return
    with torch.cuda.device(self.dev):
      stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
      rc = self.lib.gc_plan_check_range(self._plan, self._ws.data_ptr(), stream)
    if rc == nat.ERANGE:
      raise nat.GcastRangeError(self.lib.gc_last_error().decode())
    nat.check(rc, "gc_plan_check_range")

  def close(self):
    if self._plan:
      self.lib.gc_plan_destroy(self._plan)
      self._plan = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
