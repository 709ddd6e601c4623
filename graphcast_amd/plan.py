"""`NativePlan`: the step driven through the C-ABI's plan API exactly as a C / C++ host drives it (include/gcast.h:
gc_plan_create / gc_plan_workspace_bytes / gc_step_forward / gc_plan_check_range) -- two calls per step, an opaque
workspace.  ``engine.StepEngine`` sits on the same plan and the same launch program (gc_plan_program), with the
Python host's verification and partitioning hooks around it; the Python side here only flattens the reference-layout
arrays into the C descriptors.
"""
import ctypes
from typing import Mapping, Optional

import torch

from graphcast_amd import _native as nat
from graphcast_amd.engine import create_plan, tensor_descs      # noqa: F401  (one flattening of the reference-layout arrays)


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
    handle = create_plan(self.lib, graphs, params, num_steps=num_steps, c_in=c_in, c_out=c_out, precision=precision,
                         device=self.dev)
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
