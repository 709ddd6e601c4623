#!/usr/bin/env python
"""Host<->device copy rates on the GPU box (what bounds GraphCast.__call__ on host Datasets): pageable and pinned
H2D / D2H of a 1 GiB fp32 block, and the host-side copy into pinned pages."""
import json
import time
import numpy as np
import torch

def t(fn, n=3):
  fn(); torch.cuda.synchronize()
  ts = []
  for _ in range(n):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  return min(ts)

n = 1 << 28
src = torch.from_numpy(np.random.default_rng(0).standard_normal(n, dtype=np.float32))
pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
dev = torch.empty(n, dtype=torch.float32, device="cuda:0")
gb = 4 * n / 1e9
res = {"threads": torch.get_num_threads(), "GB": gb,
       "pageable_h2d_GBps": gb / t(lambda: dev.copy_(src)),
       "host_to_pinned_GBps": gb / t(lambda: pin.copy_(src)),
       "pinned_h2d_GBps": gb / t(lambda: dev.copy_(pin, non_blocking=True)),
       "pinned_d2h_GBps": gb / t(lambda: pin.copy_(dev, non_blocking=True)),
       "pageable_d2h_GBps": gb / t(lambda: src.copy_(dev)),
       "pinned_alloc_cached_s": t(lambda: torch.empty(n, dtype=torch.float32, pin_memory=True))}
def chunked():
  k = 8; m = n // k
  for i in range(k):
    pin[i*m:(i+1)*m].copy_(src[i*m:(i+1)*m]); dev[i*m:(i+1)*m].copy_(pin[i*m:(i+1)*m], non_blocking=True)
res["chunked_pipeline_h2d_GBps"] = gb / t(chunked)
print(json.dumps(res))

# where a pipeline of pieces loses the rate: 8 x 128 MiB
k = 8; m = n // k
sl = [slice(i * m, (i + 1) * m) for i in range(k)]
def dma_nb():
  for s in sl: dev[s].copy_(pin[s], non_blocking=True)
def dma_sync():
  for s in sl: dev[s].copy_(pin[s]); torch.cuda.synchronize()
def host_only():
  for s in sl: pin[s].copy_(src[s])
def pageable_pieces():
  for s in sl: dev[s].copy_(src[s])
def interleaved_blocking():
  for s in sl: pin[s].copy_(src[s]); dev[s].copy_(pin[s])
def two_phase():
  for s in sl: pin[s].copy_(src[s])
  for s in sl: dev[s].copy_(pin[s], non_blocking=True)
pieces = {f.__name__: gb / t(f) for f in (dma_nb, dma_sync, host_only, pageable_pieces, interleaved_blocking, two_phase)}
# fresh (never-copied) pageable arrays, one per call: what a caller's new Dataset looks like
fresh = [torch.from_numpy(np.ones(m, dtype=np.float32) * i) for i in range(4 * k)]
it = iter(fresh)
def fresh_pageable():
  for _ in range(k): dev[sl[0]].copy_(next(it))
t0 = time.perf_counter(); fresh_pageable(); torch.cuda.synchronize(); pieces["fresh_pageable_first_touch"] = gb / (time.perf_counter() - t0)
it = iter(fresh[k:])
def fresh_via_pinned():
  for i in range(k): pin[sl[i]].copy_(next(it)); dev[sl[i]].copy_(pin[sl[i]], non_blocking=True)
t0 = time.perf_counter(); fresh_via_pinned(); torch.cuda.synchronize(); pieces["fresh_via_pinned"] = gb / (time.perf_counter() - t0)
print(json.dumps({"pieces_GBps": pieces}))
