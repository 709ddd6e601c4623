#!/bin/bash
# Round-3 session 15: upper bound of what cheaper correction products could buy -- profiling libraries
# that DROP one (hdrop1) or both (hdrop3) correction MFMAs of every f16x3 product (results wrong by
# construction): single launches, then the whole step with the library swapped in (scratch copy only).
# (libraries: scripts/probes/build_probe_lib.sh f16x3_drop_correction_mfmas.patch ab_libs/libgcast_hdrop<bits>.so -DGC_H_DROP=<bits>)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-s15}
mkdir -p "$OUT"
HALF_BUILDS="drop_lo_hi:@ab_libs/libgcast_hdrop1.so;drop_both:@ab_libs/libgcast_hdrop3.so" PROBE_SHAPES=proc_edge,dec_edge_onepass,node_grid timeout 300 python -u scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe_mfma_drop.json" 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tail -5 | tee "$OUT/probe_mfma_drop.log"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check 2>&1 | grep -v amdgpu.ids | tail -1 > "$OUT/bench_f16x3_product.json"
for v in 1 3; do
  cp ab_libs/libgcast_hdrop$v.so graphcast_amd/csrc/libgcast_hip.so
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check 2>&1 | grep -v amdgpu.ids | tail -1 > "$OUT/bench_f16x3_hdrop$v.json"
done
