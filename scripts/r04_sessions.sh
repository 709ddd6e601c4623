#!/bin/bash
# The GPU-box sessions of round 4 in ONE file (no more one-off scripts/s<N>_session.sh): `bash
# scripts/r04_sessions.sh <name>` runs one of them from the repository root; results land under gpurun_out/r04_<name>/,
# what is kept was copied to profiles/r04_<name>_*.  Sessions that only ran scripts/session.sh (bench A/B of environment
# switches) or one command are listed at the bottom as the command itself.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NAME=${1:?session name}
case "$NAME" in
  s1)
    # Round-4 session 1: A/B of the persistent launches' tile -> workgroup map (GCAST_TILE_MAP=xcd) and of
    # one workgroup per CU (GCAST_GRID_CAP=256), each as its own bench.py process of the headline workload.
    OUT=gpurun_out/r04_s1; mkdir -p "$OUT"
    B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check"
    run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 400 $B > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "rc=$?"; python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], {k: round(v["ms"], 3) for k, v in j["roofline"]["stages"].items()})
PY
    }
    run default GCAST_X=0
    run xcd GCAST_TILE_MAP=xcd
    run cap256 GCAST_GRID_CAP=256
    run cap256_xcd GCAST_GRID_CAP=256 GCAST_TILE_MAP=xcd
    run default2 GCAST_X=0
    echo "== parity with the xcd map"
    GCAST_TILE_MAP=xcd timeout 600 python -m pytest tests/test_step_gpu.py tests/test_rowmlp_gpu.py -m gpu -q -x 2>&1 | tail -3
    ;;
  s3)
    # Round-4 session 3: the eight-wave "helper waves" form of the half-N launches (GCAST_HELPERS=1): parity gate, then A/B bench.
    OUT=gpurun_out/r04_s3; mkdir -p "$OUT"
    echo "== parity, GCAST_HELPERS=1 (per-launch tests first: a hang costs 120 s, not the session)"
    GCAST_HELPERS=1 timeout 240 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=100 2>&1 | tail -4 | tee "$OUT/pytest_rowmlp.log"
    grep -q " passed" "$OUT/pytest_rowmlp.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest_rowmlp.log" || { echo "GATE: per-launch parity failed"; exit 1; }
    GCAST_HELPERS=1 timeout 400 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_deepgnn_gpu.py -m gpu -q -x --timeout=200 2>&1 | tail -4 | tee "$OUT/pytest_step.log"
    grep -q " passed" "$OUT/pytest_step.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest_step.log" || { echo "GATE: step parity failed"; exit 1; }
    bash scripts/session.sh bench-ab r04_s3 "GCAST_HELPERS=0" "GCAST_HELPERS=1" "GCAST_HELPERS=1 GCAST_TILE_MAP=xcd" "GCAST_HELPERS=0"
    ;;
  s4)
    # Round-4 session 4: phase traces (wave 0 = a multiplying wave) of the same launches in pair / lone / helper-wave form.
    OUT=gpurun_out/r04_s4; mkdir -p "$OUT"
    run() { tag=$1; shift; echo "== $tag"; env "$@" HALF_TRACE=1 PROBE_SHAPES=proc_edge,gemm_only_mlp,node_grid timeout 300 python -u scripts/half_probe.py --out "$OUT/trace_$tag.json" 2>&1 | grep htrace | cut -c1-900; }
    run pair GCAST_HELPERS=0
    run lone GCAST_HELPERS=0 GCAST_GRID_CAP=256
    run helpers GCAST_HELPERS=1
    ;;
  s5)
    # Round-4 session 5: the per-CU GEMM-phase semaphore (GCAST_GEMM_MUTEX=1): parity gate, A/B bench, phase trace.
    OUT=gpurun_out/r04_s5; mkdir -p "$OUT"
    GCAST_GEMM_MUTEX=1 timeout 400 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q -x --timeout=150 2>&1 | tail -4 | tee "$OUT/pytest.log"
    grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
    bash scripts/session.sh bench-ab r04_s5 "GCAST_GEMM_MUTEX=0" "GCAST_GEMM_MUTEX=1" "GCAST_GEMM_MUTEX=0" "GCAST_GEMM_MUTEX=1"
    echo "== trace with the semaphore"
    GCAST_GEMM_MUTEX=1 HALF_TRACE=1 PROBE_SHAPES=proc_edge,gemm_only_mlp,node_grid timeout 300 python -u scripts/half_probe.py --out "$OUT/trace_mutex.json" 2>&1 | grep htrace | cut -c1-900
    ;;
  s6)
    # Round-4 session 6: the new / changed GPU tests, the bench line with the rollout extra, bench --mode partition at N = 1.
    OUT=gpurun_out/r04_s6; mkdir -p "$OUT"
    timeout 1500 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_partition_gpu.py tests/test_deepgnn_gpu.py tests/test_bf16_tier_gpu.py tests/test_rollout_gpu.py -m gpu -q --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
    grep -E "passed|failed|BF16_TIER_ERROR_SHAPE|NCCL_WS1|C host|in-range|Error|error" "$OUT/pytest.log" | tail -25 | cut -c1-400
    echo "== bench (rollout extra, no cpu baseline)"
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$?"
    python - "$OUT/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], j.get("rollout"), j["roofline"]["traffic_detail"], j["roofline"]["pmc"], j["cross_check"])
PY
    echo "== bench --mode partition (N = 1: a single-rank RCCL group)"
    timeout 600 python bench.py --mode partition --steps 5 --warmup 2 > "$OUT/bench_partition1.json" 2> "$OUT/bench_partition1.err"; echo "rc=$?"; cut -c1-600 "$OUT/bench_partition1.json"; tail -3 "$OUT/bench_partition1.err"
    echo "== the same under torchrun with WORLD_SIZE=1 (what the driver's N = 1 scaling point does)"
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_torchrun1.json" 2> "$OUT/bench_torchrun1.err"; echo "rc=$?"; cut -c1-300 "$OUT/bench_torchrun1.json"; tail -2 "$OUT/bench_torchrun1.err"
    ;;
  s7)
    # Round-4 session 7: the helper-wave form (now with the parked accumulators in LDS) on the big node-side launches only.
    OUT=gpurun_out/r04_s7; mkdir -p "$OUT"
    GCAST_HELPERS=1 timeout 300 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=100 -k "chain or persistent or mlp_ln or edge_block" 2>&1 | tail -3 | tee "$OUT/pytest.log"
    grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
    bash scripts/session.sh bench-ab r04_s7 "GCAST_HELPERS_MIN_ROWS=0" "GCAST_HELPERS_MIN_ROWS=65536" "GCAST_HELPERS=1" "GCAST_HELPERS_MIN_ROWS=0" "GCAST_HELPERS_MIN_ROWS=65536"
    ;;
  s10)
    # Round-4 session 10: VALU bursts of the layer-1 loops woven behind the previous chunk's MFMAs (GC_H_WEAVE, default) vs bursts
    # (ab_libs/libgcast_noweave.so = -DGC_H_WEAVE=0): parity gate, A/B bench.
    OUT=gpurun_out/${1:-r04_s10}; mkdir -p "$OUT"
    timeout 600 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q -x --timeout=200 2>&1 | tail -3 | tee "$OUT/pytest.log"
    grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
    bash scripts/session.sh bench-ab ${1:-r04_s10} "GCAST_LIB_PATH=ab_libs/libgcast_noweave.so" "GCAST_X=weave" "GCAST_LIB_PATH=ab_libs/libgcast_noweave.so" "GCAST_X=weave"
    ;;
  s12)
    # Round-4 session 12: upper bound of routing the layer-1 row fragments through LDS-DMA: a profiling library that never
    # re-loads them (results wrong), pair and helper form, single-launch timings.
    OUT=gpurun_out/r04_s12; mkdir -p "$OUT"
    for H in 0 1; do
      echo "== GCAST_HELPERS=$H"
      GCAST_HELPERS=$H HALF_BUILDS="norows:@ab_libs/libgcast_norows.so" PROBE_SHAPES=proc_edge,gemm_only_mlp,node_grid timeout 300 python -u scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe_helpers$H.json" 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -4
    done
    ;;
  s13)
    # Round-4 session 13: launches of at most one tile per CU in the helper form (GCAST_HELPERS_SMALL, default on): parity on the
    # small-graph suites (every launch there is small), then the emulated 8-way partition A/B and the 1 deg step.
    OUT=gpurun_out/r04_s13; mkdir -p "$OUT"
    timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_partition_gpu.py tests/test_conditioned_gpu.py tests/test_deepgnn_gpu.py -m gpu -q -x --timeout=300 2>&1 | tail -3 | tee "$OUT/pytest.log"
    grep -q " passed" "$OUT/pytest.log" && ! grep -q "failed\|rror\|Timeout" "$OUT/pytest.log" || { echo "GATE: parity failed"; exit 1; }
    for v in 0 1; do
      echo "== partition emulated 8-way, GCAST_HELPERS_SMALL=$v"
      GCAST_HELPERS_SMALL=$v timeout 900 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_small$v.json" > "$OUT/partition8_small$v.log" 2>&1; echo "rc=$?"; tail -1 "$OUT/partition8_small$v.log" | cut -c1-900
    done
    for v in 0 1; do
      echo "== bench 1deg_13L_M5, GCAST_HELPERS_SMALL=$v"
      GCAST_HELPERS_SMALL=$v timeout 300 python bench.py --config 1deg_13L_M5 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 2>/dev/null | python -c "
    import json,sys
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], {k: round(v['ms'],3) for k,v in j['roofline']['stages'].items()})"
    done
    ;;
  *) echo "unknown session $NAME"; exit 2;;
esac
# One-command sessions of the round:
#   s2   scripts/ubench/gh_skeleton <variant 0..13>            (hipcc --offload-arch=gfx950 -O3 scripts/ubench/gh_skeleton.hip)
#   s8   scripts/session.sh bench-ab r04_s8 "GCAST_MUL_PRIO=0" "GCAST_MUL_PRIO=1" ...     (s_setprio probe, removed again)
#   s9   python -m pytest tests/test_conditioned_gpu.py tests/test_deepgnn_gpu.py tests/test_plan_gpu.py -m gpu
#   s11  = s10 with scripts/probes/f16x3_weave_valu_bursts.patch's one-pass weave built in
#   s14  ADV_LIBS="rows2:@...;rows4:@...;rows16:@..." python scripts/advance_probe.py     (-DGC_ADV_ROWS=<n> builds)
#   final / final2   bash scripts/final_session.sh r04_final
#   s18 / s19  python scripts/host_boundary_bench.py   (host-stacked Datasets; then per-variable upload + device stacking)
#              python scripts/probes/pcie_probe.py      (pageable / pinned H2D, D2H rates of the box)
#   s20  python scripts/probes/hipgraph_step_probe.py   (the step replayed as one hipGraph vs eager enqueue)
#   s21  HALF_TRACE=1 PROBE_QUEUE=0 PROBE_SHAPES=proc_edge,node_grid,dec_edge python -u scripts/half_probe.py   (static walk: per-workgroup finish times)
#   s22  = s21 with PROBE_QUEUE=1 + scripts/session.sh bench-ab r04_s22 "GCAST_TILE_QUEUE=0" "GCAST_TILE_QUEUE=1" ... (queue on every launch with a second round)
#   s23  scripts/session.sh bench-ab r04_s23 "GCAST_TILE_QUEUE=0" "GCAST_TILE_QUEUE=1" ...   (queue from 4 tiles per workgroup on: shipped)
#   final3   bash scripts/final_session.sh r04_final3   (-> profiles/r04_final_*, current_*; the static-walk build's kept as r04_final2_static_walk_*)
