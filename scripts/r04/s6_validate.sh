#!/bin/bash
# Round-4 session 6: the new / changed GPU tests, the bench line with the rollout extra, bench --mode partition at N = 1.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
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
