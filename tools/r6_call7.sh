#!/bin/bash
# round 6, call 7: the captured data-parallel step (GPU test with two gloo ranks on the one GPU; bench --gpus 2 over gloo), whole-tower kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_graph.py -m gpu -q -x > $O/c7_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/c7_pytest.log
MACX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-extra-dp > $O/c7_bench_gloo2.json 2> $O/c7_bench_gloo2.err; echo "gloo2 rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/c7_bench_gloo2.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["n_gpus"]); print(d.get("launch")); print(d.get("launch_note")); print(d.get("eager_step"))
except Exception as e:
    print("no line", e); print(open("gpurun_out/c7_bench_gloo2.err").read()[-1500:])
PY
bash tools/model_level_kstats.sh c7 2>&1 | tail -50
