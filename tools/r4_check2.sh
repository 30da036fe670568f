#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_gpu_reference_golden.py tests/test_gpu_stem.py tests/test_gpu_encoder.py tests/test_gpu_output.py -x -q > gpurun_out/check2_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/check2_pytest.log
MACX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/check2_gloo2.json 2> gpurun_out/check2_gloo2.err; echo "gloo2 rc=$?"
tail -c 1500 gpurun_out/check2_gloo2.json; tail -3 gpurun_out/check2_gloo2.err
timeout 900 python bench.py --no-cpu-baseline --no-native > gpurun_out/check2_bench.json 2> gpurun_out/check2_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/check2_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('fwd_only_p4','train_b128_p12_adam_ema','train_step_graph','gqa_shape_p4_args3','gqa_shape_p4_args4','model_level'):
    v=d.get(k); print(k, v and {a:v[a] for a in v if a in ('value','ms_per_step','ms_per_batch','graph_replay','launch')})
PY
