#!/bin/bash
# the whole -m gpu suite, log kept: gpurun_out/gpu_pytest.log (copied to profiles/<round>_gpu_pytest.log), then smoke() and the bench step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/gpu_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/gpu_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-model-level --no-native --no-extra-legs > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/check_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
# the N > 1 path on one GPU: two gloo ranks share it (not a scaling number)
MACX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-extra-dp > gpurun_out/check_bench_gloo2.json 2> gpurun_out/check_bench_gloo2.err; echo "gloo2 rc=$?"
tail -c 400 gpurun_out/check_bench_gloo2.json | head -c 400; echo
