#!/bin/bash
# one GPU call: the mask-word tests (dropout stream, fused cell, generic cell, captured training step) + the graph tests + a bench step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_units.py tests/test_gpu_graph.py tests/test_gpu_cell.py tests/test_gpu_generic.py tests/test_gpu_unit_exports.py -x -q -k "not random_option" > gpurun_out/word_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/word_pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-model-level --no-native --no-extra-legs > gpurun_out/word_bench.json 2> gpurun_out/word_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/word_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d.get('train_step_graph'))"
