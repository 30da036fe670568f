#!/bin/bash
# one GPU call: the parity tests that cover the cell's kernels + the graph captures, the train-step capture probe in NPROC fresh
# processes, then the bench step (no side legs unless BENCH_FLAGS says otherwise)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
TESTS=${TESTS:-"tests/test_gpu_graph.py tests/test_gpu_cell.py tests/test_gpu_reference_golden.py tests/test_gpu_unit_exports.py tests/test_gpu_units.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py"}
timeout 1500 python -m pytest $TESTS -x -q > gpurun_out/check_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/check_pytest.log
: > gpurun_out/graph_probe_train.log
for i in $(seq ${NPROC:-10}); do timeout 120 python tools/graph_replay_probe_train.py 2>&1 | tail -1 >> gpurun_out/graph_probe_train.log; done
sort gpurun_out/graph_probe_train.log | uniq -c
FL="--steps 20 --warmup 3 --no-cpu-baseline --no-model-level --no-native --no-extra-legs $BENCH_FLAGS"
timeout 300 python bench.py $FL > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; echo "bench rc=$?"
tail -c 1200 gpurun_out/check_bench.json
