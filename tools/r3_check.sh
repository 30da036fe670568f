#!/bin/bash
# one GPU call: the parity tests that cover the cell's kernels, then the bench step (no side legs)
#   TESTS="tests/test_gpu_cell.py ..." overrides the test list; BENCH_FLAGS adds bench.py flags
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
TESTS=${TESTS:-"tests/test_gpu_cell.py tests/test_gpu_reference_golden.py tests/test_gpu_unit_exports.py tests/test_gpu_units.py tests/test_gpu_configs.py::test_metric_configuration_all_gradients_fp64 tests/test_gpu_fuzz.py"}
timeout 1200 python -m pytest $TESTS -x -q > gpurun_out/check_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/check_pytest.log
FL="--steps 20 --warmup 3 --no-cpu-baseline --no-model-level --no-native --no-extra-legs $BENCH_FLAGS"
timeout 300 python bench.py $FL > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/check_bench.json
