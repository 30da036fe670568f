#!/bin/bash
# round 6, call 4: targeted tests after the epilogue-prefetch / state-view changes + the new parity tests, kernel stats, bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
MACX_RECORD_MARGINS=1 timeout 1500 python -m pytest tests/test_gpu_cell.py tests/test_gpu_knobs.py tests/test_gpu_reference_golden.py tests/test_gpu_output.py tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_units.py tests/test_gpu_unit_exports.py -m gpu -q -x > $O/c4_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/c4_pytest.log
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe"
rocprofv3 --kernel-trace -d $O/c4e -o r -- $B --eager --steps 5 --warmup 2 > $O/c4e.log 2>&1
python tools/rocpd_stats.py $O/c4e/r_results.db > $O/c4_kernel_stats.txt
python tools/step_timeline.py $O/c4e/r_results.db --brief > $O/c4_timeline.txt
rm -rf $O/c4e
head -12 $O/c4_kernel_stats.txt | cut -c1-60,100-170
head -3 $O/c4_timeline.txt
python bench.py --no-model-level --no-native --no-extra-legs > $O/c4_bench.json 2> $O/c4_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c4_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["dtype"])
print({k:d["roofline"][k] for k in ("achieved","peak","frac","kernel_ms","executed_frac","back_to_back_kernel_ms","profile_in_step_kernel_ms")})
print(d["cpu_baseline"])
PY
