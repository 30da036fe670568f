#!/bin/bash
# round 6, call 6: full -m gpu suite (batched word loads in the control kernels, the split K = p d linear, kernel-timestamp probe),
# kernel stats of the eager step, bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_pytest.log
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe"
rocprofv3 --kernel-trace -d $O/c6e -o r -- $B --eager --steps 5 --warmup 2 > $O/c6e.log 2>&1
python tools/rocpd_stats.py $O/c6e/r_results.db > $O/c6_kernel_stats.txt
python tools/step_timeline.py $O/c6e/r_results.db --brief > $O/c6_timeline.txt
rm -rf $O/c6e
head -32 $O/c6_kernel_stats.txt | cut -c1-60,100-170
head -1 $O/c6_timeline.txt
python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs > $O/c6_bench.json 2> $O/c6_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c6_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["dtype"])
print({k:d["roofline"][k] for k in ("achieved","peak","frac","kernel_ms","executed_frac","back_to_back_kernel_ms","profile_in_step_kernel_ms")})
PY
