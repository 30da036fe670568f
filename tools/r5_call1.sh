#!/bin/bash
# round 5, GPU call 1: grid-barrier probe, the tests the round's changes touch, same-process A/B of the weight-gradient pipeline
# (macx_debug_set(10, v)) under a kernel trace, the stem's convolution A/B, the full -m gpu suite, one bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/probes/bin/grid_barrier_probe > $O/r05_grid_barrier_probe.txt 2>&1; echo "probe rc=$?"; cat $O/r05_grid_barrier_probe.txt
timeout 1200 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_unit_exports.py tests/test_gpu_h2.py tests/test_gpu_stem.py tests/test_gpu_dp.py tests/test_gpu_limits.py -m gpu -q > $O/r5c1_targeted.log 2>&1
echo "targeted rc=$?"; tail -25 $O/r5c1_targeted.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r5c1_kv -o r -- python $GRAFT_REPO_ROOT/tools/kv_sweep.py --key 10 0 1 2 --steps 10 --rounds 3 > $GRAFT_REPO_ROOT/$O/r5c1_kv10.txt 2>&1)
echo "kv rc=$?"; grep -v "^W2\|rocprof" $O/r5c1_kv10.txt | tail -16
python tools/rocpd_stats.py $O/r5c1_kv/r_results.db > $O/r5c1_kv10_kernel_stats.txt 2>&1; rm -rf $O/r5c1_kv
head -12 $O/r5c1_kv10_kernel_stats.txt | cut -c1-70,100-170
timeout 300 python tools/stem_chain_ab.py > $O/r5c1_stem_ab.txt 2>&1; cat $O/r5c1_stem_ab.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_pytest.log 2>&1
echo "full pytest rc=$?"; tail -8 $O/gpu_pytest.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-native > $O/r5c1_bench.json 2> $O/r5c1_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c1_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['timing'], d['roofline']['kernel_ms'], d.get('eager_step'), d.get('model_level'))
PY
