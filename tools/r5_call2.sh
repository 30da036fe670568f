#!/bin/bash
# round 5, GPU call 2: the dual-A contraction for dW1a / dW1b (key 8: 1 = sb_h2w, 2 = dual) and the pair launches (key 11) --
# knob tests first, then same-process A/Bs under kernel traces, the full suite, one bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_cell.py tests/test_gpu_graph.py -m gpu -q -x > $O/r5c2_targeted.log 2>&1
echo "targeted rc=$?"; tail -12 $O/r5c2_targeted.log
for spec in "8 1 2" "11 0 32 64 128 256"; do
  set -- $spec; key=$1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r5c2_kv -o r -- python $GRAFT_REPO_ROOT/tools/kv_sweep.py --key $spec --steps 10 --rounds 3 > $GRAFT_REPO_ROOT/$O/r5c2_kv$key.txt 2>&1)
  echo "kv $key rc=$?"; grep -v "^W2\|rocprof\|^E2026" $O/r5c2_kv$key.txt | tail -22
  python tools/rocpd_stats.py $O/r5c2_kv/r_results.db > $O/r5c2_kv${key}_kernel_stats.txt 2>&1; rm -rf $O/r5c2_kv
  head -14 $O/r5c2_kv${key}_kernel_stats.txt | cut -c1-75,100-175
done
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_pytest.log 2>&1
echo "full pytest rc=$?"; tail -8 $O/gpu_pytest.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-native --no-extra-legs > $O/r5c2_bench.json 2> $O/r5c2_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c2_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['roofline']['kernel_ms'], d.get('eager_step'), d.get('model_level'))
PY
