#!/bin/bash
# round 5, GPU call 3: dW1a / dW1b from the dual-A contraction (key 8 = 2) against sb_h2w (key 8 = 1): one kernel trace per value
# (chain_fwd keeps X * y only under 2: its cost shows in chain_fwd's own row), an interleaved same-process A/B, knob tests, a bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -x > $O/r5c3_targeted.log 2>&1
echo "targeted rc=$?"; tail -4 $O/r5c3_targeted.log
for v in 1 2; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r5c3_kv -o r -- python $GRAFT_REPO_ROOT/tools/kv_sweep.py --key 8 $v --steps 10 --rounds 2 > $GRAFT_REPO_ROOT/$O/r5c3_kv8_$v.txt 2>&1)
  echo "kv 8=$v rc=$?"; grep -E "^(kv|round)" $O/r5c3_kv8_$v.txt
  python tools/rocpd_stats.py $O/r5c3_kv/r_results.db > $O/r5c3_kv8_${v}_kernel_stats.txt 2>&1; rm -rf $O/r5c3_kv
  head -9 $O/r5c3_kv8_${v}_kernel_stats.txt | cut -c1-75,100-175
done
timeout 300 python tools/kv_sweep.py --key 8 1 2 --steps 20 --rounds 4 2>&1 | grep -E "^(kv|round)" > $O/r5c3_kv8_ab.txt; cat $O/r5c3_kv8_ab.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-native --no-extra-legs > $O/r5c3_bench.json 2> $O/r5c3_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c3_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_blocks'], d['roofline']['kernel_ms'], d.get('eager_step', {}).get('ms_per_step'), d.get('model_level', {}).get('value'))
PY
