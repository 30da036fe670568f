#!/bin/bash
# round 5: one quick A/B call -- bash tools/r5_quick.sh <tag> <key> <v0> <v1> [kernel-name regex]
#   knob + cell tests, a kernel trace per value (tools/kv_sweep.py --key K v), an interleaved same-process A/B, the full-size parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
TAG=$1; KEY=$2; V0=$3; V1=$4; RX=${5:-chain_|wgrad_h2|sb_h2w|kb_gemm_h2}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_cell.py -m gpu -q > $O/${TAG}_targeted.log 2>&1
echo "targeted rc=$?"; tail -3 $O/${TAG}_targeted.log
for v in $V0 $V1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/${TAG}_kv -o r -- python $GRAFT_REPO_ROOT/tools/kv_sweep.py --key $KEY $v --steps 10 --rounds 2 > $GRAFT_REPO_ROOT/$O/${TAG}_kv${KEY}_$v.txt 2>&1)
  echo "key $KEY = $v rc=$?"; grep -E "^(kv|round)" $O/${TAG}_kv${KEY}_$v.txt
  python tools/rocpd_stats.py $O/${TAG}_kv/r_results.db > $O/${TAG}_kv${KEY}_${v}_kernel_stats.txt 2>&1; rm -rf $O/${TAG}_kv
  grep -E "$RX|kernel  " $O/${TAG}_kv${KEY}_${v}_kernel_stats.txt | cut -c1-75,100-175
done
timeout 300 python tools/kv_sweep.py --key $KEY $V0 $V1 --steps 20 --rounds 4 2>&1 | grep -E "^(kv|round)" > $O/${TAG}_kv${KEY}_ab.txt; cat $O/${TAG}_kv${KEY}_ab.txt
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q > $O/${TAG}_configs.log 2>&1
echo "configs rc=$?"; tail -3 $O/${TAG}_configs.log
