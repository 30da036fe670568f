#!/bin/bash
# round 5, GPU call 8: chain kernels with the first weight slice of every product requested ahead of the preceding epilogue (key 7 = 255 -> the default variants) against key 7 = 68
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_cell.py -m gpu -q > $O/r5c9_targeted.log 2>&1
echo "targeted rc=$?"; tail -6 $O/r5c9_targeted.log
for v in 68 255; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r5c9_kv -o r -- python $GRAFT_REPO_ROOT/tools/kv_sweep.py --key 7 $v --steps 10 --rounds 2 > $GRAFT_REPO_ROOT/$O/r5c9_kv7_$v.txt 2>&1)
  echo "kv 13=$v rc=$?"; grep -E "^(kv|round)" $O/r5c9_kv7_$v.txt
  python tools/rocpd_stats.py $O/r5c9_kv/r_results.db > $O/r5c9_kv7_${v}_kernel_stats.txt 2>&1; rm -rf $O/r5c9_kv
  grep -E "chain_bwd|chain_fwd|kernel  " $O/r5c9_kv7_${v}_kernel_stats.txt | cut -c1-75,100-175
done
timeout 300 python tools/kv_sweep.py --key 7 68 255 --steps 20 --rounds 4 2>&1 | grep -E "^(kv|round)" > $O/r5c9_kv7_ab.txt; cat $O/r5c9_kv7_ab.txt
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_graph.py -m gpu -q > $O/r5c9_configs.log 2>&1
echo "configs rc=$?"; tail -4 $O/r5c9_configs.log
