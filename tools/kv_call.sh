#!/bin/bash
# one GPU call: sb_h2 128x128 (key 8 = 0) vs sb_h2w 128x256 (1): parity vs each other, ms/step, kernel stats; then parity tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python tools/kv_sweep.py 0 1 --key 8 --steps 20 --rounds 3 > $O/sbw_sweep.log 2>&1; echo "sweep rc=$?"
grep -v "^ref" $O/sbw_sweep.log | tail -10
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kv_k -o r -- python tools/kv_sweep.py 0 1 --key 8 --steps 4 --rounds 1 > $O/sbw_k.log 2>&1
python tools/rocpd_stats.py $O/kv_k/r_results.db > $O/sbw_kernel_stats.txt
rm -rf $O/kv_k
grep "sb_h2\|slab_reduce" $O/sbw_kernel_stats.txt | cut -c1-75,100-160
timeout 1200 python -m pytest tests/test_gpu_limits.py tests/test_gpu_configs.py tests/test_gpu_cell.py tests/test_gpu_fuzz.py -x -q > $O/sbw_pytest.log 2>&1; echo "pytest rc=$?"
tail -6 $O/sbw_pytest.log
