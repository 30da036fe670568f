#!/bin/bash
# one GPU call: side-queue modes (macx_debug_set(6, .)) side by side: parity vs mode 0, ms per step, kernel stats of mode 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python tools/kv_sweep.py 0 4 --key 6 --steps 20 --rounds 3 > $O/ov_sweep.log 2>&1; echo "sweep rc=$?"
grep -v "^ref" $O/ov_sweep.log | tail -12
MACX_OVERLAP=4 timeout 600 rocprofv3 --kernel-trace --stats -d $O/kv_k -o r -- python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --steps 6 --warmup 2 > $O/ov_k.log 2>&1
python tools/rocpd_stats.py $O/kv_k/r_results.db > $O/ov4_kernel_stats.txt
rm -rf $O/kv_k
head -14 $O/ov4_kernel_stats.txt | cut -c1-75,100-160
MACX_OVERLAP=4 timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_cell.py -x -q > $O/ov_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/ov_pytest.log
