#!/bin/bash
# round 6, call 16: the whole [B,d] tail (attention over the KB, write unit, y) on chain_fwd's filler workgroups: parity, A/B 0 2 1, kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_cell.py tests/test_gpu_graph.py -m gpu -q -x > $O/c20_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/c20_pytest.log
timeout 600 python tools/kv_sweep.py --key pre_fill 0 2 1 --steps 30 --rounds 4 > $O/c20_sweep.txt 2>&1; tail -8 $O/c20_sweep.txt
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe --eager --steps 6 --warmup 2"
rocprofv3 --kernel-trace -d $O/c20_k -o r -- $B > $O/c20_k.log 2>&1
python tools/rocpd_stats.py $O/c20_k/r_results.db > $O/c20_kernel_stats.txt
python tools/step_timeline.py $O/c20_k/r_results.db --brief > $O/c20_timeline.txt
rm -rf $O/c20_k
head -14 $O/c20_timeline.txt
head -4 $O/c20_kernel_stats.txt | cut -c1-60,84-150
