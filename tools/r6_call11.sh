#!/bin/bash
# round 6, call 11: dKB jobs on chain_bwd's idle CUs -- parity (fill 0 / 1 / 2 / 3), same-process A/B of the step, kernel stats + timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -x -k "dkb_on_idle" > $O/c11_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/c11_pytest.log
timeout 600 python tools/kv_sweep.py --key dkb_fill 0 3 2 4 --steps 20 --rounds 3 > $O/c11_sweep.txt 2>&1; tail -14 $O/c11_sweep.txt
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe --eager --steps 6 --warmup 2"
rocprofv3 --kernel-trace -d $O/c11_k -o r -- $B > $O/c11_k.log 2>&1
python tools/rocpd_stats.py $O/c11_k/r_results.db > $O/c11_kernel_stats.txt
python tools/step_timeline.py $O/c11_k/r_results.db --brief > $O/c11_timeline.txt
rm -rf $O/c11_k
head -12 $O/c11_timeline.txt
