#!/bin/bash
# round 6, call 13: pipelined dKB jobs -- parity, job knobs, sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -x -k "dkb_on_idle" > $O/c13_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/c13_pytest.log
for K in 0 1 4 8 13; do
  M=$((K << 22))
  rocprofv3 --kernel-trace -d $O/c13_$K -o r -- python tools/mask_steps.py $M 6 > $O/c13_$K.log 2>&1
  python tools/rocpd_stats.py $O/c13_$K/r_results.db > $O/c13_${K}_kernel_stats.txt
  rm -rf $O/c13_$K
  echo "== job knobs $K"
  grep -E "chain_bwd|chain_dkb|chain_fwd" $O/c13_${K}_kernel_stats.txt | awk '{printf "   %-50s %6s %10s %9s %9s %9s\n", substr($1,9,50), $2, $3, $4, $5, $6}'
done
timeout 600 python tools/kv_sweep.py --key dkb_fill 0 3 2 --steps 30 --rounds 4 > $O/c13_sweep.txt 2>&1; tail -12 $O/c13_sweep.txt
