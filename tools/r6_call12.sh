#!/bin/bash
# round 6, call 12: what a dKB job is made of -- the closing launch and chain_bwd (with fillers) under the job's timing knobs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for K in 0 1 2 4 8 16 5 13; do
  M=$((K << 22))
  rocprofv3 --kernel-trace -d $O/c12_$K -o r -- python tools/mask_steps.py $M 6 > $O/c12_$K.log 2>&1
  python tools/rocpd_stats.py $O/c12_$K/r_results.db > $O/c12_${K}_kernel_stats.txt
  rm -rf $O/c12_$K
  echo "== job knobs $K"
  grep -E "chain_bwd|chain_dkb|chain_fwd" $O/c12_${K}_kernel_stats.txt | awk '{printf "   %-50s %6s %10s %9s %9s %9s\n", substr($1,9,50), $2, $3, $4, $5, $6}'
done
