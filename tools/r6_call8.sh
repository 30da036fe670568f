#!/bin/bash
# round 6, call 8: what stage B0 of chain_bwd / stage 0 of chain_fwd cost on their own (stop-after-stage phase masks): duration from a
# kernel trace, vector / memory instruction counts and wait cycles from a PMC pass of the same runs -- the numbers behind DESIGN's
# answer to "make B0 VALU-light".  Timing only: results under a non-zero mask are wrong.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for M in 0 131072 393216 4096 12288; do       # full | bwd: stop after B0 | after B1 | fwd: stop after stage 0 | after stage 1
  rocprofv3 --kernel-trace -d $O/c8k_$M -o r -- python tools/mask_steps.py $M 4 > $O/c8k_$M.log 2>&1
  echo "== mask $M (durations)"; python tools/rocpd_stats.py $O/c8k_$M/r_results.db | grep -E "chain_(fwd|bwd)_kernel" | awk '{printf "%-72s %6s %9s %9s\n", substr($1,1,72), $2, $4, $5}'
  rm -rf $O/c8k_$M
done > $O/r06_chain_stage_times.txt 2>&1
for M in 0 131072 4096; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/c8p_$M -o p -- python tools/mask_steps.py $M 2 > $O/c8p_$M.log 2>&1
  echo "== mask $M (counters, mean per dispatch)"; python tools/pmc_summary.py $O/c8p_$M/p_counter_collection.csv 2>&1 | grep -E "kernel|chain_(fwd|bwd)_kernel" | cut -c1-260
  rm -rf $O/c8p_$M
done > $O/r06_chain_stage_pmc.txt 2>&1
cat $O/r06_chain_stage_times.txt; cat $O/r06_chain_stage_pmc.txt
