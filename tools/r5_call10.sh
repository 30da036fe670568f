#!/bin/bash
# round 5, GPU call 10: the metric's step under --relu STD / readCtrlAct TANH / NON (the chain_bwd variants that used to spill 102 registers)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r5c10 -o r -- python $GRAFT_REPO_ROOT/tools/relu_variants.py > $GRAFT_REPO_ROOT/$O/r5c10_relu.txt 2>&1)
grep "ms per step" $O/r5c10_relu.txt
python tools/rocpd_stats.py $O/r5c10/r_results.db > $O/r5c10_kernel_stats.txt 2>&1; rm -rf $O/r5c10
grep -E "chain_bwd|chain_fwd|kernel  " $O/r5c10_kernel_stats.txt | cut -c1-75,100-175
