#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/slowk
rocprofv3 --kernel-trace -d $O/slowk -o r -- python tools/slow_step_loop.py > $O/slow_step.log 2>&1
grep "fwd/bwd" $O/slow_step.log
python tools/slow_step_trace.py $O/slowk/r_results.db > $O/slow_step_trace.txt 2>&1
head -24 $O/slow_step_trace.txt | cut -c1-230
rm -rf $O/slowk
