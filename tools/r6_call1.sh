#!/bin/bash
# round 6, call 1: the start-of-round state on one box -- bench line, kernel stats of the eager step, one step's launch timeline
# (eager and replayed graph)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe"
rocprofv3 --kernel-trace -d $O/c1e -o r -- $B --eager --steps 5 --warmup 2 > $O/c1e.log 2>&1
python tools/rocpd_stats.py $O/c1e/r_results.db > $O/r06_start_kernel_stats.txt
python tools/step_timeline.py $O/c1e/r_results.db > $O/r06_start_timeline_eager.txt
rocprofv3 --kernel-trace -d $O/c1g -o r -- $B --steps 5 --warmup 2 > $O/c1g.log 2>&1
python tools/step_timeline.py $O/c1g/r_results.db > $O/r06_start_timeline_graph.txt
rm -rf $O/c1e $O/c1g
python bench.py --no-model-level --no-native --no-extra-legs > $O/r06_start_bench.json 2> $O/r06_start_bench.err
head -30 $O/r06_start_kernel_stats.txt | cut -c1-60,100-170
tail -40 $O/r06_start_timeline_graph.txt
cut -c1-400 $O/r06_start_bench.json
