#!/bin/bash
# same-box A/B of the working tree against ab_head/ (a `git archive` of an earlier commit with its own built library): kernel stats of
# the eager bench step in both trees, alternating, two rounds.   bash tools/r6_ab.sh [tag]
TAG=${1:-ab}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe --eager --steps 6 --warmup 2"
for r in 1 2; do
  for T in head new; do
    D=$GRAFT_REPO_ROOT; [ $T = head ] && D=$GRAFT_REPO_ROOT/ab_head
    (cd $D && rocprofv3 --kernel-trace -d $O/${TAG}_${T}_$r -o r -- $B > $O/${TAG}_${T}_$r.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $O/${TAG}_${T}_$r/r_results.db > $O/${TAG}_${T}_${r}_kernel_stats.txt; python $GRAFT_REPO_ROOT/tools/step_timeline.py $O/${TAG}_${T}_$r/r_results.db --brief | head -1 > $O/${TAG}_${T}_${r}_span.txt)
    rm -rf $O/${TAG}_${T}_$r
    echo "== $T round $r: $(cat $O/${TAG}_${T}_${r}_span.txt)"
    grep -E "chain_fwd|chain_bwd|chain_dkb|TOTAL" $O/${TAG}_${T}_${r}_kernel_stats.txt | awk '{printf "   %-60s %6s %10s %9s %9s\n", substr($1,1,60), $2, $3, $4, $5}'
  done
done
