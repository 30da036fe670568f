#!/bin/bash
# kernel-trace statistics of a short bench run: gpurun_out/<tag>_kernel_stats.txt
TAG=${1:-ks}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs $BENCH_FLAGS"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_k -o r -- $B --steps 5 --warmup 2 > $O/${TAG}_k.log 2>&1
python tools/rocpd_stats.py $O/${TAG}_k/r_results.db > $O/${TAG}_kernel_stats.txt
head -24 $O/${TAG}_kernel_stats.txt | cut -c1-60,100-170
rm -rf $O/${TAG}_k
