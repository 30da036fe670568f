#!/bin/bash
# kernel-trace statistics of the whole-tower training step (bench.py's model_level leg): gpurun_out/<tag>_model_level_kernel_stats.txt
TAG=${1:-ml}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats -d $O/${TAG}_mlk -o r -- python -c "
import sys; sys.path.insert(0, '.')
import torch, bench, macx
dev = torch.device('cuda:0')
print(bench.model_level(macx, dev, 1234, steps=7))
" > $O/${TAG}_mlk.log 2>&1
python tools/rocpd_stats.py $O/${TAG}_mlk/r_results.db > $O/${TAG}_model_level_kernel_stats.txt
tail -2 $O/${TAG}_mlk.log
head -45 $O/${TAG}_model_level_kernel_stats.txt | cut -c1-70,100-170
rm -rf $O/${TAG}_mlk
