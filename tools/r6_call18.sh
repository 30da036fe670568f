#!/bin/bash
# round 6, call 33: timestamps of chain_fwd's filler 0 / first tile (library built with -DMACX_FILL_PROF) for pre_fill 1 and 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1; do python tools/fill_prof.py $v 2>&1 | grep -v "^W\|amdgpu.ids" | tee gpurun_out/c33_prof_$v.txt; done
