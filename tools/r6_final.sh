#!/bin/bash
# round 6 profile of record: kernel stats + PMC passes + bench line (profile_round.sh), whole-tower kernel stats, the full GPU suite
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG
bash tools/model_level_kstats.sh $TAG
bash tools/full_gpu.sh
cp gpurun_out/gpu_pytest.log gpurun_out/${TAG}_gpu_pytest.log
