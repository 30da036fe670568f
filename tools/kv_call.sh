#!/bin/bash
# one GPU call: the stem on the 3-term fp16 family: parity tests, timing (tools/microbench.py --stem), kernel stats, tower tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stem.py -x -q > $O/stem_pytest.log 2>&1; echo "stem pytest rc=$?"; tail -5 $O/stem_pytest.log
timeout 300 python tools/microbench.py --stem > $O/stem_micro.log 2>&1; tail -6 $O/stem_micro.log
MACX_GEMM=split timeout 300 python tools/microbench.py --stem > $O/stem_micro_split.log 2>&1; tail -3 $O/stem_micro_split.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kv_k -o r -- python tools/microbench.py --stem > $O/stem_k.log 2>&1
python tools/rocpd_stats.py $O/kv_k/r_results.db > $O/stem_kernel_stats.txt; rm -rf $O/kv_k
head -14 $O/stem_kernel_stats.txt | cut -c1-90,100-160
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_dp.py tests/test_gpu_output.py -x -q > $O/stem_pytest2.log 2>&1; echo "tower pytest rc=$?"; tail -3 $O/stem_pytest2.log
