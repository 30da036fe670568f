#!/bin/bash
# round 6, call 21: B0 reads sum_n a_n da_n = dinfo . info from kb_att_da_kernel: cell / config / golden parity, A/B against ab_head (= a4968de)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cell.py tests/test_gpu_configs.py tests/test_gpu_reference_golden.py tests/test_gpu_knobs.py tests/test_gpu_unit_exports.py -m gpu -q -x > $O/c21_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/c21_pytest.log
bash tools/r6_ab.sh c21
