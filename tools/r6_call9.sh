#!/bin/bash
# round 6, call 9: chain_fwd stage 0 with the hash under the load latency + biases ahead of the K loops: parity tests, then same-box A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_cell.py tests/test_gpu_reference_golden.py tests/test_gpu_h2.py tests/test_gpu_limits.py tests/test_gpu_graph.py -m gpu -q -x > gpurun_out/c9_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c9_pytest.log
bash tools/r6_ab.sh c9
