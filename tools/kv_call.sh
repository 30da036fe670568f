#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
: > $O/graph_verify.log
for i in 1 2 3 4 5 6 7 8; do timeout 120 python tools/graph_train_probe_verify.py 2>&1 | tail -1 | cut -c1-200 >> $O/graph_verify.log; done
sort $O/graph_verify.log | uniq -c
for i in 1 2 3; do timeout 120 python -m pytest tests/test_gpu_graph.py -q -k metric_shape 2>&1 | tail -1; done
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_cell.py tests/test_gpu_units.py tests/test_gpu_unit_exports.py tests/test_gpu_unit_parity.py tests/test_gpu_encoder.py tests/test_gpu_output.py tests/test_gpu_generic.py -x -q > $O/memset_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/memset_pytest.log
