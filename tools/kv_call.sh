#!/bin/bash
# one GPU call: sequential (key 9 = 0) vs pipelined forward (1): bitwise parity, ms/step, then the tests that compare the
# one-call path with the stepwise path and the oracle
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python tools/kv_sweep.py 0 1 --key 9 --steps 20 --rounds 3 > $O/pipe_sweep.log 2>&1; echo "sweep rc=$?"
grep -v "^ref" $O/pipe_sweep.log | tail -10
timeout 900 python -m pytest tests/test_gpu_cell.py tests/test_gpu_configs.py tests/test_gpu_graph.py tests/test_gpu_dp.py -x -q > $O/pipe_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pipe_pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs > $O/pipe_bench.json 2> $O/pipe_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/pipe_bench.json')); print(d['value'], d['ms_per_step'])"
