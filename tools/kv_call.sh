#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cell.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_reference_golden.py tests/test_gpu_graph.py tests/test_gpu_dp.py -x -q > $O/wb_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/wb_pytest.log
bash tools/kstats.sh wb > /dev/null 2>&1
head -40 $O/wb_kernel_stats.txt | cut -c1-80,100-160
timeout 300 python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs > $O/wb_bench.json 2> $O/wb_bench.err
python -c "
import json; d=json.load(open('$O/wb_bench.json')); print(d['value'], d['ms_per_step'])"
