#!/bin/bash
# round 6, call 25: dKB jobs with the epilogue on the accumulators: parity, stamps, sweep dkb_fill 3 / 4 / 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -x -k "dkb_on_idle" > $O/c25_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/c25_pytest.log
python tools/fill_prof.py 1 2>&1 | grep -v "^W\|amdgpu.ids" | tail -3 | tee $O/c25_prof.txt
timeout 600 python tools/kv_sweep.py --key dkb_fill 3 4 0 --steps 30 --rounds 3 > $O/c25_sweep.txt 2>&1; tail -10 $O/c25_sweep.txt
