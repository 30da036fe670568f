#!/bin/bash
# round 6, call 28: the question encoder on a stream of its own beside the stem (MACNet.overlap_encoder): parity, whole-tower rate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_dp.py -m gpu -q -x > $O/c28_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/c28_pytest.log
python - <<'PY' 2>&1 | grep -v "^W\|amdgpu.ids" | tee gpurun_out/c28_model_level.txt
import sys; sys.path.insert(0, '.')
import torch, bench, macx
dev = torch.device('cuda:0')
for ov in (True, False, True, False):
    orig = macx.MACNet.__init__
    def init(self, *a, _o=orig, _ov=ov, **k):
        k.setdefault('overlap_encoder', _ov); _o(self, *a, **k)
    macx.MACNet.__init__ = init
    r = bench.model_level(macx, dev, 1234, steps=8)
    macx.MACNet.__init__ = orig
    print('overlap_encoder', ov, r['value'], 'q/s', r['ms_per_step'], 'ms', 'loss', r['final_loss'])
PY
