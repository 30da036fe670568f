#!/bin/bash
# A/B of the fused read-chain kernel against the four launches it replaces: parity tests, then the bench step both ways
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cell.py tests/test_gpu_reference_golden.py tests/test_gpu_unit_exports.py -x -q > gpurun_out/ab_pytest.log 2>&1
echo "pytest rc=$?" ; tail -15 gpurun_out/ab_pytest.log
FL="--steps 10 --warmup 2 --no-cpu-baseline --no-model-level --no-native --no-extra-legs"
timeout 300 python bench.py $FL > gpurun_out/ab_chain.json 2> gpurun_out/ab_chain.err; echo "chain rc=$?"
MACX_CHAIN=0 timeout 300 python bench.py $FL > gpurun_out/ab_nochain.json 2> gpurun_out/ab_nochain.err; echo "nochain rc=$?"
python - <<'PY'
import json
for n in ("ab_chain", "ab_nochain"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
