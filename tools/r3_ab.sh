#!/bin/bash
# same-box A/B of a library knob on the bench step: AB_ENV="MACX_SB_DEFER=0" (the B side's environment)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
FL="--steps 20 --warmup 3 --no-cpu-baseline --no-model-level --no-native --no-extra-legs"
for rep in 1 2; do
  timeout 300 python bench.py $FL > gpurun_out/ab_a$rep.json 2> gpurun_out/ab_a.err; echo "A rc=$?"
  env $AB_ENV timeout 300 python bench.py $FL > gpurun_out/ab_b$rep.json 2> gpurun_out/ab_b.err; echo "B rc=$?"
done
python - <<'PY'
import json
for n in ("ab_a1", "ab_b1", "ab_a2", "ab_b2"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
