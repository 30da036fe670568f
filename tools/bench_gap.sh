#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/bg
rocprofv3 --kernel-trace -d $O/bg -o r -- python bench.py --no-cpu-baseline > $O/bg_bench.json 2> $O/bg_bench.err
python - <<PY
import json
d=json.loads(open("$O/bg_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("eager_step",{}).get("ms_per_step"), d.get("settle"))
for k in ("fwd_only_p4","train_b128_p12_adam_ema","gqa_shape_p4_args3","model_level","other_families"):
    v=d.get(k); print(k, v if not isinstance(v,dict) else {a:v[a] for a in v if a in ("value","ms_per_step","ms_per_batch","split_bf16_6term","native_f32_mfma")})
PY
python tools/gap_report.py $O/bg/r_results.db > $O/bg_gap_report.txt
rm -rf $O/bg
