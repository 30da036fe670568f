#!/bin/bash
# tools/knob_profile.sh "<mask> <mask> ..." [kernel-name regex]  (GPU box): per-kernel durations of the bench step under the
# measurement knobs of macx_opts.tune[MACX_TUNE_PHASE_MASK] (MACX_DBG) -- results are WRONG under a non-zero mask, only the timing means something.
export TMPDIR=/tmp
F="--steps 6 --warmup 2 --no-cpu-baseline --no-model-level --no-native --no-extra-legs"
for m in $1; do
  rm -rf /tmp/knob_$m
  MACX_DBG=$m timeout 300 rocprofv3 --kernel-trace -d /tmp/knob_$m -o k -- python bench.py $F > /dev/null 2>&1
  echo "== mask $m"
  python tools/rocpd_stats.py /tmp/knob_$m/k_results.db 2>&1 | grep -E "${2:-sb_h2|wgrad_h2_kernel|kb_gemm_h2}" | awk '{printf "%-80s %8s %10s\n", substr($1,1,80), $2, $4}'
done
