#!/bin/bash
# tools/small_batch_profile.sh <per-gpu-batch> (GPU box): where a small-batch step goes -- launches per step, sum of kernel
# durations vs wall time per step (the difference is launch gaps), the longest kernels.
export TMPDIR=/tmp
b=${1:-8}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe --per-gpu-batch $b"
python bench.py $F 2>/dev/null | grep -a "^{" | sed -E "s/.*\"ms_per_step\": ([0-9.]+).*/wall ms per step (no profiler): \1/"
rm -rf /tmp/sb_$b
timeout 300 rocprofv3 --kernel-trace -d /tmp/sb_$b -o k -- python bench.py $F > /dev/null 2>&1
python tools/rocpd_stats.py /tmp/sb_$b/k_results.db > /tmp/sb_$b/stats.txt 2>&1
awk 'NR>1 {c+=$2; t+=$3} END {printf "launches per step %.0f, kernel time per step %.3f ms\n", c/25, t/25/1000}' /tmp/sb_$b/stats.txt
awk 'NR>1 {printf "%-72s %5.1f/step %8.1f us avg %7.1f us/step\n", substr($1,1,72), $2/25, $4, $3/25}' /tmp/sb_$b/stats.txt | sort -t' ' -k6 -n -r | head -${2:-22}
