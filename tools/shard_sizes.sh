#!/bin/bash
# per-GPU shard sizes of the metric's batch on one GPU: ms per step at b = 8, 16, 32, 64 (one box, one call), eager and from one graph
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
for b in 8 16 32 64 64; do
  F="--steps 20 --warmup 5 --no-cpu-baseline --no-model-level --no-native --no-extra-legs --per-gpu-batch $b"
  python bench.py $F 2>/dev/null | grep -a "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b=$b  eager %.3f ms per step' % d['ms_per_step'])"
done
bash tools/small_batch_profile.sh 8 14
