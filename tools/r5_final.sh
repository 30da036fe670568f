#!/bin/bash
# round 5, last GPU call: the whole -m gpu suite (log + parity margins kept), smoke(), the round profile (kernel trace, four PMC passes,
# full bench line), the whole-tower kernel trace, the per-GPU shard sizes of the metric's batch, the 2-rank gloo bench on one GPU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/gpu_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
bash tools/profile_round.sh r05 2>&1 | tail -24
bash tools/model_level_kstats.sh r05 2>&1 | tail -50
bash tools/shard_sizes.sh > $O/r05_shard_sizes_raw.txt 2>&1; cat $O/r05_shard_sizes_raw.txt
MACX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-extra-dp > $O/check_bench_gloo2.json 2> $O/check_bench_gloo2.err; echo "gloo2 rc=$?"
tail -c 600 $O/check_bench_gloo2.json; echo
