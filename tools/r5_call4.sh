#!/bin/bash
# round 5, GPU call 4: 8-wave [B,d] linears (key 12) A/B under a kernel trace per value; phase knobs of the deferred contractions
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_cell.py -m gpu -q -x > $O/r5c4_targeted.log 2>&1
echo "targeted rc=$?"; tail -4 $O/r5c4_targeted.log
for v in 0 1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r5c4_kv -o r -- python $GRAFT_REPO_ROOT/tools/kv_sweep.py --key 12 $v --steps 10 --rounds 2 > $GRAFT_REPO_ROOT/$O/r5c4_kv12_$v.txt 2>&1)
  echo "kv 12=$v rc=$?"; grep -E "^(kv|round)" $O/r5c4_kv12_$v.txt
  python tools/rocpd_stats.py $O/r5c4_kv/r_results.db > $O/r5c4_kv12_${v}_kernel_stats.txt 2>&1; rm -rf $O/r5c4_kv
  grep -E "small_linear|kernel  " $O/r5c4_kv12_${v}_kernel_stats.txt | cut -c1-75,100-175
done
timeout 300 python tools/kv_sweep.py --key 12 0 1 --steps 20 --rounds 4 2>&1 | grep -E "^(kv|round)" > $O/r5c4_kv12_ab.txt; cat $O/r5c4_kv12_ab.txt
echo "# phase knobs (macx_debug_set(1, mask)), 6 eager steps per mask, per-kernel average us" > $O/r5c4_phase_knobs.txt
for m in 0 512 1024 2048 3072 32 64 96; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r5c4_m -o r -- python $GRAFT_REPO_ROOT/tools/mask_steps.py $m > /dev/null 2>&1)
  echo "== mask $m" >> $O/r5c4_phase_knobs.txt
  python tools/rocpd_stats.py $O/r5c4_m/r_results.db 2>/dev/null | grep -E "sb_h2w|wgrad_h2_kernel|kb_gemm_h2" | awk '{printf "%-78s %6s %10s %10s\n", substr($1,1,78), $2, $4, $5}' >> $O/r5c4_phase_knobs.txt
  rm -rf $O/r5c4_m
done
cat $O/r5c4_phase_knobs.txt
