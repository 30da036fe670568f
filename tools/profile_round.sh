cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-model-level --no-native"
rocprofv3 --kernel-trace --stats -d gpurun_out/r1k -o r -- $B --steps 5 --warmup 2 > gpurun_out/r1k.log 2>&1
python tools/rocpd_stats.py gpurun_out/r1k/r_results.db > gpurun_out/r01_kernel_stats_split.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r1f -o f -- $B --steps 1 --warmup 0 > gpurun_out/r1f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r1w -o w -- $B --steps 1 --warmup 0 > gpurun_out/r1w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r1s -o s -- $B --steps 1 --warmup 0 > gpurun_out/r1s.log 2>&1
ls gpurun_out/r1f gpurun_out/r1w gpurun_out/r1s
python tools/pmc_summary.py gpurun_out/r1f/f_counter_collection.csv gpurun_out/r1w/w_counter_collection.csv > gpurun_out/r01_pmc_hbm_split.txt 2>&1
python tools/pmc_summary.py gpurun_out/r1s/s_counter_collection.csv > gpurun_out/r01_pmc_sq_split.txt 2>&1
head -8 gpurun_out/r01_pmc_hbm_split.txt | cut -c1-130; head -8 gpurun_out/r01_pmc_sq_split.txt | cut -c1-200
tail -2 gpurun_out/r1s.log | cut -c1-200
