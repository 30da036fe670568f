cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-model-level --no-native --steps 1 --warmup 0"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r1l -o l -- $B > gpurun_out/r1l.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --output-format csv -d gpurun_out/r1v -o v -- $B > gpurun_out/r1v.log 2>&1
python tools/pmc_summary.py gpurun_out/r1l/l_counter_collection.csv > gpurun_out/r01_pmc_lds_split.txt 2>&1
python tools/pmc_summary.py gpurun_out/r1v/v_counter_collection.csv > gpurun_out/r01_pmc_valu_split.txt 2>&1
head -12 gpurun_out/r01_pmc_lds_split.txt | cut -c1-210; head -12 gpurun_out/r01_pmc_valu_split.txt | cut -c1-210
tail -2 gpurun_out/r1v.log | cut -c1-160
