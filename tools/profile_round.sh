# Round profile of `bench.py` on the GPU box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# writes gpurun_out/<tag>_*: kernel-trace stats, four separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, LDS/VALU) and the
# roofline inputs bench.py reads (profiles/<tag>_roofline_inputs.json after `python tools/profile_post.py <tag>`, run locally).
# Counter passes never share a run with --stats or any trace domain but the kernel trace.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-model-level --no-native --no-extra-legs --no-probe --eager"   # --eager: the step as launches (same kernels as a replay; no settle blocks in the trace)
rocprofv3 --kernel-trace --stats -d $O/${TAG}_k -o r -- $B --steps 5 --warmup 2 > $O/${TAG}_k.log 2>&1
python tools/rocpd_stats.py $O/${TAG}_k/r_results.db > $O/${TAG}_kernel_stats.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_f -o f -- $B --steps 1 --warmup 0 > $O/${TAG}_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_w -o w -- $B --steps 1 --warmup 0 > $O/${TAG}_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_s -o s -- $B --steps 1 --warmup 0 > $O/${TAG}_s.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/${TAG}_l -o l -- $B --steps 1 --warmup 0 > $O/${TAG}_l.log 2>&1
python tools/pmc_summary.py $O/${TAG}_f/f_counter_collection.csv $O/${TAG}_w/w_counter_collection.csv > $O/${TAG}_pmc_hbm.txt 2>&1
python tools/pmc_summary.py $O/${TAG}_s/s_counter_collection.csv > $O/${TAG}_pmc_sq.txt 2>&1
python tools/pmc_summary.py $O/${TAG}_l/l_counter_collection.csv > $O/${TAG}_pmc_lds_valu.txt 2>&1
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
head -12 $O/${TAG}_kernel_stats.txt | cut -c1-170; head -6 $O/${TAG}_pmc_hbm.txt | cut -c1-140; cut -c1-300 $O/${TAG}_bench.json
