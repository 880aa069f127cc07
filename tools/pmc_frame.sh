# PMC passes over the framing GEMM of the v2.4 plan (no graph capture: one dispatch per op): LDS conflicts, instruction mix, wave-cycle
# breakdown.   bash tools/pmc_frame.sh <out dir under gpurun_out>   (rocprofv3 gets the python program directly, no trace domains)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/lds -o run -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py 32 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/inst -o run -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py 32 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_kernels.py $out/lds frame_fold > $out/frame_lds.txt
python3 tools/pmc_kernels.py $out/inst frame_fold > $out/frame_inst.txt
cat $out/frame_lds.txt $out/frame_inst.txt
