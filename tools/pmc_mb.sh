# PMC passes over tools/mb_probe (one shape per run): instruction counts and wave-cycle breakdown of the fused MBConv kernels
cd /tmp && export TMPDIR=/tmp
for i in ${SHAPES:-0 2}; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mbpmc$i -o run -- $GRAFT_REPO_ROOT/tools/mb_probe 32 $i > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mbpmcb$i -o run -- $GRAFT_REPO_ROOT/tools/mb_probe 32 $i > /dev/null 2>&1
done
