#!/bin/bash
# Round profile set, run ON THE GPU BOX from the repo root:   bash tools/profile_round.sh <tag> [v24|v30|perch] [batch]
# (v24 / 32 = BASELINE configs[1], the default; v30 64 = configs[2]; perch 128 = configs[3])
# Produces under gpurun_out/<tag>/: bench JSON lines (default = 4 contexts, and 1 context), the
# rocprofv3 --kernel-trace --stats CSVs of the same two commands, and the FETCH_SIZE / WRITE_SIZE
# PMC passes (separate passes, no trace domains combined with --pmc) of one batch-32 plan.
set -o pipefail
tag=${1:-prof}
model=${2:-v24}
batch=${3:-32}
MB="--model $model --batch $batch"
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py $MB > $O/bench_default.json 2> $O/bench_default.err &&
python3 $R/bench.py $MB --streams 1 --no-cpu-baseline --no-extras > $O/bench_1stream.json 2> $O/bench_1stream.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o run -- python3 $R/bench.py $MB --no-cpu-baseline --no-extras --no-host-leg > $O/trace_default.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_1stream -o run -- python3 $R/bench.py $MB --streams 1 --no-cpu-baseline --no-extras --no-host-leg > $O/trace_1stream.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/tools/pmc_run.py $batch 3 $model > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/tools/pmc_run.py $batch 3 $model > $O/pmc_write.log 2>&1 &&
# matrix-pipe utilisation, LDS conflicts, wait breakdown and the clock (SQ: 8 slots per pass, GRBM: its own 2)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 $R/tools/pmc_run.py $batch 3 $model > $O/pmc_sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_inst -o run -- python3 $R/tools/pmc_run.py $batch 3 $model > $O/pmc_inst.log 2>&1 &&
# the same two SQ passes per kernel NAME (template arguments kept): profiles/<tag>_pmc_by_kernel.txt
(python3 $R/tools/pmc_kernels.py $O/pmc_sq > $O/pmc_by_kernel_sq.txt; python3 $R/tools/pmc_kernels.py $O/pmc_inst > $O/pmc_by_kernel_inst.txt; true) &&
# BASELINE configs[4] on ONE GPU: the 24 h recording (28 800 windows), log kept
# (both gathers, labelled: "logits" = the reference-equivalent output with raw_scores, the figure rounds 1-2 quoted; "topk" = rows only)
if [ "$model" = v24 ]; then
python3 $R/tools/analyze_recording.py --hours 24 --gather logits > $O/recording_24h.log 2>&1 &&
python3 $R/tools/analyze_recording.py --hours 24 --gather topk >> $O/recording_24h.log 2>&1
fi
