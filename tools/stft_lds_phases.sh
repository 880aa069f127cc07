#!/bin/bash
# LDS bank-conflict cycles of stft_kernel by phase: one rocprofv3 --pmc pass per BN_STFT_DBG mask (ON THE GPU BOX, repo root)
#   bash tools/stft_lds_phases.sh <outdir>      masks: 0 all, 4 no mel, 6 no mel / no bin phase, 5 no transform / no mel, 3 mel only
set -o pipefail
R=$(pwd)
O=$R/${1:-gpurun_out/stft_lds}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 4 6 5 3; do
    BN_STFT_DBG=$m rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/m$m -o run -- python3 $R/tools/pmc_run.py 32 3 > $O/m$m.log 2>&1 || exit 1
    echo "mask $m: $(python3 $R/tools/pmc_kernels.py $O/m$m stft)" | tee -a $O/summary.txt
done
