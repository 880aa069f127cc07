#!/bin/bash
# Round profile set, run ON THE GPU BOX from the repo root:   bash tools/profile_round.sh <tag>
# Produces under gpurun_out/<tag>/: bench JSON lines (default = 4 contexts, and 1 context), the
# rocprofv3 --kernel-trace --stats CSVs of the same two commands, and the FETCH_SIZE / WRITE_SIZE
# PMC passes (separate passes, no trace domains combined with --pmc) of one batch-32 plan.
set -o pipefail
tag=${1:-prof}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err &&
python3 $R/bench.py --streams 1 --no-cpu-baseline > $O/bench_1stream.json 2> $O/bench_1stream.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o run -- python3 $R/bench.py --no-cpu-baseline > $O/trace_default.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_1stream -o run -- python3 $R/bench.py --streams 1 --no-cpu-baseline > $O/trace_1stream.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/tools/pmc_run.py 32 3 > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/tools/pmc_run.py 32 3 > $O/pmc_write.log 2>&1
