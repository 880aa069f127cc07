#!/bin/bash
# A/B of bench.py argument sets in ONE call (same box): bash tools/ab_args.sh <rounds> "<common args>" "<args A>" "<args B>" ...   (env assignments may lead an arg set)
rounds=$1; shift; common=$1; shift
for r in $(seq $rounds); do
  for a in "$@"; do
    envs=""; rest=""
    for w in $a; do case "$w" in [A-Z_]*=*) envs="$envs $w";; *) rest="$rest $w";; esac; done
    v=$(env $envs python bench.py $common $rest --no-cpu-baseline --no-host-leg --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['median_ms'])")
    echo "[$a] -> $v"
  done
done
