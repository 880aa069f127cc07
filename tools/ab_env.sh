#!/bin/bash
# A/B of environment settings in ONE call (same box): bash tools/ab_env.sh <rounds> "<bench args>" "VAR=1" "VAR=2 OTHER=x" ...   ("" = defaults)
rounds=$1; shift; bargs=$1; shift
for r in $(seq $rounds); do
  for e in "$@"; do
    v=$(env $e python bench.py $bargs --no-cpu-baseline --no-host-leg --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], (d.get('own_buffer') or {}).get('value'), d['device_us_per_step_sum_of_launches'])")
    echo "[${e:-defaults}] -> $v"
  done
done
