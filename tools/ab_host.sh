#!/bin/bash
# host-to-host leg under environment settings, ONE call: bash tools/ab_host.sh <rounds> "VAR=.." ...
rounds=$1; shift
for r in $(seq $rounds); do
  for e in "$@"; do
    v=$(env $e python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); h=d['host_to_host']; print(d['value'], h['value'], h['ms_per_step'], h['h2d_GBs'])")
    echo "[${e:-defaults}] -> $v"
  done
done
