#!/bin/bash
# A/B of two builds of the library in ONE call (same box): bash tools/ab_lib.sh <other .so> [rounds]
other=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for lib in "" "$other"; do
    if [ -n "$lib" ]; then export BN_LIB=$lib; else unset BN_LIB; fi
    v=$(python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-host-leg --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['device_us_per_step_sum_of_launches'])")
    echo "${lib:-current} -> $v"
  done
done
