#!/bin/bash
# v3.0 / Perch throughput of two builds in one call: bash tools/ab_models.sh <other .so>
other=$1
for m in "v30 64" "perch 128"; do set -- $m
  for lib in "" "$other" "" "$other"; do
    if [ -n "$lib" ]; then export BN_LIB=$lib; else unset BN_LIB; fi
    v=$(python bench.py --model $1 --batch $2 --steps 60 --warmup 8 --no-cpu-baseline --no-host-leg --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
    echo "$1 ${lib:-current} -> $v"
  done
done
