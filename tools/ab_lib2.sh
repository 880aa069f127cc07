#!/bin/bash
# A/B of two builds of the library over several models in ONE call: bash tools/ab_lib2.sh <other .so> <rounds>
other=$(realpath $1); rounds=${2:-2}
for spec in "v24 32 200 20" "v30 64 60 8" "perch 128 40 8"; do
  set -- $spec
  for r in $(seq $rounds); do
    for lib in "" "$other"; do
      if [ -n "$lib" ]; then export BN_LIB=$lib; else unset BN_LIB; fi
      v=$(python bench.py --model $1 --batch $2 --steps $3 --warmup $4 --no-cpu-baseline --no-host-leg --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['device_us_per_step_sum_of_launches'], d['roofline']['frac'])")
      echo "$1 ${lib:-current} -> $v"
    done
  done
done
