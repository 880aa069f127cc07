#!/bin/bash
# like sweep_env.sh, with ONE context (single-stream latency regime) next to the default four
for cfg in "$@"; do
  envs=""
  if [ "$cfg" != "-" ]; then envs=$(echo "$cfg" | tr ',' ' '); fi
  r4=$(env $envs python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-host-leg --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  r1=$(env $envs python bench.py --streams 1 --steps 100 --warmup 20 --no-cpu-baseline --no-host-leg --no-extras --no-saturated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])")
  echo "$cfg -> four: $r4   one: $r1"
done
