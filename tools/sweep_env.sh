#!/bin/bash
# bench.py under a list of environment settings: "NAME=VALUE[,NAME=VALUE...]" per argument ("-" = defaults)
for cfg in "$@"; do
  envs=""
  if [ "$cfg" != "-" ]; then envs=$(echo "$cfg" | tr ',' ' '); fi
  r=$(env $envs python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-host-leg --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_ms'])")
  echo "$cfg -> $r"
done
