for flags in "" "--no-saturated" "--no-extras" "--no-host-leg" "--no-cpu-baseline" "--no-saturated --no-extras --no-host-leg --no-cpu-baseline"; do
  for r in 1 2; do
    v=$(python bench.py --steps 20 --warmup 5 $flags 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])")
    echo "[$flags] -> $v"
  done
done
