#!/bin/bash
# same-box A/B of an environment variable over the default bench: bash tools/probe/ab_env.sh VAR "v1 v2 v3" [rounds] [bench args]
var=$1; vals=$2; rounds=${3:-2}; shift; shift; shift
for r in $(seq 1 $rounds); do
  line="round $r:"
  for v in $vals; do
    ms=$(env $var=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
    line="$line  $var=$v $ms ms"
  done
  echo "$line"
done
