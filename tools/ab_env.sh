#!/bin/bash
# same-box A/B of environment switches on the default bench: interleaved rounds, ms per step of each arm.
#   bash tools/ab_env.sh ROUNDS "NAME=VAL" "NAME=VAL" ... [-- bench args]      (an arm "-" is the default environment)
rounds=$1; shift
arms=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do arms+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq 1 $rounds); do
  line="round $r:"
  for a in "${arms[@]}"; do
    if [ "$a" = "-" ]; then v=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    else v=$(env $a python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); fi
    line="$line  [$a] $v"
  done
  echo "$line"
done
