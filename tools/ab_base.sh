#!/bin/bash
# Same-box A/B of the working tree against a baseline checkout (git worktree add tools/probe/_base <rev>; build it there with
# python -c "import __graft_entry__ as g; g.build()"): alternated default bench runs, ms per step.  Usage: bash tools/ab_base.sh [rounds] [bench args]
root=$(cd "$(dirname "$0")/.." && pwd)
rounds=${1:-3}; shift || true
ms() { (cd $1 && shift && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f (loss %.5f)' % (d['ms_per_step'], d['final_loss']))"); }
for r in $(seq 1 $rounds); do
  echo "round $r: base $(ms $root/tools/probe/_base "$@")   tree $(ms $root "$@")"
done
