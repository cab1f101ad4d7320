#!/bin/bash
# rocprofv3 kernel stats of the default bench (single stream, 3 steps + 1 warm-up) for the working tree and the baseline checkout of
# tools/ab_base.sh, side by side per kernel family.  Usage: bash tools/stats_ab_base.sh
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
for w in base tree; do
  d=$root; [ $w = base ] && d=$root/tools/probe/_base
  out=$root/gpurun_out/stats_ab/$w; mkdir -p $out
  (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $out -o s -- python $d/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/log.txt 2>&1)
  python $root/tools/kernel_families.py $out/s_kernel_stats.csv 4 > $out/families.txt
done
paste <(head -22 $root/gpurun_out/stats_ab/base/families.txt | cut -c1-100) <(head -22 $root/gpurun_out/stats_ab/tree/families.txt | cut -c1-100) | cut -c1-220
tail -1 $root/gpurun_out/stats_ab/base/families.txt; tail -1 $root/gpurun_out/stats_ab/tree/families.txt
find $root/gpurun_out/stats_ab -name "*kernel_trace.csv" -size +20M -delete
