#!/bin/bash
# Run on the MI355X box: deit_small bench line + rocprofv3 kernel stats -> gpurun_out/vit/
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/vit; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 200 python $repo/bench.py --arch deit_small --batch 128 --no-cpu-baseline > $out/bench.json 2> $out/bench.err < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/stats -o stats -- python $repo/bench.py --arch deit_small --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/stats.log 2>&1 < /dev/null
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; head -14 "$f" | cut -c1-160; fi
find $out -name "*kernel_trace.csv" -delete
