#!/bin/bash
# Run on the MI355X box: crop-producer bench line + rocprofv3 kernel stats of the same command -> gpurun_out/aug_<tag>/
tag=${1:-a}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/aug_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 120 python $repo/tools/bench_augment.py --iters 50 > $out/bench.json 2> $out/bench.err < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/stats -o stats -- python $repo/tools/bench_augment.py --iters 20 > $out/stats.log 2>&1 < /dev/null
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; head -8 "$f" | cut -c1-200; fi
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $out/pmc -o fetch -- python $repo/tools/bench_augment.py --iters 10 --cpu-images 0 > $out/pmc_fetch.log 2>&1 < /dev/null
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $out/pmc -o write -- python $repo/tools/bench_augment.py --iters 10 --cpu-images 0 > $out/pmc_write.log 2>&1 < /dev/null
find $out -name "*kernel_trace.csv" -delete
cat $out/bench.json
