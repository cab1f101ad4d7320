#!/bin/bash
# rocprofv3 kernel stats for another architecture / batch: tools/prof_arch.sh ARCH BATCH TAG
arch=$1; batch=$2; tag=$3
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $out/stats -o stats -- python $repo/bench.py --arch $arch --batch $batch --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $out/stats.log 2>&1
find $out -name "*kernel_trace.csv" -delete
