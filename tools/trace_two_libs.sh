#!/bin/bash
# kernel traces of the default bench (single stream, 2 timed steps) with two library builds -> per-launch comparison of the GEMM launches
# whose kernel differs (tools/compare_traces.py).  Usage: bash tools/trace_two_libs.sh libA.so libB.so [bench args]
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/trace_two; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
la=$(realpath $root/$1 2>/dev/null || echo $1); lb=$(realpath $root/$2 2>/dev/null || echo $2); shift; shift
ESVIT_HIP_LIB=$la rocprofv3 --kernel-trace -f csv -d $out/A -o a -- python $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --single-stream "$@" > $out/a.log 2>&1
ESVIT_HIP_LIB=$lb rocprofv3 --kernel-trace -f csv -d $out/B -o b -- python $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --single-stream "$@" > $out/b.log 2>&1
cd $root; python tools/compare_traces.py $out/A/a_kernel_trace.csv $out/B/b_kernel_trace.csv
