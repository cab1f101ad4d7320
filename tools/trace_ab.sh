#!/bin/bash
# kernel traces of the default bench (single stream, 2 timed steps) with library build $1 (default B: broad routing; A = product) and the
# round-3 choice (C) -> compare per launch
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/trace_ab; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
v=${1:-B}; shift || true
[ "$v" = "A" ] || export ESVIT_HIP_LIB=$root/tools/probe/libesvit_hip_$v.so
rocprofv3 --kernel-trace -f csv -d $out/A -o a -- python $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --single-stream "$@" > $out/a.log 2>&1
ESVIT_HIP_LIB=$root/tools/probe/libesvit_hip_C.so ESVIT_NO_P8_ROUTING=1 rocprofv3 --kernel-trace -f csv -d $out/C -o c -- python $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --single-stream "$@" > $out/c.log 2>&1
cd $root; python tools/compare_traces.py $out/A/a_kernel_trace.csv $out/C/c_kernel_trace.csv
