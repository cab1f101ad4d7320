#!/bin/bash
# rocprofv3 kernel stats of the data-parallel code path on ONE rank (ESVIT_FORCE_REDUCER=1 under torch.distributed.run): what the
# reducer adds to a step -- pack copies, hooks, the one-rank "all-reduce" -- without a second GPU.
out=$PWD/gpurun_out/rccl1; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
ESVIT_FORCE_REDUCER=1 rocprofv3 --kernel-trace --stats -f csv -d $out/stats -o stats -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 $repo/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/stats.log 2>&1
cd $repo; ls $out/stats | head; find $out -name "*kernel_trace.csv" -size +20M -delete
