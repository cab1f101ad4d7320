#!/bin/bash
# SQ instruction-mix / stall counters for every kernel of one training step (two passes)
batch=${1:-64}; arch=${2:-swin_tiny_w7}
out=$PWD/gpurun_out/pmc_step_sq; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace -f csv -d $out -o a -- python $repo/bench.py --arch $arch --batch $batch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $out -o b -- python $repo/bench.py --arch $arch --batch $batch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/b.log 2>&1
ls -la $out
