#!/bin/bash
# SQ counter pass over single GEMM shapes, per main-loop variant.  Output: gpurun_out/pmc_gemm/<tag>_counter_collection.csv
out=$PWD/gpurun_out/pmc_gemm; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
for pipe in 1 7 6; do
for spec in "gelu 25088 1536 384" "fwd 25088 1152 384" "res 25088 384 1536"; do
  tag=$(echo $spec | tr ' ' '_')_p$pipe
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -f csv -d $out -o $tag -- python $repo/tools/bench_one_gemm.py $spec $pipe 5 > $out/$tag.log 2>&1
  tail -1 $out/$tag.log
done; done
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
