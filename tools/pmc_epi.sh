#!/bin/bash
# instruction mix + stall split of the GEMM variants on one shape (epilogue-only and full)
out=$PWD/gpurun_out/pmc_epi; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
for cfg in "1 0" "6 0" "6 3" "6 4"; do
  set -- $cfg; pipe=$1; ab=$2
  tag=p${pipe}_a${ab}
  ABLATE=$ab rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --kernel-trace -f csv -d $out -o $tag -- python $repo/tools/bench_one_gemm.py fwd 25088 1152 384 $pipe 5 > $out/$tag.log 2>&1
  ABLATE=$ab rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM --kernel-trace -f csv -d $out -o ${tag}_b -- python $repo/tools/bench_one_gemm.py fwd 25088 1152 384 $pipe 5 > $out/${tag}_b.log 2>&1
  tail -2 $out/${tag}_b.log | head -1
done
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
