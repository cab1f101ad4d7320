#!/bin/bash
# first-round phase stagger of the second workgroup per CU (GEMM_STAGGER = shader cycles; +2^30 selects odd slots)
out=$PWD/gpurun_out/stagger; mkdir -p $out
M30=1073741824
for spec in "gelu 87040 1536 384" "dgrad 87040 1536 384" "fwd 87040 1152 384" "res 87040 384 1536" "gelu 1392640 384 96" "fwd 21760 65536 256"; do
for st in 0 6000 12000 20000 $((M30+6000)) $((M30+12000)) $((M30+20000)) 0; do
  GEMM_STAGGER=$st python tools/bench_one_gemm.py $spec 1 20 2>&1 | tail -1 | sed "s/^/stagger=$st /"
done; done | tee $out/ab.txt
