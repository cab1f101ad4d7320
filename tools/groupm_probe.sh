#!/bin/bash
# grouped tile order A/B (GEMM_GROUP_M: 0 = column-fastest, 8 = groups of 8 row blocks) on shapes with many column tiles
out=$PWD/gpurun_out/groupm; mkdir -p $out
for spec in "fwd 8192 8192 4096" "fwd 21760 65536 256" "fwd 12544 65536 256" "gelu 21760 3072 768" "fwd 21760 2304 768" "gelu 21760 2048 2048" "gelu 21760 2048 768" "gelu 87040 1536 384" "fwd 87040 1152 384"; do
for g in 0 4 8 16; do
  GEMM_GROUP_M=$g python tools/bench_one_gemm.py $spec 1 20 2>&1 | tail -1 | sed "s/^/group_m=$g /"
done; done | tee $out/ab.txt
