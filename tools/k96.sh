#!/bin/bash
# stage-0 (K = 96) GEMM shapes: register-staged BK32 loop (dma=1 routes K<192 there) vs LDS-DMA BK64 (dma=2 forces it)
for spec in "fwd 401408 288 96" "gelu 401408 384 96" "res 401408 96 96" "res 401408 96 384" "dgrad 401408 96 288" "dgrad 401408 96 384" "fwd 100352 576 192" "gelu 100352 768 192"; do
for dma in 1 2; do
  GEMM_DMA=$dma python tools/bench_one_gemm.py $spec 1 30 2>&1 | tail -1 | sed "s/$/ dma=$dma/"
done; done
