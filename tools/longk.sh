#!/bin/bash
# long-K forward / dgrad shapes under the LDS-DMA pipeline variants (1: 2xBK64, 3: 3xBK64 ring, 4: 4xBK32 ring, 5: 3xBK32, 2: 2xBK32)
for spec in "res 25088 384 1536" "dgrad 25088 384 1536" "res 6272 768 3072" "fwd 6272 2304 768" "fwd 10880 2048 2048" "dgrad 10880 256 65536"; do
for pipe in 1 3 4 5 2; do
  python tools/bench_one_gemm.py $spec $pipe 30 2>&1 | tail -1
done; done
