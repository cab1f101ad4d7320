#!/bin/bash
# epilogue-only (ABLATE=3: no loads, no MFMA) store time of the persistent kernel for a fixed number of output
# elements (28.9 M) at different row widths N -- does the tile-strided write pattern cost bandwidth?
for spec in "fwd 225792 128 64" "fwd 112896 256 64" "fwd 75264 384 64" "fwd 50176 576 64" "fwd 37632 768 64" "fwd 25088 1152 64" "fwd 12544 2304 64" "fwd 6272 4608 64"; do
  ABLATE=3 python tools/bench_one_gemm.py $spec 6 30 2>&1 | tail -1
done
