#!/bin/bash
# time the GEMM main-loop variants; for the persistent kernel also with parts removed (ABLATE bits: 1 no MFMA, 2 no loads, 4 no epilogue)
for spec in "gelu 25088 1536 384" "fwd 25088 1152 384" "res 25088 384 1536" "fwd 401408 288 96" "gelu 401408 384 96" "dgrad 25088 384 1536" "wgrad 25088 1152 384"; do
for pipe in 1 6 7; do
for ab in ${ABLATES:-0 4 3}; do
  if [ $pipe = 1 ] && [ $ab != 0 ]; then continue; fi
  ABLATE=$ab python tools/bench_one_gemm.py $spec $pipe 30 2>&1 | tail -1
done; done; done
