#!/bin/bash
# L2-prefetch A/B (GEMM_PF 0/1) of the LDS-DMA GEMM on the step's heaviest shapes, then the whole step
out=$PWD/gpurun_out/pf; mkdir -p $out
for spec in "fwd 8192 8192 4096" "gelu 87040 1536 384" "dgrad 87040 1536 384" "fwd 87040 1152 384" "res 87040 384 1536" "gelu 1392640 384 96" "res 1392640 96 384" "fwd 21760 65536 256" "dgrad 21760 256 65536" "wgrad 87040 1536 384" "wgrad 21760 65536 256"; do
for pf in 0 1 0 1; do
  GEMM_PF=$pf python tools/bench_one_gemm.py $spec 1 20 2>&1 | tail -1 | sed "s/^/pf=$pf /"
done; done | tee $out/ab.txt
for pf in 0 1 0 1; do ESVIT_GEMM_PF=$pf python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step pf=$pf', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['gemm_ms_per_step'],2))"; done | tee -a $out/ab.txt
