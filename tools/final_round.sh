#!/bin/bash
# round-end artifacts on one MI355X: full GPU test suite (+ observed parity deltas), default bench + rocprofv3 stats + PMC passes
# (tools/gpu_profile.sh), the other BASELINE configurations / batch sizes, SQ counters of the step's kernels, the RCCL code path on a
# single rank, the crop producer.  Usage: bash tools/final_round.sh [tag]
tag=${1:-final}
out=$PWD/gpurun_out/$tag; mkdir -p $out
rm -f gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests -m gpu -q -rP 2>&1 | grep -v "Warning\|warn" | grep "PARITY\|REDUCER\|passed\|failed\|Error" | cut -c1-900 > $out/gpu_tests.txt; tail -2 $out/gpu_tests.txt
cp gpurun_out/parity_observed.jsonl $out/parity_observed.jsonl 2>/dev/null
bash tools/gpu_profile.sh $tag 128 > $out/profile.log 2>&1; tail -c 600 gpurun_out/prof_$tag/bench.json; echo
python tools/pmc_traffic.py gpurun_out/prof_$tag/pmc/fetch_counter_collection.csv gpurun_out/prof_$tag/pmc/write_counter_collection.csv swin_tiny_w7 128 3
for cfg in "swin_tiny_w7 64" "swin_tiny_w7 32" "swin_tiny_w14 128" "swin_base_w14 32" "swin_base_w14 64" "cvt_s1 64" "cvt_s1 128" "deit_tiny 128" "deit_small 128" "vit_base 64" "vil_tiny 64" "vil_tiny 128" "vil_small 64"; do
  set -- $cfg
  python bench.py --arch $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_$1_b$2.json
  python -c "import json; d=json.load(open('$out/bench_$1_b$2.json')); print('$1 B=$2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'step_mfma_frac', d.get('step_mfma_frac'))"
done
ESVIT_FORCE_REDUCER=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $out/bench_rccl1.json
python -c "import json; d=json.load(open('$out/bench_rccl1.json')); print('rccl nproc=1', round(d['value'],1), 'img/s')"
ESVIT_FORCE_REDUCER=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --grad-payload bf16 2>&1 | tail -1 > $out/bench_rccl1_bf16_payload.json
python -c "import json; d=json.load(open('$out/bench_rccl1_bf16_payload.json')); print('rccl nproc=1, bf16 payload', round(d['value'],1), 'img/s')"
bash tools/pmc_kernel_sq.sh step_$tag -- python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > /dev/null 2>&1; head -14 gpurun_out/pmc_step_$tag/summary.txt | cut -c1-200
python bench.py --no-cpu-baseline --no-roofline --augment --steps 10 --warmup 3 2>/dev/null | tail -1 > $out/bench_with_crop_producer.json
# round 6: counter-derived MFMA-busy of the step (BASELINE.json's "MFMA util %"), kernel-family table of the single-stream stats run
bash tools/pmc_mfma_busy.sh 128 swin_tiny_w7 $out/step_mfma_busy.json > $out/mfma_busy.log 2>&1; tail -14 $out/mfma_busy.log
python tools/kernel_families.py $(ls gpurun_out/prof_$tag/stats/*kernel_stats.csv 2>/dev/null | head -1) 4 > $out/step_kernel_families.txt 2>/dev/null; head -24 $out/step_kernel_families.txt
