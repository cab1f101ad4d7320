#!/bin/bash
# round-end artifacts on one MI355X: full GPU test suite, default bench + rocprofv3 stats + PMC passes (tools/gpu_profile.sh),
# the other BASELINE configurations / batch sizes, and the RCCL code path on a single rank.
out=$PWD/gpurun_out/final; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/gpu_tests.txt; cat $out/gpu_tests.txt
bash tools/gpu_profile.sh final 128 > $out/profile.log 2>&1; tail -c 600 gpurun_out/prof_final/bench.json
for cfg in "swin_tiny_w7 64" "swin_tiny_w7 32" "swin_tiny_w14 64" "swin_base_w14 32" "cvt_s1 64" "deit_tiny 128" "deit_small 128" "vit_base 64"; do
  set -- $cfg
  python bench.py --arch $1 --batch $2 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_$1_b$2.json
  python -c "import json; d=json.load(open('$out/bench_$1_b$2.json')); print('$1 B=$2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
done
ESVIT_FORCE_REDUCER=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 > $out/bench_rccl1.json
python -c "import json; d=json.load(open('$out/bench_rccl1.json')); print('rccl nproc=1', round(d['value'],1), 'img/s')"
# the crop producer: bench line (+ Pillow baseline), kernel stats, PMC passes; and the step with the producer inside it
bash tools/aug_profile.sh final > $out/aug_profile.log 2>&1; tail -c 400 gpurun_out/aug_final/bench.json
python bench.py --no-cpu-baseline --no-roofline --augment 2>/dev/null | tail -1 > $out/bench_with_crop_producer.json
