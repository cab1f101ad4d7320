out=$PWD/gpurun_out/final2; mkdir -p $out
timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention or window" 2>&1 | tail -2
for cfg in "swin_tiny_w7 64" "swin_tiny_w7 32" "swin_tiny_w14 64" "swin_base_w14 32" "cvt_s1 64"; do
  set -- $cfg
  timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > $out/bench_$1_b$2.json
  python -c "import json; d=json.load(open('$out/bench_$1_b$2.json')); print('$1 B=$2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
done
ESVIT_FORCE_REDUCER=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 < /dev/null | tail -1 > $out/bench_rccl1.json
python -c "import json; d=json.load(open('$out/bench_rccl1.json')); print('rccl nproc=1', round(d['value'],1), 'img/s')"
