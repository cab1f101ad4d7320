out=$PWD/gpurun_out/w14o; mkdir -p $out
for cfg in "swin_tiny_w14 128" "swin_base_w14 32" "swin_base_w14 64"; do
  set -- $cfg
  python bench.py --arch $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_$1_b$2.json
  python -c "import json; d=json.load(open('$out/bench_$1_b$2.json')); print('$1 B=$2', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', d.get('step_mfma_frac'))"
done
repo=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $out/stats -o stats -- python $repo/bench.py --arch swin_tiny_w14 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/stats.log 2>&1
cd $repo
find $out -name "*kernel_trace.csv" -delete
python tools/kernel_families.py $(find $out/stats -name "*kernel_stats.csv" | head -1) 4 | head -14
