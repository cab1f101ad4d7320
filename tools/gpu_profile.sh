#!/bin/bash
# Run on the MI355X box: default bench line, rocprofv3 kernel stats (single-stream schedule, so that no kernel's duration
# contains a concurrent kernel of the teacher stream), and the two PMC passes (FETCH_SIZE / WRITE_SIZE
# in separate runs, --kernel-trace only).  Outputs under gpurun_out/prof_$1/.
tag=${1:-b128}; batch=${2:-128}
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp
python bench.py --batch $batch --gemm-table $out/gemm_table.txt > $out/bench.json 2> $out/bench.err
tail -c 2500 $out/bench.json
repo=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $out/stats -o stats -- python $repo/bench.py --batch $batch --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $out/pmc -o fetch -- python $repo/bench.py --batch $batch --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $out/pmc -o write -- python $repo/bench.py --batch $batch --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/pmc_write.log 2>&1
cd $repo
ls -la $out $out/stats $out/pmc | head -40
# keep only the summaries small enough to travel
find $out -name "*kernel_trace.csv" -size +20M -delete
