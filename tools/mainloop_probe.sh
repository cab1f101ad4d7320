#!/bin/bash
# Asymptotic main-loop rate of the LDS-DMA GEMM: a square long-K shape whose tile count is a whole number of rounds for
# every variant (4096 tiles), per pipeline variant, then LDS / wait counters for the default one.
out=$PWD/gpurun_out/mainloop; mkdir -p $out; repo=$PWD
for pipe in 1 3 4 5 2; do
  python tools/bench_one_gemm.py fwd 8192 8192 4096 $pipe 10 2>&1 | tail -1
done | tee $out/pipes.txt
GEMM_M256=1 python tools/bench_one_gemm.py fwd 8192 8192 4096 1 10 2>&1 | tail -1 | sed 's/^/m256 /' | tee -a $out/pipes.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3-avail list 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_WAIT[A-Z_]*\|TCP_[A-Z_]*STALL[A-Z_]*\|TCP_PENDING[A-Z_]*\|TA_BUSY[A-Z_]*" | sort -u | tr '\n' ' ' > $out/avail.txt
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -f csv -d $out -o $tag -- python $repo/tools/bench_one_gemm.py fwd 8192 8192 4096 1 3 > $out/$tag.log 2>&1
done
cd $repo
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/mainloop/*counter_collection.csv')):
    d = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'gemm_dma' in r['Kernel_Name']:
            d[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print(f.split('/')[-1], {k: '%.3g' % (v / max(n[k], 1)) for k, v in d.items()})
PY
rm -f $out/*kernel_trace.csv $out/*agent_info.csv
