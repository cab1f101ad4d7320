#!/bin/bash
# SQ instruction-mix / stall counters of the kernels one command launches (three PMC passes, --kernel-trace only):
#   tools/pmc_kernel_sq.sh TAG -- python tools/bench_mlp.py
tag=$1; shift; shift
out=$PWD/gpurun_out/pmc_$tag; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace -f csv -d $out -o a -- "$@" > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -f csv -d $out -o b -- "$@" > $out/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -f csv -d $out -o c -- "$@" > $out/c.log 2>&1
cd $repo
python - <<PY
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("$out/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$out/summary.txt", "w") as fh:
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
        wc = d.get("SQ_WAVE_CYCLES", 1.0)
        line = "%-70s " % k + " ".join("%s=%.3g" % (n, v) for n, v in sorted(d.items()))
        frac = " | wait_any %.2f wait_inst %.2f active %.2f valu/mfma %.1f lds_conflict/idx %.2f" % (
            d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            d.get("SQ_INSTS_VALU", 0) / max(d.get("SQ_INSTS_MFMA", 1), 1), d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1))
        fh.write(line + frac + "\n")
print(open("$out/summary.txt").read())
PY
find $out -name "*kernel_trace.csv" -size +5M -delete
