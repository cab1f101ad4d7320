#!/bin/bash
# L2 hit rate of the GEMM kernels of one bench_gemm.py shape filter: bash tools/gemm_l2_hit.sh "s2 fc1"   (on the GPU box)
out=$PWD/gpurun_out/l2hit; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -f csv -d $out -o a -- python $repo/tools/bench_gemm.py --only "$1" > $out/a.log 2>&1
cd $repo
python - <<PY
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:90] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:14]:
    h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print("%-120s hit %.3f  req/launch %.3g  ea_rd/launch %.3g" % (k, h / max(h + m, 1), d.get("TCC_REQ_sum", 0) / max(n[k] / 4, 1), d.get("TCC_EA0_RDREQ_sum", 0) / max(n[k] / 4, 1)))
PY
