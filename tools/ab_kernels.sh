#!/bin/bash
# per-kernel A/B of one environment switch on the default step: bash tools/ab_kernels.sh VAR "pattern1|pattern2"  (on the GPU box)
# writes gpurun_out/ab_<VAR>/{1,0}_kernel_stats.csv and prints the kernels matching the pattern, per step
var=$1; pat=$2
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  d=$GRAFT_REPO_ROOT/gpurun_out/ab_$var/$v; mkdir -p $d
  env $var=$v rocprofv3 --kernel-trace --stats -d $d -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --single-stream > $d/log.txt 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $var=$v"; python - "$f" "$pat" <<'P'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels: %.2f ms / 6 steps" % (tot / 1e6))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print("%9.1f us total %5d calls  %s" % (float(r["TotalDurationNs"]) / 1e3, int(r["Calls"]), r["Name"][:110]))
P
done
