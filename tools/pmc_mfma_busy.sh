#!/bin/bash
# Counter-derived MFMA-busy of one training step (BASELINE.json's metric asks for "MFMA util %"):
#     mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE)
# over every kernel of the step (one PMC pass, --kernel-trace only; kernels are serialised by the collection, so both sums are per-kernel
# cycles and the clock cancels).  Per-family rows as well.  Usage: bash tools/pmc_mfma_busy.sh [batch] [arch] [out.json]
batch=${1:-128}; arch=${2:-swin_tiny_w7}; outjson=${3:-profiles/r06_step_mfma_busy.json}
out=$PWD/gpurun_out/pmc_mfma_busy; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -f csv -d $out -o m -- python $repo/bench.py --arch $arch --batch $batch --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --single-stream > $out/m.log 2>&1
cd $repo
python - <<PY
import collections, csv, glob, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
tot = collections.defaultdict(float)
rows = 0
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        m = re.search(r"([A-Za-z_0-9]+_kernel)", k)
        fam = m.group(1) if m else k[:40]
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        rows += 1
steps = 3.0  # warm-up + 2 timed steps ran under the collection (setup-time launches are a few per cent of the sums)
# GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked below through the clock it implies against the kernel-trace durations)
dur_ns = 0.0
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur_ns += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
XCDS = 8.0
gui = tot.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
implied_ghz = gui / dur_ns if dur_ns else None
for d in agg.values():
    if "GRBM_GUI_ACTIVE" in d:
        d["GRBM_GUI_ACTIVE"] /= XCDS
res = {"arch": "$arch", "batch": $batch, "steps_counted": steps, "counter_rows": rows, "gui_active_instances": XCDS, "implied_clock_ghz": implied_ghz,
       "note": "mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8 XCD instances); every kernel of the step, PMC-serialised, single stream",
       "SQ_VALU_MFMA_BUSY_CYCLES_per_step": tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / steps, "GRBM_GUI_ACTIVE_per_step": gui / steps,
       "SQ_INSTS_MFMA_per_step": tot.get("SQ_INSTS_MFMA", 0) / steps, "simds": 1024,
       "mfma_busy": (tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * gui)) if gui else None,
       "kernel_ms_per_step": dur_ns / steps / 1e6,
       "mfma_busy_vs_kernel_time_at_2p4GHz": (tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * dur_ns * 2.4)) if dur_ns else None, "families": {}}
for fam, d in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:24]:
    g = d.get("GRBM_GUI_ACTIVE", 0.0)
    res["families"][fam] = {"gui_active_share": g / gui if gui else None, "mfma_busy": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * g) if g else None,
                            "cycles_per_mfma": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d["SQ_INSTS_MFMA"] if d.get("SQ_INSTS_MFMA") else None}
json.dump(res, open("$outjson", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "families"}))
for fam, d in list(res["families"].items())[:12]:
    print("%-34s share %.3f  mfma_busy %s  cycles/mfma %s" % (fam, d["gui_active_share"] or 0, d["mfma_busy"], d["cycles_per_mfma"]))
PY
find $out -name "*kernel_trace.csv" -size +5M -delete; find $out -name "*counter_collection.csv" -size +5M -delete
