"""Fold two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; both with --kernel-trace
-f csv) of `python bench.py --steps S --warmup W --batch B` into profiles/r05_pmc_traffic.json (ESVIT_PMC_TRAFFIC_OUT overrides the name).

  python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv ARCH BATCH STEPS_TOTAL

STEPS_TOTAL = warm-up + profiled + timed steps the command ran (every step launches the same kernels).
The counters are in KB.  FETCH_SIZE is doubled: on gfx950 it tallies the 128-byte requests of wide coalesced reads
at 64 bytes (MI355X_MICROARCH.md, HBM section).  Both counters sit on the L2's fabric side, so Infinity-Cache
hits are included: this is traffic below the L2, an upper bound on HBM traffic."""
import collections, csv, json, os, re, sys

GEMM_KERNELS = ("gemm_dma_kernel", "gemm_kernel", "gemm_p8_kernel", "gemm_p8n_kernel", "splitk_reduce_kernel")


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"([A-Za-z_0-9]+_kernel)", name)
    return m.group(1) if m else name[:40]


def fold(path):
    agg = collections.defaultdict(float)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            agg[family(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1e3  # KB -> bytes
    return agg


def main():
    fetch, write = fold(sys.argv[1]), fold(sys.argv[2])
    arch, batch, steps = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    fams = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))
    per_kernel = {k: {"fetch_bytes_per_step": 2 * fetch.get(k, 0) / steps, "write_bytes_per_step": write.get(k, 0) / steps} for k in fams}
    gemm = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for k, v in per_kernel.items() if k in GEMM_KERNELS)
    total = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in per_kernel.values())
    run = {"arch": arch, "batch": batch, "steps_counted": steps, "gemm_bytes_per_step": gemm, "all_kernels_bytes_per_step": total,
           "fetch_correction": 2.0, "per_kernel": per_kernel}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", os.environ.get("ESVIT_PMC_TRAFFIC_OUT", "r05_pmc_traffic.json"))
    doc = {"runs": []}
    if os.path.exists(out):
        with open(out) as fh:
            doc = json.load(fh)
    doc["runs"] = [r for r in doc["runs"] if not (r["arch"] == arch and r["batch"] == batch)] + [run]
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("gemm %.1f GB/step, all kernels %.1f GB/step" % (gemm / 1e9, total / 1e9))


if __name__ == "__main__":
    main()
