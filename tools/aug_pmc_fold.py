"""Fold the two PMC passes of tools/aug_profile.sh (FETCH_SIZE / WRITE_SIZE of tools/bench_augment.py) into
profiles/r02_crop_producer_pmc.json: bytes below the L2 per batch and per kernel, FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950 (both counters include Infinity-Cache hits: an upper bound on HBM traffic).

    python tools/aug_pmc_fold.py gpurun_out/aug_<tag> BATCHES_RENDERED
"""
import collections, csv, json, os, re, sys


def fold(path):
    agg = collections.defaultdict(float)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(aug_[a-z_]+kernel(?:<[^>]*>)?)", r["Kernel_Name"])
            if m:
                agg[m.group(1)] += float(r["Counter_Value"]) * 1e3  # KB -> bytes
    return agg


def main():
    d, batches = sys.argv[1], int(sys.argv[2])
    fetch, write = fold(os.path.join(d, "pmc", "fetch_counter_collection.csv")), fold(os.path.join(d, "pmc", "write_counter_collection.csv"))
    per = {k: {"fetch_bytes_per_batch": 2 * fetch.get(k, 0) / batches, "write_bytes_per_batch": write.get(k, 0) / batches} for k in sorted(set(fetch) | set(write))}
    total = sum(v["fetch_bytes_per_batch"] + v["write_bytes_per_batch"] for v in per.values())
    bench = json.load(open(os.path.join(d, "bench.json")))
    alg = bench["roofline"]["box_bytes"] + bench["roofline"]["out_bytes"]
    out = {"batches_counted": batches, "fetch_correction": 2.0, "bytes_below_l2_per_batch": total, "algorithmic_bytes_per_batch": alg,
           "ratio": total / alg, "per_kernel": per}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_crop_producer_pmc.json")
    json.dump(out, open(path, "w"), indent=1)
    print("below the L2: %.0f MB per batch, algorithmic %.0f MB (x%.2f)" % (total / 1e6, alg / 1e6, total / alg))


if __name__ == "__main__":
    main()
