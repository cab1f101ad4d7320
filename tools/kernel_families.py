"""Per-family table of a rocprofv3 kernel_stats.csv: python tools/kernel_families.py <kernel_stats.csv> <steps>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fam = collections.defaultdict(lambda: [0.0, 0])
def family(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(_kernel)?I", n)
    if m:
        return m.group(1)
    m = re.match(r"([A-Za-z0-9_:]+?)(_kernel)?[<(]", n)
    return (m.group(1) if m else n)[:50]
for r in rows:
    f = family(r["Name"])
    fam[f][0] += float(r["TotalDurationNs"]); fam[f][1] += int(r["Calls"])
tot = sum(v[0] for v in fam.values())
for f, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:28]:
    print("%-44s %7.1f launches/step %8.3f ms/step %5.1f%%" % (f, c / steps, t / 1e6 / steps, 100 * t / tot))
print("total %.2f ms/step" % (tot / 1e6 / steps))
