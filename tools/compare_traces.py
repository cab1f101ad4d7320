"""In-step comparison of two rocprofv3 kernel traces of the same bench command run with two library builds (tools/ab_routing.sh
variants): the GEMM launches are matched by their position in the launch sequence, and every launch whose kernel differs between
the two runs is listed with both durations.

    python tools/compare_traces.py A_kernel_trace.csv C_kernel_trace.csv [skip_first_n_launches]
"""
import collections
import csv
import re
import sys


def gemms(path):
    out = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if re.search(r"gemm_(dma_|p8n?_)?kernel", n):
            short = re.sub(r"\(anonymous namespace\)::|void |\(esvit_gemm_desc.*", "", n)
            out.append((short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, r["Grid_Size_X"], r["Grid_Size_Y"]))
    return out


a, c = gemms(sys.argv[1]), gemms(sys.argv[2])
print("GEMM launches:", len(a), len(c))
n = min(len(a), len(c))
agg = collections.OrderedDict()
for i in range(n):
    if a[i][0] != c[i][0]:
        key = (a[i][0], c[i][0], c[i][2], c[i][3])
        agg.setdefault(key, []).append((a[i][1], c[i][1]))
tot_a = tot_c = 0.0
for (ka, kc, gx, gy), v in agg.items():
    sa, sc = sum(x for x, _ in v), sum(y for _, y in v)
    tot_a += sa
    tot_c += sc
    print("%-36s vs %-58s grid(%s,%s) x%-3d  A %8.1f us  C %8.1f us  (%+.1f %%)" % (ka[:36], kc[:58], gx, gy, len(v), sa / len(v), sc / len(v), 100 * (sa / sc - 1)))
print("launches that differ: A %.2f ms, C %.2f ms in this trace" % (tot_a / 1e3, tot_c / 1e3))
sa, sc = sum(x[1] for x in a[:n]), sum(x[1] for x in c[:n])
print("all GEMM launches: A %.2f ms, C %.2f ms" % (sa / 1e3, sc / 1e3))
