"""Idle time between kernels of one stream in a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv> [steps]
Prints per-step busy time, the gaps between consecutive dispatches (count, sum, histogram) and the kernels that precede the
largest share of gap time."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows), key=lambda t: t[0])
byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
for q, lst in byq.items():
    busy = sum(e[1] - e[0] for e in lst)
    gaps, after = [], collections.Counter()
    for a, b in zip(lst, lst[1:]):
        g = b[0] - a[1]
        if 0 < g < 200000:  # ignore host-side pauses between steps
            gaps.append(g)
            after[a[2][:60]] += g
    print("queue %s: %d kernels, busy %.2f ms/step, gaps %d (%.2f ms/step, mean %.2f us)" % (q, len(lst) // steps, busy / 1e6 / steps, len(gaps) // steps,
                                                                                          sum(gaps) / 1e6 / steps, sum(gaps) / max(len(gaps), 1) / 1e3))
    hist = collections.Counter(min(int(g / 1000), 20) for g in gaps)
    print("   gap histogram (us: count/step):", {k: v // steps for k, v in sorted(hist.items())})
    for k, v in after.most_common(8):
        print("   after %-60s %.3f ms/step" % (k, v / 1e6 / steps))
