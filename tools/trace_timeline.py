"""Tuning aid: per-queue timeline of the LAST optimiser step in a rocprofv3 --kernel-trace csv.
usage: trace_timeline.py kernel_trace.csv [marker_kernel]   (a step ends with the Adam launch k_adam_tf1)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adam_tf1"
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1]
step = rows[lo:hi + 1]
t0 = step[0]["s"]
print(f"step: {len(step)} kernels, {(step[-1]['e'] - t0) / 1e6:.3f} ms")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:58]
byq = collections.defaultdict(list)
for r in step:
    byq[(r["Queue_Id"], r["Stream_Id"])].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"--- queue {q[0]} stream {q[1]}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms, span {(rs[0]['s'] - t0) / 1e6:.2f} .. {(rs[-1]['e'] - t0) / 1e6:.2f} ms")
if len(sys.argv) > 3:
    for r in step:
        if (r["e"] - r["s"]) > 40000:
            print(f"{(r['s'] - t0) / 1e6:8.3f} {(r['e'] - t0) / 1e6:8.3f} {(r['e'] - r['s']) / 1e3:8.1f} us  q{r['Queue_Id']}/s{r['Stream_Id']}  {short(r['Kernel_Name'])}")
