"""Tuning aid: the kernels of ONE optimiser step of a rocprofv3 --kernel-trace csv inside a time window, with their queues.
usage: trace_window.py kernel_trace.csv <step index> <from us> <to us> <min duration us>   (steps end with k_adam_tf1; pick a TIMED step --
the last steps of a bench.py run are the one-stream / per-stage instrumented ones)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ends = [i for i, r in enumerate(rows) if "k_adam_tf1" in r["Kernel_Name"]]
k = int(sys.argv[2])
lo, hi = ends[k - 1] + 1, ends[k]
step = rows[lo:hi + 1]
t0 = rows[ends[k - 1]]["e"]
T = (step[-1]["e"] - t0) / 1e3
print("step", k, len(step), "kernels", T, "us; per-step:", [round((rows[b]['e'] - rows[a]['e']) / 1e6, 2) for a, b in zip(ends[:-1], ends[1:])])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:56]
a, b = float(sys.argv[3]), float(sys.argv[4])
for r in step:
    s, e = (r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3
    if e < a or s > b or e - s < float(sys.argv[5]): continue
    print(f"{s:8.1f} {e:8.1f} {e-s:7.1f} us q{r['Queue_Id']} {short(r['Kernel_Name'])}")
