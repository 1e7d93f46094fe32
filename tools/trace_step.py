import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ends = [i for i, r in enumerate(rows) if "k_adam_tf1" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -8
lo, hi = ends[k - 1] + 1, ends[k]
step = rows[lo:hi + 1]
t0 = step[0]["s"]
print(f"step: {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
qs = sorted({(r["Queue_Id"], r["Stream_Id"]) for r in step})
for r in step:
    col = qs.index((r["Queue_Id"], r["Stream_Id"]))
    print(f"{(r['s'] - t0) / 1e3:8.1f} {(r['e'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f}  " + "                                  " * col + short(r["Kernel_Name"]))
