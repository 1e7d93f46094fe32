import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ends = [i for i, r in enumerate(rows) if "k_adam_tf1" in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1]
step = rows[lo:hi + 1]
t0 = step[0]["s"]
T = (step[-1]["e"] - t0) / 1e6
print(f"step: {len(step)} kernels, {T:.3f} ms")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:52]
a, b = float(sys.argv[2]), float(sys.argv[3])
for r in step:
    s, e = (r["s"] - t0) / 1e6, (r["e"] - t0) / 1e6
    if e >= a and s <= b and (r["e"] - r["s"]) > 8000:
        print(f"{s:8.3f} {e:8.3f} {(r['e'] - r['s']) / 1e3:8.1f} us  q{r['Queue_Id']}  {short(r['Kernel_Name'])}")
