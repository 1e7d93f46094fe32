"""Mean of arbitrary rocprofv3 --pmc counters per kernel: pmc_counters.py <dir> [name-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if len(sys.argv) > 2 and not any(t in k for t in sys.argv[2:]):
        continue
    print(k[:60], {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, "n=%d" % len(next(iter(acc[k].values()))))
