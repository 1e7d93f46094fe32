"""Tuning aid: top kernels of a rocprofv3 --kernel-trace --stats run (kernel_stats.csv), per step.  usage: kstats_top.py csv steps [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / steps / 1e6:.2f} ms")
for r in rows[:n]:
    print(r["Name"][:110].ljust(110), f'{float(r["Calls"]) / steps:6.1f} x {float(r["AverageNs"]) / 1e3:8.1f} us = {float(r["TotalDurationNs"]) / steps / 1e6:6.2f} ms  {100 * float(r["TotalDurationNs"]) / tot:5.1f} %')
