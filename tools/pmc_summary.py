"""Summarises two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) into profiles/<round>_pmc_traffic.json.
HBM bytes per launch = 2 x FETCH_SIZE (gfx950 reports half of the streamed reads, MI355X_MICROARCH.md) + WRITE_SIZE,
both counters in KB.  usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict

def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                # base name: no namespace, no `void`, no template arguments (instantiations of one kernel are averaged)
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0].strip()
                acc[name].append(float(r["Counter_Value"]))
    return acc

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE (then, separate run, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                  "--steps 20 --warmup 2 --no-cpu-baseline ; the same two passes for `--workload cfg5 --steps 2 --warmup 1 "
                  "--repeats 1` (tools/collect_profiles.sh)",
       "note": "separate passes per counter as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled (gfx950 reports half of "
               "streamed reads); the guide has no calibration for 8 B/lane loads, which is what these kernels issue",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
    w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
    out["kernels"][k] = {"FETCH_SIZE_KB_mean": f, "dispatches_fetch": len(fetch.get(k, [])), "WRITE_SIZE_KB_mean": w,
                         "dispatches_write": len(write.get(k, [])), "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernels:", len(out["kernels"]))
