"""Summarises separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) into profiles/<round>_pmc_traffic.json.
HBM bytes per launch = 2 x FETCH_SIZE (gfx950 reports half of the streamed reads, MI355X_MICROARCH.md) + WRITE_SIZE,
both counters in KB.  usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>
<fetch_dir> / <write_dir> hold one sub-directory per workload (cfg2, cfg3, sp800, cfg5): kernels are keyed `k_name` for the
config-2 step (what bench.py looks up for its default line) and `<workload>:k_name` for the others; `workloads` holds, per
workload, the HBM bytes of ONE optimiser step = sum over its kernels of mean bytes x dispatches / number of Adam launches."""
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


out = {"command": "rocprofv3 --pmc FETCH_SIZE (then, separate run, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                  "[--workload cfg3|sprites800|cfg5] (short runs; tools/collect_profiles.sh)",
       "note": "separate passes per counter as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled (gfx950 reports half of "
               "streamed reads); the guide has no calibration for 8 B/lane loads, which is what the float64 kernels issue",
       "kernels": {}, "workloads": {}}
subs = sorted({os.path.basename(p) for root in sys.argv[1:3] for p in glob.glob(os.path.join(root, "*")) if os.path.isdir(p)}) or [""]
for w in subs:
    fetch, write = load(os.path.join(sys.argv[1], w), "FETCH_SIZE"), load(os.path.join(sys.argv[2], w), "WRITE_SIZE")
    tot, steps = 0.0, 0
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
        wv = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
        n = max(len(fetch.get(k, [])), len(write.get(k, [])))
        key = k if w in ("", "cfg2") else f"{w}:{k}"
        out["kernels"][key] = {"FETCH_SIZE_KB_mean": f, "dispatches_fetch": len(fetch.get(k, [])), "WRITE_SIZE_KB_mean": wv,
                               "dispatches_write": len(write.get(k, [])), "hbm_bytes_per_launch_corrected": (2 * f + wv) * 1024,
                               "hbm_bytes_total_corrected": (2 * f + wv) * 1024 * n, "dispatches": n}
        tot += (2 * f + wv) * 1024 * n
        if k.startswith("k_adam_tf1"):
            steps += n
    if w and steps:
        out["workloads"][w] = {"optimiser_steps_profiled": steps, "hbm_bytes_per_step_corrected": tot / steps,
                               "note": "all kernels of the run (incl. the one parity step without an Adam launch) / Adam launches"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernels:", len(out["kernels"]), "workloads:", list(out["workloads"]))
