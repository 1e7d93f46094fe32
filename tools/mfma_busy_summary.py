"""Matrix-pipe busy share and wave wait shares per kernel from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY` pass per workload (sub-directories of <dir>).
  mfma_busy    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)     share of SIMD cycles with the matrix pipe busy
  parked       = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                             wave lifetime at s_waitcnt / s_barrier
  issue_stalled= SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                                        waiting to issue (matrix pipe busy, RAW)
usage: mfma_busy_summary.py <dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                  "SQ_ACTIVE_INST_ANY --output-format csv -- python3 bench.py --workload <w> ... (tools/collect_profiles.sh)",
       "note": __doc__.split("usage")[0].strip(), "kernels": {}}
for wdir in sorted(glob.glob(os.path.join(sys.argv[1], "*"))):
    w = os.path.basename(wdir)
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(wdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if not m.get("GRBM_GUI_ACTIVE") or not m.get("SQ_WAVE_CYCLES"):
            continue
        out["kernels"][f"{w}: {k}"] = {
            "dispatches": len(next(iter(c.values()))),
            "mfma_busy": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0),
            "parked": m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"],
            "issue_stalled": m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"],
            "cycles_per_xcd": m["GRBM_GUI_ACTIVE"] / 8.0}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("kernels:", len(out["kernels"]))
