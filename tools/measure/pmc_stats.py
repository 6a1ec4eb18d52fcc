"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (sum over dispatches).
A counter that is collected in SEVERAL passes (GRBM_GUI_ACTIVE rides in all of them) is summed over all of them, so its
dispatch count is the number of (pass, dispatch) instances -- round 3 counted unique dispatch ids, which made
GRBM_GUI_ACTIVE per dispatch 3x too large and `mfma_util_est` 3x too small."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(root, "pass*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[(k, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
out = {}
for k, cs in agg.items():
    out[k] = {c: {"sum": v, "dispatches": len(ndisp[(k, c)])} for c, v in cs.items()}
json.dump(out, open(os.path.join(root, "pmc_by_kernel.json"), "w"), indent=1)
for k, cs in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", {"sum": 0})["sum"])[:10]:
    print(k[:90])
    for c, v in cs.items():
        print("    %-32s %16.0f  (%d dispatches, %.1f per dispatch)" % (c, v["sum"], v["dispatches"], v["sum"] / max(1, v["dispatches"])))
