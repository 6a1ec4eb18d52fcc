"""Per kernel name: dispatches, mean duration, GRBM_GUI_ACTIVE per dispatch, effective clock = counter / duration."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
dur = {}
name = {}
for f in glob.glob(os.path.join(root, "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        d = r["Dispatch_Id"]
        if d not in dur:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += dur[d]
        a[2] += float(r["Counter_Value"])
for k, (n, us, cyc) in sorted(agg.items()):
    if n:
        print("%-90s n %3d  %9.1f us  %12.0f cycles  %6.0f MHz" % (k[:90], n, us / n, cyc / n, cyc / us))
