#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats CSV -> profiles/rocprofv3_kernel_stats_<workload>.json, stamped with the sha256 of the kernel
sources it was measured on (tools/measure/src_hash.py); bench.py attaches it to `roofline.rocprofv3` only when that hash is
the hash of the sources it runs on (a profile of other sources is refused, not quoted).
usage: rocprof_stats_json.py <kernel_stats.csv> <workload> "<command line that was profiled>" > out.json"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import kernel_sources_sha256

src, workload, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
for r in csv.DictReader(open(src)):
    rows.append({"name": r["Name"], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                 "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
                 "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "percentage": float(r["Percentage"])})
json.dump({"workload": workload, "kernel_sources_sha256": kernel_sources_sha256(), "command": cmd,
           "source_commit": os.environ.get("SMK_SOURCE_COMMIT"), "kernels": rows}, sys.stdout, indent=1)
