#!/bin/bash
# Round 4: what does a plain streaming kernel reach on this box?  (tools/pmc_calib.hip: 256 MiB per launch, 16-byte loads / stores /
# non-temporal stores) -- durations from the kernel trace -> GB/s.  The B = 64 step's memory-bound kernels all sit at 2.7-2.9 TB/s.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04l; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/pmc_calib.hip -o $O/pmc_calib.bin || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace -f csv -d $O/kt -- $O/pmc_calib.bin > $O/run.out 2>&1
cd $R
python - <<PY | tee $O/stream_rates.txt
import csv, glob, collections
B = 256 << 20
agg = collections.defaultdict(list)
for f in glob.glob("$O/kt/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, v in sorted(agg.items()):
    if k in ("store16", "store8", "store_nt16", "load16"):
        print("%-12s launches %d  us %s  -> %.2f TB/s (best)" % (k, len(v), [round(x, 1) for x in v], B / min(v) / 1e6))
PY
rm -rf $O/kt $O/pmc_calib.bin
