#!/bin/bash
# Calibrate FETCH_SIZE / WRITE_SIZE on known byte counts (tools/pmc_calib.hip) -> gpurun_out/pmc_calib/pmc_calibration.json
# (copied to profiles/pmc_calibration.json by the final script; tools/measure/pmc_traffic.py applies it).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/pmc_calib; mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/pmc_calib.hip -o $O/pmc_calib.bin || exit 1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$C
  timeout 120 rocprofv3 --pmc $C --kernel-trace -f csv -d $O/$C -- $O/pmc_calib.bin > $O/$C.out 2> $O/$C.err
  echo "calib pass $C exit $?"
done
cd $R
python - <<PY
import csv, glob, json, collections
BYTES = 256 << 20
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
out = {"bytes_per_launch": BYTES, "unit_note": "counter values are KB (x 1024 = bytes)", "kernels": {}}
for k, cs in agg.items():
    out["kernels"][k] = {c: {"per_launch": sum(v) / len(v), "launches": len(v), "ratio_counter_bytes_over_true": sum(v) / len(v) * 1024 / BYTES}
                         for c, v in cs.items()}
g = lambda k, c: out["kernels"].get(k, {}).get(c, {}).get("ratio_counter_bytes_over_true")
out["write_ratio_16B"] = g("store16", "WRITE_SIZE")
out["write_ratio_8B"] = g("store8", "WRITE_SIZE")
out["write_ratio_nt16B"] = g("store_nt16", "WRITE_SIZE")
out["fetch_ratio_16B"] = g("load16", "FETCH_SIZE")
json.dump(out, open("$O/pmc_calibration.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("B")}))
PY
rm -f $O/pmc_calib.bin; find $O -name "*.csv" -size +2M -delete
