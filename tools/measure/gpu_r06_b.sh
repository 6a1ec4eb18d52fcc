#!/bin/bash
# Round 6: where conv_pp_kernel's interval goes: K-loop ablations (MEASURE build) + SQ counters of conv_pp vs conv_wreg on l3.0.ds at B = 64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06b; rm -rf $O; mkdir -p $O
SMK_LIB=$R/build_variants/measure/siammask_amd/libsiammask_hip.so timeout 900 python tools/measure/gpu_pp_ablate.py 64 20 2>&1 | grep -v amdgpu.ids | tee $O/pp_ablate_b64.txt
cd /tmp && export TMPDIR=/tmp
for w in pp wreg; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
     -d $O/pmc_$w -o pmc --output-format csv -- python $R/tools/measure/gpu_pp_one.py $w l3.0.ds 64 10 > $O/pmc_$w.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc_$w/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:40]
        if "conv_pp" not in k and "conv_wreg" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    disp = max(n[(k, c)] for c in d)
    per = {c: v / n[(k, c)] for c, v in d.items()}
    print("$w", k, "dispatches", disp)
    for c, v in sorted(per.items()): print("   %-28s %.4g" % (c, v))
    if "SQ_WAVE_CYCLES" in per:
        wc = per["SQ_WAVE_CYCLES"]
        print("   wait_any/wave %.3f  wait_inst/wave %.3f  active/wave %.3f  mfma_busy / (GUI_ACTIVE*1024/8...) see raw" % (per.get("SQ_WAIT_ANY",0)/wc, per.get("SQ_WAIT_INST_ANY",0)/wc, per.get("SQ_ACTIVE_INST_ANY",0)/wc))
PY
done 2>&1 | tee $O/pmc_summary.txt
