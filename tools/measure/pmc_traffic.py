#!/usr/bin/env python
"""profiles/pmc_traffic_<workload>.json from the per-kernel PMC sums (tools/measure/pmc_stats.py output):
HBM bytes per launch of the MFMA convolution kernels (conv_igemm_kernel + conv3x3_halo_kernel), FETCH_SIZE
corrected as MI355X_MICROARCH.md prescribes for gfx950.
usage: pmc_traffic.py <pmc_by_kernel.json> <workload> <source label> > profiles/pmc_traffic_<workload>.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import kernel_sources_sha256

src, workload, label = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.load(open(src))
# calibration of the two byte counters on known traffic (tools/pmc_calib.hip via gpu_pmc_calib.sh), if it has been measured:
# counter bytes / true bytes for 16-byte-per-lane loads (the guide: 0.5 on gfx950) and 16-byte-per-lane stores
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CAL = None
for cand in (os.path.join(REPO, "gpurun_out", "pmc_calib", "pmc_calibration.json"), os.path.join(REPO, "profiles", "pmc_calibration.json")):
    if os.path.exists(cand):
        CAL = json.load(open(cand))
        break
FR = (CAL or {}).get("fetch_ratio_16B") or 0.5
WR = (CAL or {}).get("write_ratio_16B") or 1.0
if not (0.3 < FR < 1.2):
    FR = 0.5
if not (0.3 < WR < 2.5):
    WR = 1.0
CONV = ("conv_igemm_kernel", "conv3x3_halo_kernel", "conv_wreg_kernel", "conv_seq_kernel")
per = {}
tot = {"launches": 0, "fetch": 0.0, "write": 0.0}
for k, c in d.items():
    if not any(s in k for s in CONV) or "FETCH_SIZE" not in c:
        continue
    n = c["FETCH_SIZE"]["dispatches"]
    f, w = c["FETCH_SIZE"]["sum"], c["WRITE_SIZE"]["sum"]
    hit, req = c.get("TCC_HIT_sum", {"sum": 0})["sum"], c.get("TCC_REQ_sum", {"sum": 1})["sum"]
    gui = c.get("GRBM_GUI_ACTIVE", {"sum": 0, "dispatches": 1})
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", {"sum": 0, "dispatches": 1})
    per[k] = {"dispatches": n, "fetch_kb_per_launch": round(f / n, 1), "write_kb_per_launch": round(w / n, 1),
              "hbm_bytes_per_launch_corrected": int((f / FR + w / WR) / n * 1024),
              "tcc_hit_rate": round(hit / max(1.0, req), 3),
              # matrix-pipe busy cycles per SIMD / shader cycles of the kernel (GRBM_GUI_ACTIVE counts every XCD: / 8); both per
              # (pass, dispatch) instance -- pmc_stats.py counts GUI_ACTIVE's instances over all the passes it rides in
              "mfma_util_est": round((mf["sum"] / max(1, mf["dispatches"]) / 1024.0) /
                                     (gui["sum"] / max(1, gui["dispatches"]) / 8.0 + 1e-9), 3),
              "lds_bank_conflict": c.get("SQ_LDS_BANK_CONFLICT", {"sum": 0})["sum"]}
    tot["launches"] += n; tot["fetch"] += f; tot["write"] += w
def one(name):
    for k, c in d.items():
        if name in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            nf, nw = c["FETCH_SIZE"]["dispatches"], c["WRITE_SIZE"]["dispatches"]
            f, w = c["FETCH_SIZE"]["sum"] / nf, c["WRITE_SIZE"]["sum"] / nw
            hit, req = c.get("TCC_HIT_sum", {"sum": 0, "dispatches": 1}), c.get("TCC_REQ_sum", {"sum": 1, "dispatches": 1})
            return {"kernel": k, "fetch_kb_per_launch": round(f, 1), "write_kb_per_launch": round(w, 1),
                    "hbm_bytes_per_launch_corrected": int((f / FR + w / WR) * 1024),
                    "tcc_hit_rate": round((hit["sum"] / max(1, hit["dispatches"])) / max(1.0, req["sum"] / max(1, req["dispatches"])), 3)}
    return None


out = {"workload": workload,
       "kernel_sources_sha256": kernel_sources_sha256(),
       "by_kernel": {n: one(n) for n in ("conv_seq_kernel", "chain_mask_kernel", "dw_xcorr_tall_kernel", "dw_xcorr_kernel") if one(n)},
       "source": "%s (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes, tools/measure/gpu_pmc.sh)" % label,
       "correction": ("FETCH_SIZE / %.3f, WRITE_SIZE / %.3f: counter bytes per true byte measured on 256 MiB of 16-byte-per-lane loads / "
                      "stores (tools/pmc_calib.hip, %s)" % (FR, WR, "profiles/pmc_calibration.json" if CAL else "NOT measured: the guide's 0.5 for "
                      "FETCH_SIZE, WRITE_SIZE as reported")),
       "calibration": {k: v for k, v in (CAL or {}).items() if k.endswith("B")} or None,
       "mfma_util_note": "(SQ_VALU_MFMA_BUSY_CYCLES per launch / 1024 SIMDs) / (GRBM_GUI_ACTIVE per launch / 8 XCDs): matrix-pipe busy "
                         "fraction of the kernel's shader cycles, averaged over all launches of the instantiation",
       "conv_igemm_family": {"kernels": list(CONV), "launches": tot["launches"],
                             "fetch_kb_per_launch": round(tot["fetch"] / tot["launches"], 1),
                             "write_kb_per_launch": round(tot["write"] / tot["launches"], 1),
                             "hbm_bytes_per_launch_corrected": int((tot["fetch"] / FR + tot["write"] / WR) / tot["launches"] * 1024)},
       "per_instantiation": per}
json.dump(out, sys.stdout, indent=1)
