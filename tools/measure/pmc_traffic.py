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
def per_launch(c, name, default=0.0):
    """a counter's value PER LAUNCH with the counter's OWN dispatch count: the passes do not see the same number of dispatches
    (a timed-out pass, a warm-up launch more or less), so dividing one pass's sum by another pass's count is wrong -- round 4's
    per_instantiation table divided the WRITE_SIZE / TCC sums (407 dispatches) by the FETCH pass's 327"""
    v = c.get(name)
    return v["sum"] / max(1, v["dispatches"]) if v else default


for k, c in d.items():
    if not any(s in k for s in CONV) or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    n = c["FETCH_SIZE"]["dispatches"]
    f, w = per_launch(c, "FETCH_SIZE"), per_launch(c, "WRITE_SIZE")          # KB per launch, each by its own pass's count
    per[k] = {"dispatches": {x: c[x]["dispatches"] for x in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_REQ_sum") if x in c},
              "fetch_kb_per_launch": round(f, 1), "write_kb_per_launch": round(w, 1),
              "hbm_bytes_per_launch_corrected": int((f / FR + w / WR) * 1024),
              "tcc_hit_rate": round(per_launch(c, "TCC_HIT_sum") / max(1.0, per_launch(c, "TCC_REQ_sum", 1.0)), 3),
              # matrix-pipe busy cycles per SIMD / shader cycles of the kernel (GRBM_GUI_ACTIVE counts every XCD: / 8); both per
              # (pass, dispatch) instance -- pmc_stats.py counts GUI_ACTIVE's instances over all the passes it rides in
              "mfma_util_est": round((per_launch(c, "SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0) / (per_launch(c, "GRBM_GUI_ACTIVE") / 8.0 + 1e-9), 3),
              "lds_bank_conflict_per_launch": round(per_launch(c, "SQ_LDS_BANK_CONFLICT"), 1)}
    # family totals: per-launch figures weighted by the FETCH pass's launch count
    tot["launches"] += n; tot["fetch"] += f * n; tot["write"] += w * n
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
