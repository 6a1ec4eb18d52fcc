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
              "hbm_bytes_per_launch_corrected": int((2 * f + w) / n * 1024),
              "tcc_hit_rate": round(hit / max(1.0, req), 3),
              "mfma_util_est": round((mf["sum"] / max(1, mf["dispatches"])) /
                                     (1024.0 * gui["sum"] / max(1, gui["dispatches"]) / 8.0 + 1e-9), 3),
              "lds_bank_conflict": c.get("SQ_LDS_BANK_CONFLICT", {"sum": 0})["sum"]}
    tot["launches"] += n; tot["fetch"] += f; tot["write"] += w
def one(name):
    for k, c in d.items():
        if name in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            nf, nw = c["FETCH_SIZE"]["dispatches"], c["WRITE_SIZE"]["dispatches"]
            f, w = c["FETCH_SIZE"]["sum"] / nf, c["WRITE_SIZE"]["sum"] / nw
            hit, req = c.get("TCC_HIT_sum", {"sum": 0, "dispatches": 1}), c.get("TCC_REQ_sum", {"sum": 1, "dispatches": 1})
            return {"kernel": k, "fetch_kb_per_launch": round(f, 1), "write_kb_per_launch": round(w, 1),
                    "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024),
                    "tcc_hit_rate": round((hit["sum"] / max(1, hit["dispatches"])) / max(1.0, req["sum"] / max(1, req["dispatches"])), 3)}
    return None


out = {"workload": workload,
       "kernel_sources_sha256": kernel_sources_sha256(),
       "by_kernel": {n: one(n) for n in ("conv_seq_kernel", "chain_mask_kernel", "dw_xcorr_tall_kernel", "dw_xcorr_kernel") if one(n)},
       "source": "%s (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes, tools/measure/gpu_pmc.sh)" % label,
       "correction": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); "
                     "WRITE_SIZE as reported (uncalibrated)",
       "mfma_util_note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8 XCDs): matrix-pipe busy fraction "
                         "averaged over all launches of the instantiation",
       "conv_igemm_family": {"kernels": list(CONV), "launches": tot["launches"],
                             "fetch_kb_per_launch": round(tot["fetch"] / tot["launches"], 1),
                             "write_kb_per_launch": round(tot["write"] / tot["launches"], 1),
                             "hbm_bytes_per_launch_corrected": int((2 * tot["fetch"] + tot["write"]) / tot["launches"] * 1024)},
       "per_instantiation": per}
json.dump(out, sys.stdout, indent=1)
