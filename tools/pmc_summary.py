#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (results.db per pass) of `bench.py` into profiles/<round>_pmc.json.

Passes (collected separately, as MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ...        rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
              SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace ...
Corrections: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read
stream (guide, section HBM), so read bytes = 2 * FETCH_SIZE * 1024 -- validated here against conv2_1's known input size.
GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).
usage: pmc_summary.py fetch.db write.db sq.db steps out.json
"""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    out = {}
    for name, val, cnt in db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        out[name] = (val, cnt)
    return out


def main(fetch_db, write_db, sq_db, steps, out_path):
    steps = int(steps)
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    sq = {c: per_kernel(sq_db, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                                            "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES")}
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if "ctpn::" not in name:
            continue
        f, nf = fetch.get(name, (0, 1))
        w, nw = write.get(name, (0, 1))
        e = {"launches_per_step": nf / steps, "hbm_read_bytes_per_launch": 2 * f * 1024 / max(nf, 1), "hbm_write_bytes_per_launch": w * 1024 / max(nw, 1)}
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        mf = sq["SQ_VALU_MFMA_BUSY_CYCLES"].get(name)
        ga = sq["GRBM_GUI_ACTIVE"].get(name)
        if mf and ga and ga[0] > 0:
            e["mfma_util"] = mf[0] / (ga[0] / 8.0 * 1024.0)
            wc = sq["SQ_WAVE_CYCLES"].get(name, (0, 1))[0]
            if wc:
                e["wave_cycle_split"] = {k: sq[c].get(name, (0, 1))[0] / wc for k, c in (("issue_stall", "SQ_WAIT_INST_ANY"), ("waitcnt_barrier", "SQ_WAIT_ANY"), ("active", "SQ_ACTIVE_INST_ANY"))}
            e["lds_bank_conflict_frac_of_busy"] = sq["SQ_LDS_BANK_CONFLICT"].get(name, (0, 1))[0] / max(sq["SQ_BUSY_CYCLES"].get(name, (1, 1))[0], 1)
        kernels[name] = e
    conv = [k for k in kernels if "conv3x3" in k]
    tot_bytes = sum(kernels[k]["hbm_bytes_per_launch"] * kernels[k]["launches_per_step"] for k in conv)
    # the ragged-column edge kernel runs INSIDE its layer's main launch (same layer, shared CUs): its bytes count, its launches do not
    tot_launch = sum(kernels[k]["launches_per_step"] for k in conv if "edge_kernel" not in k)
    out = {"steps_profiled": steps, "conv3x3_family": {"launches_per_step": tot_launch, "hbm_bytes_per_step": tot_bytes,
                                                       "hbm_bytes_per_launch": tot_bytes / max(tot_launch, 1)},
           "hbm_bytes_per_launch": tot_bytes / max(tot_launch, 1), "kernels": kernels}
    json.dump(out, open(out_path, "w"), indent=1)
    print("wrote", out_path, "conv family bytes/launch %.3e" % out["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main(*sys.argv[1:6])
