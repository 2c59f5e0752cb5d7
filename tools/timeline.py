#!/usr/bin/env python
"""Timeline of ONE bench step out of a rocprofv3 kernel-trace database: every dispatch with start / end relative to the step's
first kernel, so that gaps between dependent launches and overlap of the side streams show.  usage: timeline.py results.db [step]"""
import sqlite3
import sys


def main(db_path, step=3):
    db = sqlite3.connect(db_path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    t = sorted([x for x in tabs if "kernel_dispatch" in x], key=len)[0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
    ks = [x for x in tabs if "kernel_symbol" in x][0]
    kc = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    idc = "id" if "id" in kc else kc[0]
    nmc = "kernel_name" if "kernel_name" in kc else [c for c in kc if "name" in c][0]
    names = dict(db.execute("select %s,%s from %s" % (idc, nmc, ks)))
    kid = "kernel_id" if "kernel_id" in cols else [c for c in cols if "kernel" in c][0]
    st = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    en = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    qc = [c for c in cols if "queue" in c or "stream" in c]
    rows = list(db.execute("select %s,%s,%s%s from %s order by %s" % (kid, st, en, ("," + qc[0]) if qc else "", t, st)))
    # a step starts at its first kernel: image_to_q (uint8 feed of the 16-bit modes) or one of the conv_first kernels
    starts = [i for i, r in enumerate(rows) if "image_to_q" in names.get(r[0], "")]
    if not starts:
        starts = [i for i, r in enumerate(rows) if "conv_first" in names.get(r[0], "")]
    if len(starts) <= step + 1:
        step = max(0, len(starts) - 2)
    lo, hi = starts[step], starts[step + 1]
    t0 = rows[lo][1]
    prev_end = None
    for r in rows[lo:hi]:
        nm = names.get(r[0], str(r[0])).replace("ctpn::", "")
        main_stream = any(k in nm for k in ("image_to_q", "conv_first", "conv3x3", "lstm_pre", "bilstm")) or ("igemm" in nm)
        gap = ""
        if main_stream and prev_end is not None and "igemm_kernel<ctpn::bf16_s, ctpn::bf16_s" not in nm:
            gap = "gap %+7.1f" % ((r[1] - prev_end) / 1e3)
        print("%9.1f %9.1f %8.1f  q=%s %-12s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qc else "-", gap, nm[:70]))
        if main_stream and "igemm_kernel<ctpn::bf16_s, ctpn::bf16_s" not in nm:
            prev_end = r[2]
    print("step span us:", (rows[hi][1] - t0) / 1e3)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
