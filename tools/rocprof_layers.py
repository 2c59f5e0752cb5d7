#!/usr/bin/env python
"""Per-launch timeline of one bench step out of a rocprofv3 results database (rocprofv3 --kernel-trace ... _results.db):
the dispatches of the main stream's longest-running kernels are grouped by their position inside a step and averaged
over the steps.  usage: rocprof_layers.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tabs if "kernel_dispatch" in t]
    if not disp:
        print("tables:", tabs)
        return
    t = sorted(disp, key=len)[0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
    print("# table", t, cols)
    ksym = [x for x in tabs if "kernel_symbol" in x]
    names = {}
    if ksym:
        kc = [r[1] for r in db.execute("pragma table_info(%s)" % ksym[0])]
        idc = "id" if "id" in kc else kc[0]
        nmc = "kernel_name" if "kernel_name" in kc else [c for c in kc if "name" in c][0]
        for i, n in db.execute("select %s, %s from %s" % (idc, nmc, ksym[0])):
            names[i] = n
    kid = "kernel_id" if "kernel_id" in cols else [c for c in cols if "kernel" in c][0]
    st = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    en = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    gx = [c for c in cols if c in ("grid_size_x", "grid_x")]
    q = "select %s, %s, %s%s from %s order by %s" % (kid, st, en, (", " + gx[0]) if gx else "", t, st)
    rows = list(db.execute(q))
    out = []
    main = ("image_to_q", "conv_first", "conv3x3", "lstm_pre", "igemm", "bilstm")     # the forward stream; the proposal stream's kernels interleave freely
    for r in rows:
        nm = names.get(r[0], str(r[0]))
        if any(m in nm for m in main):
            out.append((nm, r[1], r[2], r[3] if gx else 0))
    # a step starts at its first kernel: image_to_q (uint8 feed of the 16-bit modes), else one of the conv_first kernels
    first = "image_to_q" if any("image_to_q" in n for n, _, _, _ in out) else "conv_first"
    steps, cur = [], None
    for n, s, e, g in out:
        if first in n:
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append((n, s, e, g))
    steps = [s for s in steps if len(s) == max(len(x) for x in steps)]
    if not steps:
        print("no complete step found")
        return
    k = len(steps[0])
    res = []
    for i in range(k):
        nm = steps[-1][i][0]
        short = nm.split("(")[0].replace("ctpn::", "")[:110]
        durs = [(s[i][2] - s[i][1]) / 1e3 for s in steps if s[i][0] == nm]
        res.append((i, short, steps[-1][i][3], sum(durs) / max(len(durs), 1), len(durs)))
    for r in res:
        print("%3d %-112s grid %8d  %9.1f us  (n=%d)" % r)
    if out_path:
        with open(out_path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["pos", "kernel", "grid_x", "avg_us", "samples"])
            w.writerows(res)


if __name__ == "__main__":
    main(*sys.argv[1:])
