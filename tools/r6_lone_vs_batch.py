#!/usr/bin/env python
"""A lone image against the same image inside a batch of four, tensor by tensor (which layer's bits depend on the batch size?).
    CTPN_NO_TORCH=1 python tools/r6_lone_vs_batch.py [--precisions bf16 split]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precisions", nargs="+", default=["bf16", "split"])
    ap.add_argument("--keep-acts", type=int, default=1)
    args = ap.parse_args()
    import ctpn_amd
    from oracle import network as N
    arena = ctpn_amd.make_synthetic_arena(0)
    imgs = ctpn_amd.weights.synthetic_images(4, 600, 900, 21)
    names = []
    for nm in N.CONVS:
        names.append(nm)
        if nm in N.POOL_AFTER:
            names.append(N.POOL_AFTER[nm])
    names += ["lstm_pre", "lstm_out", "rpn_bbox_pred", "rpn_cls_prob_reshape"]
    for prec in args.precisions:
        got = {}
        for n in (4, 1):
            with ctpn_amd.Context(0, n, 600, 900, prec, options={"keep_acts": args.keep_acts}) as ctx:
                ctx.load_weights(arena)
                lines, rois = ctx.detect(imgs if n == 4 else imgs[3:4], want_rois=True)
                t = {}
                for nm in names:
                    try:
                        t[nm] = ctx.get_tensor(nm)[3 if n == 4 else 0]
                    except Exception as e:      # not stored in this configuration
                        t[nm] = None
                got[n] = (t, rois[3 if n == 4 else 0], lines[3 if n == 4 else 0])
        out = []
        for nm in names:
            a, b = got[4][0][nm], got[1][0][nm]
            if a is None or b is None:
                out.append("%s: n/a" % nm)
            elif np.array_equal(a, b):
                out.append("%s: =" % nm)
            else:
                d = np.argwhere(a != b)
                out.append("%s: DIFFERS in %d elements, max |d| %.3g, first at %s" % (nm, len(d), float(np.abs(a - b).max()), d[0].tolist()))
        print(prec, "rois equal:", np.array_equal(got[4][1], got[1][1]), "lines equal:", np.array_equal(got[4][2], got[1][2]))
        print("   " + "\n   ".join(out), flush=True)


if __name__ == "__main__":
    main()
