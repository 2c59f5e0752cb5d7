#!/usr/bin/env python
"""Does a batch in flight change another batch's bits? Two submits in flight (slot 0: the batch, slot 1: the batch REVERSED), repeated, against
the synchronous call's rois / lines / heads -- per precision and conv_p64 setting. Prints which (repetition, slot, image) differ and by how much.

    CTPN_NO_TORCH=1 python tools/r6_pipeline_race.py [--reps 6] [--batch 32]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def diagnose(ctpn_amd, arena, imgs, rev, prec, options, reps, layers=None, dump=None):
    from oracle import network as N
    n, h, w = imgs.shape[:3]
    names = []
    for nm in N.CONVS:
        names.append(nm)
        if nm in N.POOL_AFTER:
            names.append(N.POOL_AFTER[nm])
    names += ["lstm_pre", "lstm_out", "rpn_bbox_pred"]
    opt = dict(options); opt["keep_acts"] = 1
    found = []
    with ctpn_amd.Context(0, n, h, w, prec, options=opt) as ctx:
        ctx.load_weights(arena)
        ctx.detect(rev)
        ref = {}
        pick = names[:6] + names[-3:] if not layers else [nm for nm in names if nm in layers]
        for nm in pick:
            ref[nm] = ctx.get_tensor(nm)
        for rep in range(reps):
            ctx.detect_submit(images=imgs, slot=0)
            ctx.detect_submit(images=rev, slot=1)
            ctx.detect_collect(0)
            ctx.detect_collect(1)
            for nm in ref:
                got = ctx.get_tensor(nm)
                if not np.array_equal(got, ref[nm]):
                    d = np.argwhere(got != ref[nm])
                    found.append({"rep": rep, "first_layer": nm, "elements": int(len(d)), "image_positions": sorted(set(int(x) for x in d[:, 0]))[:8],
                                  "y": [int(d[:, 1].min()), int(d[:, 1].max())], "x": [int(d[:, 2].min()), int(d[:, 2].max())],
                                  "c": [int(d[:, 3].min()), int(d[:, 3].max())], "max_abs_diff": float(np.abs(got - ref[nm]).max()),
                                  "max_abs_ref": float(np.abs(ref[nm]).max())})
                    if dump and not os.path.exists(dump):
                        # forensic patch: the differing outputs, the synchronous ones and the layer's input around them (which weights did the wave use?)
                        i0 = int(d[0, 0]); dd = d[d[:, 0] == i0]
                        y0, y1, x0, x1 = int(dd[:, 1].min()), int(dd[:, 1].max()), int(dd[:, 2].min()), int(dd[:, 2].max())
                        prev = pick[pick.index(nm) - 1] if pick.index(nm) > 0 else None
                        inp = ctx.get_tensor(prev)[n - 1 - i0 if False else i0] if prev else None
                        np.savez_compressed(dump, layer=nm, prev=str(prev), image_position=i0, y0=y0, x0=x0, got=got[i0, max(y0 - 4, 0):y1 + 5, max(x0 - 16, 0):x1 + 17],
                                            ref=ref[nm][i0, max(y0 - 4, 0):y1 + 5, max(x0 - 16, 0):x1 + 17], oy=max(y0 - 4, 0), ox=max(x0 - 16, 0),
                                            inp=(inp[max(y0 - 6, 0):y1 + 7, max(x0 - 18, 0):x1 + 19] if inp is not None else np.zeros(1)), iy=max(y0 - 6, 0), ix=max(x0 - 18, 0),
                                            inp_same=bool(prev and np.array_equal(ctx.get_tensor(prev), ref[prev])))
                    break
    return found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--variants", nargs="+", default=["split:conv_p64=1", "split:conv_p64=0", "bf16:"])
    ap.add_argument("--layers", nargs="+", default=None, help="--diagnose: the tensors to compare (default: the first six and the last three)")
    ap.add_argument("--dump", default=None, help="--diagnose: npz of the first differing patch (outputs, synchronous outputs, the layer's input around it)")
    ap.add_argument("--heads", action="store_true", help="also compare the second submit's network outputs (not its rois) with a synchronous forward")
    ap.add_argument("--diagnose", action="store_true", help="keep_acts = 1: after every pipelined pair compare the activations the second forward left behind with a "
                                                            "synchronous forward of the same batch, layer by layer; print where the first difference is")
    args = ap.parse_args()
    import ctpn_amd
    n, h, w = args.batch, 600, 900
    arena = ctpn_amd.make_synthetic_arena(0)
    imgs = ctpn_amd.weights.synthetic_images(n, h, w, 1)
    rev = imgs[::-1].copy()
    out = {}
    for v in args.variants:
        prec, opts = v.split(":")
        options = {k: int(x) for k, x in (kv.split("=") for kv in opts.split(",") if kv)}
        bad = []
        with ctpn_amd.Context(0, n, h, w, prec, options=options) as ctx:
            ctx.load_weights(arena)
            lines, rois = ctx.detect(imgs, want_rois=True)
            l2, r2 = ctx.detect(imgs, want_rois=True)
            same_sync = all(np.array_equal(a, b) for a, b in zip(rois, r2))
            for rep in range(args.reps):
                ctx.detect_submit(images=imgs, slot=0)
                ctx.detect_submit(images=rev, slot=1)
                l0, r0 = ctx.detect_collect(0, want_rois=True)
                l1, r1 = ctx.detect_collect(1, want_rois=True)
                for i in range(n):
                    for slot, rr in ((0, r0[i]), (1, r1[n - 1 - i])):
                        if rr.shape != rois[i].shape or not np.array_equal(rr, rois[i]):
                            d = float(np.abs(rr - rois[i]).max()) if rr.shape == rois[i].shape else -1.0
                            bad.append({"rep": rep, "slot": slot, "image": i, "position": i if slot == 0 else n - 1 - i, "max_abs_diff": d,
                                        "rows_differ": int((np.abs(rr - rois[i]).max(axis=1) > 0).sum()) if rr.shape == rois[i].shape else -1})
            if args.heads:
                # the network outputs of the SECOND submit against a synchronous forward of the same batch: independent of what the proposal tail
                # computes, so this also works with the diagnostic option debug_nms (parts of the one-workgroup NMS switched off, wrong rois)
                ctx.detect(rev)
                ref_h = {nm: ctx.get_tensor(nm) for nm in ("rpn_bbox_pred", "rpn_cls_prob_reshape")}
                hb = 0
                for rep in range(args.reps):
                    ctx.detect_submit(images=imgs, slot=0)
                    ctx.detect_submit(images=rev, slot=1)
                    ctx.detect_collect(0)
                    ctx.detect_collect(1)
                    hb += int(any(not np.array_equal(ctx.get_tensor(nm), ref_h[nm]) for nm in ref_h))
                print(v, "heads of the second submit differ in %d of %d pairs" % (hb, args.reps), flush=True)
            # synchronous call on the reversed batch: position dependence without anything in flight
            lr, rr_ = ctx.detect(rev, want_rois=True)
            pos_dep = [i for i in range(n) if not np.array_equal(rr_[n - 1 - i], rois[i])]
        out[v] = {"sync_repeat_identical": bool(same_sync), "sync_reversed_batch_differs_for_images": pos_dep, "pipelined_mismatches": len(bad), "first": bad[:12]}
        print(v, json.dumps(out[v]), flush=True)
        if args.diagnose:
            print(v, "diagnose:", json.dumps(diagnose(ctpn_amd, arena, imgs, rev, prec, options, args.reps, args.layers, args.dump)), flush=True)


if __name__ == "__main__":
    main()
