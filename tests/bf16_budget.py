#!/usr/bin/env python
"""bf16 error budget of the throughput path, on the CPU (test infrastructure: imports oracle/).

The device's bf16 configuration = every conv layer reads bf16 activations and bf16 weights, accumulates in fp32 and stores bf16; lstm_pre
reads the bf16 rpn_conv output and bf16 weights and stores fp32; recurrence, FC and heads are fp32. This script EMULATES that rounding in
the oracle (a product of two bf16 values is exact in fp32, so rounding operands and outputs of the fp32 oracle ops is the same
arithmetic up to summation order) and asks two questions VERDICT r2 left open:

  1. per-layer budget: with ONLY layer L rounded (everything else fp32), how far do rpn_cls_prob / the rois / the text lines move?
  2. mixed-precision points: which single relaxation of the all-bf16 configuration buys the most accuracy?

    python tests/bf16_budget.py --images 2 --out profiles/r03_bf16_budget.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def bf16_round(x):
    """fp32 -> nearest-even bf16 -> fp32 (numpy), the rounding of v_cvt_pk_bf16_f32."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def forward_emulated(img_u8, w, rounded, N):
    """Oracle forward with the layers named in `rounded` computed the way the bf16 device path computes them. Names: the 14 conv layers,
    'lstm_pre' (operands rounded, fp32 out). A conv layer rounds its weights and its OUTPUT; its input is whatever the previous layer
    produced (already bf16 if that layer was rounded too)."""
    x = N.image_blob(img_u8)
    for name in N.CONVS:
        wt = w[name + "/weights"]
        if name in rounded:
            # conv1_1 on the device: exact pixels (the means move to the bias in hi + lo form) x bf16 weights
            x = N.conv3x3_relu(x if name == "conv1_1" else bf16_round(x), bf16_round(wt), w[name + "/biases"])
            x = bf16_round(x)
        else:
            x = N.conv3x3_relu(x, wt, w[name + "/biases"])
        if name in N.POOL_AFTER:
            x = N.maxpool2x2(x)
    if "lstm_pre" in rounded:
        wl = dict(w)
        for d in ("fw", "bw"):
            k = w["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d].copy()
            k[:512] = bf16_round(k[:512])
            wl["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d] = k
        lo = N.bilstm(bf16_round(x), wl)
    else:
        lo = N.bilstm(x, w)
    fc = N.dense(lo, w["lstm_o/weights"], w["lstm_o/biases"])
    bbox = N.dense(fc, w["rpn_bbox_pred/weights"], w["rpn_bbox_pred/biases"])
    cls = N.pair_softmax(N.dense(fc, w["rpn_cls_score/weights"], w["rpn_cls_score/biases"]))
    return cls, bbox


def metrics(cls, bbox, ref, P, h, w):
    from util import match_rois
    from accuracy_report import _frac_lines
    info = np.array([h, w, 1.0], np.float32)
    rois = P.proposal_layer(cls, bbox, info)
    lines = P.text_detect(rois[:, 1:5], rois[:, 0], (h, w), "H")
    d = np.abs(cls - ref["cls"])
    return {"cls_max": float(d.max()), "cls_mean": float(d.mean()),
            "roi_1px_1e-3": match_rois(rois, ref["rois"], 1.0, 1e-3), "line_1px": _frac_lines(lines, ref["lines"], 1.0)[0]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import ctpn_amd
    from oracle import network as N
    from oracle import postproc as P
    torch.set_grad_enabled(False)
    arena = ctpn_amd.make_synthetic_arena(0)
    w = ctpn_amd.arena_views(arena)
    h, wd = args.height, args.width
    all_layers = list(N.CONVS) + ["lstm_pre"]
    configs = [("all_bf16 (the device's throughput configuration)", set(all_layers))]
    configs += [("only " + l, {l}) for l in all_layers]
    configs += [("all_bf16 except " + l, set(all_layers) - {l}) for l in ("lstm_pre", "rpn_conv/3x3", "conv5_3", "conv1_1", "conv1_2")]
    configs += [("all_bf16 except rpn_conv/3x3 + lstm_pre", set(all_layers) - {"rpn_conv/3x3", "lstm_pre"}),
                ("all_bf16 except conv5_x + rpn_conv/3x3 + lstm_pre", set(all_layers) - {"conv5_1", "conv5_2", "conv5_3", "rpn_conv/3x3", "lstm_pre"}),
                ("all_bf16 except conv4_x .. lstm_pre", set(all_layers) - {"conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_conv/3x3", "lstm_pre"}),
                ("bf16 conv1_1 .. conv3_3 only", {"conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3"}),
                ("bf16 conv1_1 .. conv2_2 only", {"conv1_1", "conv1_2", "conv2_1", "conv2_2"})]
    rows = {name: [] for name, _ in configs}
    for i in range(args.images):
        img = ctpn_amd.weights.synthetic_images(1, h, wd, 1 + i)
        cls, bbox = forward_emulated(img, w, set(), N)
        info = np.array([h, wd, 1.0], np.float32)
        rr = P.proposal_layer(cls, bbox, info)
        ref = {"cls": cls, "rois": rr, "lines": P.text_detect(rr[:, 1:5], rr[:, 0], (h, wd), "H")}
        for name, rounded in configs:
            c, b = forward_emulated(img, w, rounded, N)
            rows[name].append(metrics(c, b, ref, P, h, wd))
            print(i, name, rows[name][-1], flush=True)
    out = {"images": args.images, "height": h, "width": wd, "method": __doc__.split("\n\n")[1].replace("\n", " "),
           "configs": {name: {k: float(np.mean([r[k] for r in v])) if k != "cls_max" else float(np.max([r[k] for r in v])) for k in v[0]} for name, v in rows.items()}}
    txt = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
