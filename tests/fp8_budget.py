#!/usr/bin/env python
"""A cheaper parity-grade arithmetic? Budget BEFORE any kernel (VERDICT r5 "next" 6; test infrastructure: imports oracle/; CPU only).

Split precision (CTPN_PREC_SPLIT) spends three bf16 MFMAs per product: x w ~= x_hi w_hi + x_lo w_hi + x_hi w_lo. CDNA4's MX-scaled FP8 MFMA
(v_mfma_scale_f32_32x32x64_f8f6f4) runs at twice the 16-bit rate. Candidate "f16 + 2 x mxfp8":

    x = x_hi + x_lo,  x_hi = fp16(x);   w = w_hi + w_lo,  w_hi = fp16(w)
    x w ~= x_hi w_hi  [fp16 MFMA, exact products]  +  q8(x_lo) q8(w_hi)  +  q8(x_hi) q8(w_lo)  [two MX-FP8 MFMAs: 2 x 0.5 = 1.0 MFMA-equivalent]

= 2.0 MFMA-equivalents per product instead of 3.0. q8 = OCP MX FP8: blocks of 32 consecutive K elements (here: 32 channels of one pixel /
one tap of one output channel) share an E8M0 power-of-two scale 2^(floor(log2 max|block|) - 8), elements are e4m3 (3 mantissa bits,
saturating at 448). The correction terms are ~2^-11 of the product and carry a relative error of ~2^-4 each: ~2^-15 of the product, against
~2^-17 for split-bf16's dropped x_lo w_lo.

This script emulates that arithmetic in the oracle for every conv layer but conv1_1 (exact integer pixels x (hi, lo) weights, as in split
precision) and for lstm_pre, on the benchmark images, against the fp32 oracle -- and, to calibrate the emulation, split-bf16 the same way
(the device measures cls_prob 3.4e-5 / 100 % / 100 % for it). Decision rule of the review: build the kernel only if cls_prob <= 2e-4 and rois
and lines stay at 100 %.

    python tests/fp8_budget.py --images 6 --out profiles/r06_fp8_correction_budget.json
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
import bf16_budget as BB  # noqa: E402


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def e4m3(v):
    """fp32 -> nearest e4m3 value (OCP FP8 E4M3: bias 7, subnormal quantum 2^-9, max 448, saturating), as fp32."""
    v = np.asarray(v, np.float32)
    a = np.abs(v)
    e = np.floor(np.log2(np.maximum(a, np.float32(2.0 ** -20))))
    e = np.clip(e, -6, 8)
    q = np.exp2(e - 3).astype(np.float32)
    r = np.rint(a / q) * q                     # round half to even on the 3-bit mantissa
    return (np.sign(v) * np.minimum(r, np.float32(448.0))).astype(np.float32)


def mxfp8(x, axis):
    """Quantise-dequantise along `axis` in blocks of 32 with a shared E8M0 scale (OCP MX: scale exponent = floor(log2 max|block|) - 8)."""
    x = np.moveaxis(np.asarray(x, np.float32), axis, -1)
    shp = x.shape
    assert shp[-1] % 32 == 0
    b = x.reshape(shp[:-1] + (shp[-1] // 32, 32))
    m = np.abs(b).max(axis=-1, keepdims=True)
    se = np.floor(np.log2(np.maximum(m, np.float32(2.0 ** -120)))) - 8
    se = np.clip(se, -127, 127)
    s = np.exp2(se).astype(np.float32)
    y = (e4m3(b / s) * s).astype(np.float32)
    y = np.where(m > 0, y, np.float32(0))
    return np.moveaxis(y.reshape(shp), -1, axis)


def conv_terms(N, x, w, b, scheme):
    """One conv3x3 + bias + ReLU in the named arithmetic (fp32 accumulation = the oracle op on the pre-rounded operands)."""
    zero = np.zeros_like(b)
    if scheme == "split_bf16":
        xh = BB.bf16_round(x); xl = BB.bf16_round(x - xh)
        wh = BB.bf16_round(w); wl = BB.bf16_round(w - wh)
        y = N.conv3x3_relu(xh, wh, b, relu=False) + N.conv3x3_relu(xl, wh, zero, relu=False) + N.conv3x3_relu(xh, wl, zero, relu=False)
    elif scheme == "f16_2xmxfp8":
        xh = f16(x); xl = x - xh
        wh = f16(w); wl = w - wh
        y = N.conv3x3_relu(xh, wh, b, relu=False)
        y = y + N.conv3x3_relu(mxfp8(xl, 3), mxfp8(wh, 2), zero, relu=False) + N.conv3x3_relu(mxfp8(xh, 3), mxfp8(wl, 2), zero, relu=False)
    elif scheme == "f16_only":
        y = N.conv3x3_relu(f16(x), f16(w), b, relu=False)
    else:
        raise ValueError(scheme)
    return np.maximum(y, np.float32(0)).astype(np.float32)


def forward(img, w, N, scheme):
    x = N.image_blob(img)
    for name in N.CONVS:
        if name == "conv1_1":
            x = N.conv3x3_relu(x, w[name + "/weights"], w[name + "/biases"])       # exact pixels x (hi, lo) weights in every parity-grade mode
        else:
            x = conv_terms(N, x, w[name + "/weights"], w[name + "/biases"], scheme)
        if name in N.POOL_AFTER:
            x = N.maxpool2x2(x)
    # lstm_pre: the same three terms on the [cells x 512] @ [512 x 1024] product; recurrence / FC / heads fp32
    n, hf, wf, c = x.shape
    pre = []
    for d in ("fw", "bw"):
        k = w["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d][:512]
        bb = w["lstm_o/bidirectional_rnn/%s/lstm_cell/bias" % d]
        xm = x.reshape(-1, c)
        if scheme == "split_bf16":
            xh = BB.bf16_round(xm); xl = BB.bf16_round(xm - xh); kh = BB.bf16_round(k); kl = BB.bf16_round(k - kh)
            y = xh @ kh + xl @ kh + xh @ kl + bb
        elif scheme == "f16_2xmxfp8":
            xh = f16(xm); xl = xm - xh; kh = f16(k); kl = k - kh
            y = xh @ kh + mxfp8(xl, 1) @ mxfp8(kh, 0) + mxfp8(xh, 1) @ mxfp8(kl, 0) + bb
        else:
            y = f16(xm) @ f16(k) + bb
        pre.append(y.astype(np.float32))
    pre = np.concatenate(pre, axis=1).reshape(n, hf, wf, 1024)
    lo = N.bilstm_from_pre(pre, w)
    fc = N.dense(lo, w["lstm_o/weights"], w["lstm_o/biases"])
    bbox = N.dense(fc, w["rpn_bbox_pred/weights"], w["rpn_bbox_pred/biases"])
    cls = N.pair_softmax(N.dense(fc, w["rpn_cls_score/weights"], w["rpn_cls_score/biases"]))
    return cls, bbox


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=6)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import ctpn_amd
    from oracle import network as N
    from oracle import postproc as P
    torch.set_grad_enabled(False)
    w = ctpn_amd.arena_views(ctpn_amd.make_synthetic_arena(0))
    h, wd = args.height, args.width
    schemes = ["split_bf16", "f16_2xmxfp8", "f16_only"]
    rows = {s: [] for s in schemes}
    for i in range(args.images):
        img = ctpn_amd.weights.synthetic_images(1, h, wd, 1 + i)
        ref_out = N.forward(img, w, keep=set())
        cls, bbox = ref_out["rpn_cls_prob_reshape"], ref_out["rpn_bbox_pred"]
        info = np.array([h, wd, 1.0], np.float32)
        rr = P.proposal_layer(cls, bbox, info)
        ref = {"cls": cls, "rois": rr, "lines": P.text_detect(rr[:, 1:5], rr[:, 0], (h, wd), "H")}
        for s in schemes:
            c, b = forward(img, w, N, s)
            m = BB.metrics(c, b, ref, P, h, wd)
            m["bbox_max"] = float(np.abs(b - bbox).max())
            rows[s].append(m)
            print(i, s, m, flush=True)
    agg = {s: {k: (float(np.max([r[k] for r in v])) if k in ("cls_max", "bbox_max") else float(np.mean([r[k] for r in v]))) for k in v[0]} for s, v in rows.items()}
    verdict = agg["f16_2xmxfp8"]["cls_max"] <= 2e-4 and agg["f16_2xmxfp8"]["roi_1px_1e-3"] == 1.0 and agg["f16_2xmxfp8"]["line_1px"] == 1.0
    out = {"images": args.images, "height": h, "width": wd, "method": __doc__.split("\n\n")[1].replace("\n", " "),
           "mfma_equivalents_per_product": {"split_bf16": 3.0, "f16_2xmxfp8": 2.0, "f16_only": 1.0},
           "schemes": agg, "per_image": rows,
           "decision_rule": "build only if cls_max <= 2e-4 and rois / lines at 100 %", "meets_rule": bool(verdict)}
    txt = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
