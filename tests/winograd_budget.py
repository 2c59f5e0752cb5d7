#!/usr/bin/env python
"""What would Winograd F(2x2, 3x3) cost in accuracy on this network? (test infrastructure: imports oracle/; CPU only)

DESIGN.md section 7 names the algorithmic route (16 multiplies per 4 outputs instead of 36) as the one step the power-limited clock leaves
open. Before anybody writes that kernel family: this script EMULATES its bf16 arithmetic in the oracle and measures the end-to-end effect
the same way tests/bf16_budget.py does for the direct bf16 path.

Emulated device arithmetic of one layer (everything else as in bf16_budget.forward_emulated):
    U = G g G^T        from the fp32 weights, in double, rounded ONCE to bf16              (offline)
    V = B^T d B        on 4 x 4 input tiles (stride 2) of the bf16 activations: sums of up to four bf16 values, exact in fp32,
                       then rounded to bf16 -- the MFMA operand; THIS rounding is what the direct path does not have
    M = sum_c U . V    16 channel GEMMs, fp32 accumulate (the MFMA)
    Y = A^T M A        fp32, + bias, ReLU, rounded to bf16 like every layer's output

    python tests/winograd_budget.py --images 2 --out profiles/r03_winograd_budget.json
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
from bf16_budget import bf16_round, metrics  # noqa: E402

from oracle.winograd import conv3x3_relu_winograd_2d, conv3x3_relu_winograd_x  # noqa: E402


def forward_emulated(img_u8, w, wino_layers, N, one_d=False):
    """All conv layers + lstm_pre in bf16 (the device's throughput configuration); the layers in `wino_layers` through Winograd."""
    x = N.image_blob(img_u8)
    for name in N.CONVS:
        wt = w[name + "/weights"]
        if name in wino_layers:
            x = (conv3x3_relu_winograd_x if one_d else conv3x3_relu_winograd_2d)(bf16_round(x), wt, w[name + "/biases"])
        else:
            x = N.conv3x3_relu(x if name == "conv1_1" else bf16_round(x), bf16_round(wt), w[name + "/biases"])
        x = bf16_round(x)
        if name in N.POOL_AFTER:
            x = N.maxpool2x2(x)
    wl = dict(w)
    for d in ("fw", "bw"):
        k = w["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d].copy()
        k[:512] = bf16_round(k[:512])
        wl["lstm_o/bidirectional_rnn/%s/lstm_cell/kernel" % d] = k
    lo = N.bilstm(bf16_round(x), wl)
    fc = N.dense(lo, w["lstm_o/weights"], w["lstm_o/biases"])
    bbox = N.dense(fc, w["rpn_bbox_pred/weights"], w["rpn_bbox_pred/biases"])
    cls = N.pair_softmax(N.dense(fc, w["rpn_cls_score/weights"], w["rpn_cls_score/biases"]))
    return cls, bbox


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
    # self-check of the transforms: with NO rounding both forms == the direct conv to fp32 noise (also a test: tests/test_oracle.py)
    rng = np.random.default_rng(0)
    xs = np.maximum(rng.standard_normal((1, 13, 18, 8)).astype(np.float32), 0)
    ws = rng.standard_normal((3, 3, 8, 5)).astype(np.float32) * 0.1
    want = N.conv3x3_relu(xs, ws, np.zeros(5, np.float32))
    err = max(float(np.abs(f(xs, ws, np.zeros(5, np.float32), round_operands=False) - want).max()) for f in (conv3x3_relu_winograd_2d, conv3x3_relu_winograd_x))
    assert err < 1e-4, err
    convs = [c for c in N.CONVS if c != "conv1_1"]
    configs = [("all_bf16 direct (the device's throughput configuration)", set()),
               ("winograd F(2,3) on all 13 Ci >= 64 layers", set(convs)),
               ("winograd on conv1_2 .. conv3_3 only", {"conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3"}),
               ("winograd on conv4_1 .. rpn_conv only", {"conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_conv/3x3"})]
    configs += [("1-D winograd F(2,3) along x on all 13 Ci >= 64 layers", ("1d", set(convs))),
                ("1-D winograd along x on conv1_2 .. conv3_3 only", ("1d", {"conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3"}))]
    rows = {name: [] for name, _ in configs}
    from bf16_budget import forward_emulated as forward_direct
    for i in range(args.images):
        img = ctpn_amd.weights.synthetic_images(1, h, wd, 1 + i)
        cls, bbox = forward_direct(img, w, set(), N)                              # the fp32 oracle
        info = np.array([h, wd, 1.0], np.float32)
        rr = P.proposal_layer(cls, bbox, info)
        ref = {"cls": cls, "rois": rr, "lines": P.text_detect(rr[:, 1:5], rr[:, 0], (h, wd), "H")}
        for name, wl in configs:
            c, b = forward_emulated(img, w, wl[1], N, one_d=True) if isinstance(wl, tuple) else forward_emulated(img, w, wl, N)
            rows[name].append(metrics(c, b, ref, P, h, wd))
            print(i, name, rows[name][-1], flush=True)
    out = {"images": args.images, "height": h, "width": wd, "transform_self_check_max_abs_err_without_rounding": err,
           "method": " ".join(__doc__.split("\n\n")[2].split()),
           "configs": {name: {k: float(np.mean([r[k] for r in v])) if k != "cls_max" else float(np.max([r[k] for r in v])) for k in v[0]} for name, v in rows.items()}}
    txt = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
