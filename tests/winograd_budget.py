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

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def t_bf16(x):
    return torch.from_numpy(bf16_round(x.numpy()))


def conv3x3_relu_winograd(x, w_hwio, b):
    """x (1,H,W,Ci) fp32 holding bf16 values -> relu(conv + b), fp32 (not yet rounded). SAME padding, stride 1."""
    _, H, W, Ci = x.shape
    Co = w_hwio.shape[3]
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = np.zeros((2 * th + 2, 2 * tw + 2, Ci), np.float32)
    xp[1:H + 1, 1:W + 1] = x[0]
    xt = torch.from_numpy(xp).permute(2, 0, 1).unsqueeze(0)                       # (1, Ci, Hp, Wp)
    d = xt.unfold(2, 4, 2).unfold(3, 4, 2)[0]                                     # (Ci, th, tw, 4, 4)
    bt = torch.from_numpy(BT.astype(np.float32))
    V = torch.einsum("ij,cyxjk,lk->cyxil", bt, d, bt)                             # B^T d B, exact in fp32
    V = t_bf16(V.contiguous())
    U = np.einsum("ij,jkco,lk->ilco", G, w_hwio.astype(np.float64), G)            # (4,4,Ci,Co)
    U = torch.from_numpy(bf16_round(U.astype(np.float32)))
    Vm = V.permute(3, 4, 1, 2, 0).reshape(16, th * tw, Ci)                        # (16, tiles, Ci)
    Um = U.reshape(16, Ci, Co)
    M = torch.bmm(Vm, Um).reshape(4, 4, th * tw, Co)                              # fp32 accumulate
    at = torch.from_numpy(AT.astype(np.float32))
    Y = torch.einsum("ij,jktc,lk->tilc", at, M, at).reshape(th, tw, 2, 2, Co)     # (tiles, 2, 2, Co)
    y = Y.permute(0, 2, 1, 3, 4).reshape(2 * th, 2 * tw, Co)[:H, :W]
    y = torch.clamp(y + torch.from_numpy(np.asarray(b, np.float32)), min=0)
    return y.unsqueeze(0).numpy()


def conv3x3_relu_winograd_1d(x, w_hwio, b):
    """The 1-D form: F(2, 3) along x (4 multiplies per 2 outputs instead of 6), the three ky taps direct. Same rounding points: the
    transformed input row segments and the transformed weight rows are bf16 MFMA operands."""
    _, H, W, Ci = x.shape
    Co = w_hwio.shape[3]
    tw = (W + 1) // 2
    xp = np.zeros((H + 2, 2 * tw + 2, Ci), np.float32)
    xp[1:H + 1, 1:W + 1] = x[0]
    d = torch.from_numpy(xp).unfold(1, 4, 2)                                      # (H+2, tw, Ci, 4)
    bt = torch.from_numpy(BT.astype(np.float32))
    V = t_bf16(torch.einsum("ij,ytcj->ytic", bt, d).contiguous())                 # (H+2, tw, 4, Ci)
    U = np.einsum("ij,kjco->kico", G, w_hwio.astype(np.float64))                  # (3 ky, 4, Ci, Co)
    U = torch.from_numpy(bf16_round(U.astype(np.float32)))
    M = torch.zeros((H, tw, 4, Co), dtype=torch.float32)
    for ky in range(3):
        for f in range(4):
            M[:, :, f, :] += (V[ky:ky + H, :, f, :].reshape(H * tw, Ci) @ U[ky, f]).reshape(H, tw, Co)
    at = torch.from_numpy(AT.astype(np.float32))
    y = torch.einsum("if,ytfc->ytic", at, M).reshape(H, 2 * tw, Co)[:, :W]
    y = torch.clamp(y + torch.from_numpy(np.asarray(b, np.float32)), min=0)
    return y.unsqueeze(0).numpy()


def forward_emulated(img_u8, w, wino_layers, N, one_d=False):
    """All conv layers + lstm_pre in bf16 (the device's throughput configuration); the layers in `wino_layers` through Winograd."""
    x = N.image_blob(img_u8)
    for name in N.CONVS:
        wt = w[name + "/weights"]
        if name in wino_layers:
            x = (conv3x3_relu_winograd_1d if one_d else conv3x3_relu_winograd)(bf16_round(x), wt, w[name + "/biases"])
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
    # self-check of the transform on one layer: with NO rounding Winograd == the direct conv to fp32 noise
    rng = np.random.default_rng(0)
    xs = np.maximum(rng.standard_normal((1, 13, 18, 8)).astype(np.float32), 0)
    ws = rng.standard_normal((3, 3, 8, 5)).astype(np.float32) * 0.1
    global bf16_round
    keep = bf16_round
    bf16_round = lambda a: np.asarray(a, np.float32)                               # noqa: E731
    globals()["t_bf16"] = lambda t: t
    err = float(np.abs(conv3x3_relu_winograd(xs, ws, np.zeros(5, np.float32)) - N.conv3x3_relu(xs, ws, np.zeros(5, np.float32))).max())
    err = max(err, float(np.abs(conv3x3_relu_winograd_1d(xs, ws, np.zeros(5, np.float32)) - N.conv3x3_relu(xs, ws, np.zeros(5, np.float32))).max()))
    bf16_round = keep
    globals()["t_bf16"] = lambda t: torch.from_numpy(keep(t.numpy()))
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
