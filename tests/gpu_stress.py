#!/usr/bin/env python
"""Determinism stress (GPU box): the same input through a fresh ctx N times must give bit-identical heads and rois.
usage: python tests/gpu_stress.py [precision] [n_ctx] [batch] [h] [w]   (env switches select kernel variants)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctpn_amd

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
h = int(sys.argv[4]) if len(sys.argv) > 4 else 600
w = int(sys.argv[5]) if len(sys.argv) > 5 else 900
LAYERS = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2",
          "conv5_3", "rpn_conv/3x3", "lstm_pre", "lstm_out"]
keep = os.environ.get("CTPN_KEEP_ACTS") == "1"
ref_acts = {}
arena = ctpn_amd.make_synthetic_arena(0)
imgs = ctpn_amd.weights.synthetic_images(n, h, w, 1)
ref = None
bad = 0
for r in range(reps):
    with ctpn_amd.Context(0, n, h, w, prec) as ctx:
        ctx.load_weights(arena)
        for inner in range(3):
            lines, rois = ctx.detect(imgs, want_rois=True)
            cp, bp = ctx.get_tensor("rpn_cls_prob_reshape"), ctx.get_tensor("rpn_bbox_pred")
            cur = (cp.copy(), bp.copy(), [x.copy() for x in rois])
            if ref is None:
                ref = cur
                if keep:
                    for nm in LAYERS:
                        try:
                            ref_acts[nm] = ctx.get_tensor(nm).copy()
                        except Exception as e:
                            pass
                continue
            same = np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1]) and all(np.array_equal(a, b) for a, b in zip(cur[2], ref[2]))
            if not same:
                bad += 1
                print("MISMATCH ctx %d call %d: max |dcls| %.3e  max |dbbox| %.3e" % (r, inner, np.abs(cur[0] - ref[0]).max(), np.abs(cur[1] - ref[1]).max()), flush=True)
                for nm, ra in ref_acts.items():
                    a = ctx.get_tensor(nm)
                    d = np.abs(a.astype(np.float64) - ra)
                    if d.max() > 0:
                        idx = np.argwhere(d > 0)
                        print("   first diverging layer %s: %d elements differ, max %.3e, bbox of diffs %s .. %s" % (nm, len(idx), d.max(), idx.min(0), idx.max(0)), flush=True)
                        break
print("stress %s n=%d %dx%d: %d mismatching calls of %d" % (prec, n, h, w, bad, reps * 3 - 1))
