"""fp16 instead of bf16 for the 16-bit operands and layer outputs of the throughput path? (test infrastructure: imports oracle/; CPU only)

Same emulation as tests/bf16_budget.py (every conv layer + lstm_pre round their operands and outputs), once with bf16 and once with fp16
(numpy float16, round-to-nearest-even), against the fp32 oracle. Both types run the MFMA at the same rate.

    python tests/fp16_budget.py          -> profiles/r03_fp16_vs_bf16_budget.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctpn_amd
from oracle import network as N
from oracle import postproc as P
import bf16_budget as BB
torch.set_grad_enabled(False)
def f16_round(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
w = ctpn_amd.arena_views(ctpn_amd.make_synthetic_arena(0))
h, wd = 600, 900
rows = {"bf16": [], "fp16": []}
for i in range(2):
    img = ctpn_amd.weights.synthetic_images(1, h, wd, 1 + i)
    cls, bbox = BB.forward_emulated(img, w, set(), N)
    info = np.array([h, wd, 1.0], np.float32)
    rr = P.proposal_layer(cls, bbox, info)
    ref = {"cls": cls, "rois": rr, "lines": P.text_detect(rr[:, 1:5], rr[:, 0], (h, wd), "H")}
    for name, rnd in (("bf16", BB.bf16_round), ("fp16", f16_round)):
        BB_bf = BB.bf16_round
        BB.bf16_round = rnd
        try:
            c, b = BB.forward_emulated(img, w, set(list(N.CONVS) + ["lstm_pre"]), N)
        finally:
            BB.bf16_round = BB_bf
        rows[name].append(BB.metrics(c, b, ref, P, h, wd))
        print(i, name, rows[name][-1], flush=True)
out = {k: {m: float(np.mean([r[m] for r in v])) if m != "cls_max" else float(np.max([r[m] for r in v])) for m in v[0]} for k, v in rows.items()}
print(json.dumps(out, indent=1))
json.dump({"images": 2, "height": h, "width": wd, "method": "tests/bf16_budget.forward_emulated with every conv layer + lstm_pre rounding operands and outputs to the named 16-bit type (fp16: numpy float16 RNE); against the fp32 oracle", "configs": out}, open(os.path.join(ROOT, 'profiles', 'r03_fp16_vs_bf16_budget.json'), 'w'), indent=1)
