#!/usr/bin/env python
"""Random and hostile inputs to the host-side C-ABI entry points that take caller data (no GPU): ctpn_text_lines on the host connector
(device_id = -1), ctpn_result_text, ctpn_draw_boxes, ctpn_resize_dims -- boxes with NaN / inf / huge / negative / inverted coordinates,
scores outside [0, 1], empty inputs, tiny capacities, images of one pixel. Run it under the AddressSanitizer build (tools/run_fuzz.sh does):
an error return is fine, a sanitizer report or a crash is a bug.
    python tools/fuzz_host_entry_points.py SEED SECONDS
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctpn_amd  # noqa: E402,F401
from ctpn_amd import _binding as B  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
T = float(sys.argv[2]) if len(sys.argv) > 2 else 30
SPECIAL = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e9, -1e9, 0.0, -0.0, 0.5, 2147483647.0, -2147483648.0, 65535.0])


def hostile(shape, scale):
    a = rng.normal(0, scale, shape)
    m = rng.random(shape) < rng.choice([0.0, 0.02, 0.3])
    a[m] = rng.choice(SPECIAL, int(m.sum()))
    return a


def plausible_boxes(n, h, w):
    """proposal-like boxes: 16 wide, on the anchor grid, so that the graph builder finds neighbours and chains"""
    x = rng.integers(0, max(w // 16, 1), n) * 16.0
    y = rng.uniform(0, h, n)
    hh = rng.uniform(4, 60, n)
    return np.stack([x, y, x + 15, y + hh], 1)


t0, n, errs = time.time(), 0, 0
while time.time() - t0 < T:
    k = int(rng.integers(0, 4))
    try:
        if k == 0:
            cnt = int(rng.choice([0, 1, 2, 7, 50, 300, 1000]))
            h, w = int(rng.choice([1, 16, 600, 65535])), int(rng.choice([1, 16, 900, 65535]))
            boxes = plausible_boxes(cnt, h, w) if rng.random() < 0.6 else hostile((cnt, 4), 500)
            if cnt and rng.random() < 0.3:
                idx = rng.integers(0, cnt, max(cnt // 10, 1))
                boxes[idx] = hostile((len(idx), 4), 1e4)
            scores = rng.uniform(0.6, 1.0, cnt) if rng.random() < 0.7 else hostile((cnt,), 1.0)
            B.text_lines(boxes, scores, (h, w), mode=str(rng.choice(["H", "O"])), device_id=-1, capacity=int(rng.choice([1, 4, 512])))
        elif k == 1:
            recs = hostile((int(rng.integers(0, 40)), 9), 1000)
            B.result_text(recs, float(rng.choice([1.0, 0.5, 1e-9, 0.0, -1.0, np.nan, np.inf])))
        elif k == 2:
            h, w = int(rng.choice([1, 2, 37, 300])), int(rng.choice([1, 3, 53, 450]))
            img = np.zeros((h, w, 3), np.uint8)
            B.draw_boxes(img, hostile((int(rng.integers(0, 20)), 9), float(rng.choice([10, 300, 1e6]))))
        else:
            B.load_library()
            import ctypes as C
            oh, ow = C.c_int(0), C.c_int(0)
            B.load_library().ctpn_resize_dims(int(rng.integers(-5, 70000)), int(rng.integers(-5, 70000)), float(rng.choice(SPECIAL)), float(rng.choice(SPECIAL)), C.byref(oh), C.byref(ow))
    except B.CtpnError as e:
        errs += 1
        assert e.code in (-1, -3, -4, -6), e
    n += 1
print("host entry point calls", n, "error returns", errs)
