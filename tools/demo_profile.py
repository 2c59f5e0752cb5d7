#!/usr/bin/env python
"""Where does `python ctpn/demo.py` spend a call? The reference's own data/demo files (tests/golden/demo_files.npz) through ctpn.demo.ctpn(),
seeded weights, the shipped config; cProfile over the five images after one untimed pass.   usage: python tools/demo_profile.py"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctpn_amd  # noqa: E402
from ctpn_amd.ctpn import demo  # noqa: E402
from ctpn_amd.lib.fast_rcnn.config import cfg  # noqa: E402
from ctpn_amd.lib.networks.factory import get_network  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "demo_files.npz"))
tmp = tempfile.mkdtemp(prefix="ctpn_demo_")
os.makedirs(os.path.join(tmp, "data", "demo"))
os.makedirs(os.path.join(tmp, "data", "results"))
names = []
for nm in g["names"]:
    p = os.path.join(tmp, "data", "demo", str(nm))
    open(p, "wb").write(g["file_" + str(nm).replace(".", "_")].tobytes())
    names.append(p)
os.chdir(tmp)
print("precision", cfg.TEST.PRECISION)
net = get_network("VGGnet_test")
net.restore_synthetic(0)
for p in names:
    demo.ctpn(None, net, p)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for p in names:
    demo.ctpn(None, net, p)
pr.disable()
print("per image %.1f ms" % ((time.time() - t0) / len(names) * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:4000])
