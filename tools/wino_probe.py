"""Probe: how far are the fp16 direct kernel and the fp16 Winograd kernel from their numpy restatements, in output ulps?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctpn_amd
from ctpn_amd import _binding as B
from oracle import network as N
from oracle import winograd as Wg

n, h, w, ci, co = 1, 16, 64, 128, 128
rng = np.random.default_rng(5)
x = Wg.fp16_round(np.maximum(rng.standard_normal((n, h, w, ci)).astype(np.float32), 0) * 2.0)
wt = (rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
b = (rng.standard_normal(co) * 0.1).astype(np.float32)

def ulps(a, ref):
    u = np.spacing(np.abs(ref).astype(np.float16)).astype(np.float32)
    return np.abs(a - ref) / u

direct_dev = B.debug_conv3x3(x, wt, b, precision="fp16", impl=1, fuse_pool=False)[0]
direct_ref64 = N.conv3x3_relu(x, Wg.fp16_round(wt), b)                  # fp16 operands, wide accumulate
direct_ref = Wg.fp16_round(direct_ref64)
u = ulps(direct_dev, direct_ref)
print("direct fp16 kernel vs numpy (fp16 operands): differ %.4f, mean %.3f ulp, max %.1f ulp" % ((u > 0).mean(), u.mean(), u.max()))
wino_dev = B.debug_conv3x3(x, wt, b, precision="fp16w", impl=1, fuse_pool=False)[0]
wino_ref32 = Wg.conv3x3_relu_winograd_x(x, wt, b, kind="fp16")
wino_ref = Wg.fp16_round(wino_ref32)
u = ulps(wino_dev, wino_ref)
print("winograd kernel vs oracle/winograd.py (fp16): differ %.4f, mean %.3f ulp, max %.1f ulp" % ((u > 0).mean(), u.mean(), u.max()))
e = np.abs(wino_dev - wino_ref32)
print("  |dev - oracle(fp32 out)| mean %.3e max %.3e; |oracle| mean %.3f" % (e.mean(), e.max(), np.abs(wino_ref32).mean()))
# by output column parity (out0 / out1) and by row
for par in (0, 1):
    uu = ulps(wino_dev[:, :, par::2], wino_ref[:, :, par::2])
    print("  columns of parity %d: differ %.4f" % (par, (uu > 0).mean()))
# oracle without V rounding: is the device closer to that?
ref_nov = Wg.fp16_round(conv := None) if False else None
import torch
def wino_variant(round_v, round_u):
    xx = np.asarray(x, np.float32); H, W = h, w
    tw = (W + 1) // 2
    xp = np.zeros((H + 2, 2 * tw + 2, ci), np.float32); xp[1:H + 1, 1:W + 1] = xx[0]
    d = torch.from_numpy(xp).unfold(1, 4, 2)
    bt = torch.from_numpy(Wg.BT.astype(np.float32))
    V = torch.einsum("ij,ytcj->ytic", bt, d).contiguous()
    if round_v: V = torch.from_numpy(Wg.fp16_round(V.numpy()))
    U = np.einsum("ij,kjco->kico", Wg.G, np.asarray(wt, np.float64)).astype(np.float32)
    if round_u: U = Wg.fp16_round(U)
    U = torch.from_numpy(U)
    M = torch.zeros((H, tw, 4, co), dtype=torch.float64)
    for ky in range(3):
        for f in range(4):
            M[:, :, f, :] += (V[ky:ky + H, :, f, :].reshape(H * tw, ci).double() @ U[ky, f].double()).reshape(H, tw, co)
    at = torch.from_numpy(Wg.AT)
    y = torch.einsum("if,ytfc->ytic", at, M).reshape(H, 2 * tw, co)[:, :W] + torch.from_numpy(b.astype(np.float64))
    return torch.clamp(y, min=0).unsqueeze(0).numpy().astype(np.float32)
for rv, ru in ((True, True), (False, True), (True, False), (False, False)):
    ref = wino_variant(rv, ru)
    e = np.abs(wino_dev - Wg.fp16_round(ref))
    print("  oracle variant round_V=%s round_U=%s (fp64 accumulate): differ %.4f, mean |dev - ref32| %.3e" % (rv, ru, (e > 0).mean(), np.abs(wino_dev - ref).mean()))
