"""ORACLE (test infrastructure only -- never imported by the product path).

Arithmetic specification of a device mode that does NOT exist yet ("what comes next", item 4): the 3 x 3 layers of the
bf16 throughput path computed with the Winograd minimal-filtering transform instead of nine direct taps. Written down -- and measured
end to end by tests/winograd_budget.py -- before the kernel, in the order this repo builds things: oracle, then kernel.

Reference op being restated: lib/networks/network.py:160-183 (relu(bias_add(conv2d(x, W[3,3,Ci,Co], stride 1, 'SAME')))). In exact
arithmetic both forms below EQUAL that op (tests/test_oracle.py checks it with the rounding switched off); what they change is where
bf16 rounding happens:

    direct bf16 path (the device today)    operands: bf16 activations x bf16 weights; fp32 accumulate; output rounded to bf16
    winograd_x  (1-D, F(2, 3) along x)     U[ky][f] = G w[ky][:]   from the fp32 weights, rounded ONCE to bf16 (offline)
                                           V[r][t][f] = B^T d[r][2t .. 2t+3]  sums of two bf16 values (exact in fp32), rounded to bf16:
                                                        the MFMA operand -- the rounding the direct path does not have
                                           m_f[y][t] = sum_ky sum_ci V[y+ky][t][f][ci] U[ky][f][ci][co]      fp32 accumulate
                                           out[y][2t] = m0 + m1 + m2, out[y][2t+1] = m1 - m2 - m3             fp32, + bias, ReLU
    winograd_2d (F(2 x 2, 3 x 3))          U = G w G^T, V = B^T d B on 4 x 4 tiles, 16 channel GEMMs, Y = A^T M A; same rounding points

Matrices: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (2015), F(2, 3).
"""
import numpy as np
import torch

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def bf16_round(x):
    """fp32 -> nearest-even bf16 -> fp32 (numpy): the rounding of v_cvt_pk_bf16_f32."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def fp16_round(x):
    """fp32 -> nearest-even IEEE fp16 -> fp32: the rounding of v_cvt_pk_f16_f32 / v_pk_add_f16 (round 4: the device mode CTPN_PREC_FP16W,
    csrc/conv3x3_wino.hip, computes V with packed fp16 adds and multiplies fp16 operands)."""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def fp16_round_ftz(x):
    """fp16_round with subnormal results flushed to zero (|x| < 2^-14): what an MFMA operand looks like if the matrix core flushes fp16
    denormals (probed on the device: tests/test_gpu_precision.py::test_winograd_kernel_matches_its_oracle reports which one matches)."""
    r = fp16_round(x)
    return np.where(np.abs(r) < 2.0 ** -14, np.float32(0), r).astype(np.float32)


ROUNDERS = {"bf16": bf16_round, "fp16": fp16_round, "fp16_ftz": fp16_round_ftz}


def _rnd(t, on, kind="bf16"):
    return torch.from_numpy(ROUNDERS[kind](t.numpy())) if on else t


def conv3x3_relu_winograd_x(x_nhwc, w_hwio, b, round_operands=True, relu=True, kind="bf16"):
    """(1, H, W, Ci) fp32 -> (1, H, W, Co) fp32 (the caller rounds the output like every 16-bit layer's). 'SAME' zero padding, stride 1.
    kind: the 16-bit type of the MFMA operands U and V ("bf16": round 3's reference kernel winograd.hip; "fp16": the product mode)."""
    x = np.asarray(x_nhwc, np.float32)
    assert x.ndim == 4 and x.shape[0] == 1
    _, H, W, Ci = x.shape
    Co = w_hwio.shape[3]
    tw = (W + 1) // 2
    xp = np.zeros((H + 2, 2 * tw + 2, Ci), np.float32)
    xp[1:H + 1, 1:W + 1] = x[0]
    d = torch.from_numpy(xp).unfold(1, 4, 2)                                      # (H+2, tw, Ci, 4)
    bt = torch.from_numpy(BT.astype(np.float32))
    V = _rnd(torch.einsum("ij,ytcj->ytic", bt, d).contiguous(), round_operands, kind)   # (H+2, tw, 4, Ci)
    U = np.einsum("ij,kjco->kico", G, np.asarray(w_hwio, np.float64)).astype(np.float32)   # (3 ky, 4 f, Ci, Co)
    U = torch.from_numpy(ROUNDERS[kind](U) if round_operands else U)
    M = torch.zeros((H, tw, 4, Co), dtype=torch.float32)
    for ky in range(3):
        for f in range(4):
            M[:, :, f, :] += (V[ky:ky + H, :, f, :].reshape(H * tw, Ci) @ U[ky, f]).reshape(H, tw, Co)
    at = torch.from_numpy(AT.astype(np.float32))
    y = torch.einsum("if,ytfc->ytic", at, M).reshape(H, 2 * tw, Co)[:, :W]
    y = y + torch.from_numpy(np.asarray(b, np.float32))
    if relu:
        y = torch.clamp(y, min=0)
    return y.unsqueeze(0).numpy()


def conv3x3_relu_winograd_2d(x_nhwc, w_hwio, b, round_operands=True, relu=True):
    x = np.asarray(x_nhwc, np.float32)
    assert x.ndim == 4 and x.shape[0] == 1
    _, H, W, Ci = x.shape
    Co = w_hwio.shape[3]
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = np.zeros((2 * th + 2, 2 * tw + 2, Ci), np.float32)
    xp[1:H + 1, 1:W + 1] = x[0]
    d = torch.from_numpy(xp).permute(2, 0, 1).unsqueeze(0).unfold(2, 4, 2).unfold(3, 4, 2)[0]   # (Ci, th, tw, 4, 4)
    bt = torch.from_numpy(BT.astype(np.float32))
    V = _rnd(torch.einsum("ij,cyxjk,lk->cyxil", bt, d, bt).contiguous(), round_operands)
    U = np.einsum("ij,jkco,lk->ilco", G, np.asarray(w_hwio, np.float64), G).astype(np.float32)    # (4, 4, Ci, Co)
    U = torch.from_numpy(bf16_round(U) if round_operands else U)
    M = torch.bmm(V.permute(3, 4, 1, 2, 0).reshape(16, th * tw, Ci), U.reshape(16, Ci, Co)).reshape(4, 4, th * tw, Co)
    at = torch.from_numpy(AT.astype(np.float32))
    Y = torch.einsum("ij,jktc,lk->tilc", at, M, at).reshape(th, tw, 2, 2, Co)
    y = Y.permute(0, 2, 1, 3, 4).reshape(2 * th, 2 * tw, Co)[:H, :W]
    y = y + torch.from_numpy(np.asarray(b, np.float32))
    if relu:
        y = torch.clamp(y, min=0)
    return y.unsqueeze(0).numpy()
