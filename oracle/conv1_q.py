"""ORACLE (test infrastructure only -- never imported by the product path).

Arithmetic specification of how the 16-bit modes compute conv1_1 from the uint8 feed since round 4: the q-image form
(text-detection-ctpn_amd/csrc/layers.hip: image_to_q_kernel, pack_conv1_frags, conv_first_p_kernel; csrc/conv3x3_impl.h: the producer
inside conv3x3_wr_kernel<FUSE>). Restated here in numpy so that the claim "equal to the reference op up to the 16-bit rounding of the 27
weights and of the output" is checked on the CPU, layer for layer, borders included.

Reference op being restated: _get_image_blob's mean subtraction (lib/fast_rcnn/test.py:7-11, PIXEL_MEANS lib/fast_rcnn/config.py:200)
followed by conv1_1 = relu(bias_add(conv2d(x, W[3,3,3,64], stride 1, 'SAME'))) (lib/networks/network.py:160-183, VGGnet_test.py:20-22).

The device form, per output pixel (y, x) and channel co, with m = round(mean) = (103, 116, 123) and d_c = m_c - mean_c:

    q-image       8-byte pixels (q_B, q_G, q_R, P): q_c = p_c - m_c (an integer, exact in bf16 and fp16), P = 1.0, at image pixels;
                  ALL-ZERO pixels everywhere else (a frame of two pixels and more): TF's SAME padding, and through P the indicator
                  "this tap lies inside the image"
    K slots       three MFMA steps ky = 0, 1, 2 of 16 slots each; data = q-image pixels (x - 1), (x), (x) again, (x + 1) of row y - 1 + ky:
                     slot 0..2   w16[ky][0][c]     3   G16[ky][0]                       4..6   w16[ky][1][c]     7   ky == 1 ? V_hi : G16[ky][1]
                     slot 8..10  0                 11  ky == 1 ? V_lo : 0                12..14 w16[ky][2][c]     15  G16[ky][2]
                  w16 = the weights rounded to the 16-bit type, G[ky][kx] = sum_c w16[ky][kx][c] d_c, G16 = G rounded to the type,
                  V = bias + G[1][1] + sum over the other eight taps of (G - G16), as a (hi, lo) pair of the type
    sum           fp32 accumulate over the 48 slots (MFMA), ReLU, rounded to the type

so that an interior pixel computes sum w16 (p - mean) + bias up to ~2^-17 |V|, and a border pixel -- whose missing taps contribute
neither their w16 q nor, P being 0 there, their G16 -- additionally misses the (G - G16) of its missing taps (< 2^-9 |G| each in bf16).
"""
import numpy as np

from oracle.rounding import bf16_round, fp16_round

MEANS = np.array([102.9801, 115.9465, 122.7717], np.float64)     # BGR, lib/fast_rcnn/config.py:200
M_INT = np.array([103.0, 116.0, 123.0], np.float64)


def _rnd(dtype):
    return bf16_round if dtype == "bf16" else fp16_round


def q_image(images_u8):
    """(n, h, w, 3) uint8 -> (n, h + 4, w + 4, 4) float32: the q-image with a two-pixel zero frame (the device's frame is wider; only two
    pixels of it are ever multiplied by a non-zero weight)."""
    im = np.asarray(images_u8)
    n, h, w, _ = im.shape
    q = np.zeros((n, h + 4, w + 4, 4), np.float32)
    q[:, 2:-2, 2:-2, :3] = (im.astype(np.float64) - M_INT).astype(np.float32)
    q[:, 2:-2, 2:-2, 3] = 1.0
    return q


def pack_slots(w_hwio, bias, dtype="bf16"):
    """(3, 3, 3, 64) fp32 weights, (64,) bias -> slot weights (3 ky, 16 slots, 64 co) as float32 holding 16-bit-representable values."""
    rnd = _rnd(dtype)
    w16 = rnd(np.asarray(w_hwio, np.float32)).astype(np.float64)                       # [ky][kx][c][co]
    G = (w16 * (M_INT - MEANS)[None, None, :, None]).sum(2)                             # [ky][kx][co]
    G16 = rnd(G.astype(np.float32)).astype(np.float64)                                 # [ky][kx][co]
    V = np.asarray(bias, np.float64) + G[1, 1] + (G - G16).sum((0, 1)) - (G - G16)[1, 1]
    V_hi = rnd(V.astype(np.float32)).astype(np.float64)
    V_lo = rnd((V - V_hi).astype(np.float32)).astype(np.float64)
    s = np.zeros((3, 16, 64), np.float64)
    for ky in range(3):
        s[ky, 0:3] = w16[ky, 0]
        s[ky, 3] = G16[ky, 0]
        s[ky, 4:7] = w16[ky, 1]
        s[ky, 7] = V_hi if ky == 1 else G16[ky, 1]
        s[ky, 11] = V_lo if ky == 1 else 0.0
        s[ky, 12:15] = w16[ky, 2]
        s[ky, 15] = G16[ky, 2]
    return s.astype(np.float32)


def conv1_1_from_q(images_u8, w_hwio, bias, dtype="bf16", round_output=True):
    """conv1_1 of the 16-bit modes' uint8 feed as the device computes it (float64 accumulation stands in for the MFMA's fp32: the products
    are exact in either, the sums differ by fp32 rounding of O(1e-7) relative). Returns (n, h, w, 64) float32."""
    q = q_image(images_u8).astype(np.float64)
    s = pack_slots(w_hwio, bias, dtype).astype(np.float64)
    n, hq, wq, _ = q.shape
    h, w = hq - 4, wq - 4
    out = np.zeros((n, h, w, 64), np.float64)
    for ky in range(3):
        rows = q[:, 1 + ky: 1 + ky + h]                                                 # q rows of image rows y - 1 + ky
        left, mid, right = rows[:, :, 1: 1 + w], rows[:, :, 2: 2 + w], rows[:, :, 3: 3 + w]      # q pixels x - 1, x, x + 1
        data = np.concatenate([left, mid, mid, right], axis=-1)                          # 16 slots per pixel
        out += data @ s[ky]
    out = np.maximum(out, 0.0).astype(np.float32)
    return _rnd(dtype)(out) if round_output else out


def reference_on_rounded_weights(images_u8, w_hwio, bias, dtype="bf16", round_output=True):
    """The reference op evaluated with the weights rounded to the 16-bit type (float64): what the q-image form must reproduce."""
    rnd = _rnd(dtype)
    w16 = rnd(np.asarray(w_hwio, np.float32)).astype(np.float64)
    x = np.asarray(images_u8).astype(np.float64) - MEANS
    n, h, w, _ = x.shape
    xp = np.zeros((n, h + 2, w + 2, 3), np.float64)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((n, h, w, 64), np.float64) + np.asarray(bias, np.float64)
    for ky in range(3):
        for kx in range(3):
            out += xp[:, ky: ky + h, kx: kx + w] @ w16[ky, kx]
    out = np.maximum(out, 0.0).astype(np.float32)
    return rnd(out) if round_output else out
