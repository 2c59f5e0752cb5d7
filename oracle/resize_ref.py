"""TEST INFRASTRUCTURE ONLY (oracle): cv2.resize(..., fx, fy, interpolation=cv2.INTER_LINEAR) restated in numpy.

Follows OpenCV 3.4's modules/imgproc/src/resize.cpp (the reference pins opencv_python==3.4.0.12, requirements.txt; call
sites ctpn/demo.py:25,51 and lib/fast_rcnn/test.py:23). cv2 is NOT installed in this image and OpenCV's source is not part
of the reference tree: PARITY UNPINNED against the real library -- the restatement is from the published algorithm:

    dsize = cvRound(src * f)                                   (round half to even)
    fx = float32((dx + 0.5) / f - 0.5); sx = floor(fx); fx -= sx
    columns: sx < 0 -> sx = 0, fx = 0;  sx >= w - 1 -> sx = w - 1, fx = 0      rows: source row index clamped to [0, h - 1]
    uint8  : a = cvRound(weight * 2048) as int16;  S = s[sx] * a0 + s[sx + 1] * a1  (int32)
             dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    float32: S = s[sx] * a0 + s[sx + 1] * a1;  dst = S0 * b0 + S1 * b1           (fp32, no fused multiply-add)
The HIP kernel (csrc/preprocess.hip) must match this file bit for bit (tests/test_gpu_parity.py).
"""
import numpy as np


def out_dim(src, f):
    return int(np.rint(np.float64(src) * np.float64(f)))      # np.rint rounds half to even, like cvRound


def _coords(nd, f, ns, clamp):
    d = np.arange(nd, dtype=np.float64)
    fx = ((d + 0.5) * (1.0 / np.float64(f)) - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    fx = (fx - s.astype(np.float32)).astype(np.float32)
    if clamp:
        lo = s < 0
        s[lo] = 0
        fx[lo] = 0
        hi = s >= ns - 1
        s[hi] = ns - 1
        fx[hi] = 0
    return s, fx


def _short(v):
    return np.clip(np.rint(v.astype(np.float32)), -32768, 32767).astype(np.int64)


def resize_linear(im, fx, fy):
    """im: (h,w,3) or (n,h,w,3), uint8 or float32."""
    a = np.asarray(im)
    single = a.ndim == 3
    if single:
        a = a[None]
    n, h, w, _ = a.shape
    dh, dw = out_dim(h, fy), out_dim(w, fx)
    sx, wx = _coords(dw, fx, w, True)
    sy, wy = _coords(dh, fy, h, False)
    x1 = np.minimum(sx + 1, w - 1)
    y0 = np.clip(sy, 0, h - 1)
    y1 = np.clip(sy + 1, 0, h - 1)
    if a.dtype == np.uint8:
        a0, a1 = _short((np.float32(1) - wx) * np.float32(2048)), _short(wx * np.float32(2048))
        b0, b1 = _short((np.float32(1) - wy) * np.float32(2048)), _short(wy * np.float32(2048))
        src = a.astype(np.int64)
        r0, r1 = src[:, y0], src[:, y1]
        S0 = r0[:, :, sx] * a0[None, None, :, None] + r0[:, :, x1] * a1[None, None, :, None]
        S1 = r1[:, :, sx] * a0[None, None, :, None] + r1[:, :, x1] * a1[None, None, :, None]
        v = (((b0[None, :, None, None] * (S0 >> 4)) >> 16) + ((b1[None, :, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
        out = np.clip(v, 0, 255).astype(np.uint8)
    else:
        src = a.astype(np.float32)
        a0, a1 = (np.float32(1) - wx).astype(np.float32), wx
        b0, b1 = (np.float32(1) - wy).astype(np.float32), wy
        r0, r1 = src[:, y0], src[:, y1]
        S0 = (r0[:, :, sx] * a0[None, None, :, None]).astype(np.float32) + (r0[:, :, x1] * a1[None, None, :, None]).astype(np.float32)
        S1 = (r1[:, :, sx] * a0[None, None, :, None]).astype(np.float32) + (r1[:, :, x1] * a1[None, None, :, None]).astype(np.float32)
        out = ((S0 * b0[None, :, None, None]).astype(np.float32) + (S1 * b1[None, :, None, None]).astype(np.float32)).astype(np.float32)
    return out[0] if single else out
