"""ORACLE (test infrastructure only -- never imported by the product path).

cv2.imread of a Huffman-coded JPEG file (sequential or progressive), restated in numpy. The reference reads its images with cv2.imread (ctpn/demo.py:59; training:
lib/roi_data_layer/minibatch.py:83), i.e. through a THIRD-PARTY decoder that is not part of /root/reference: OpenCV's bundled / system
libjpeg(-turbo). This container has no cv2; Pillow 12.2.0 links libjpeg-turbo 3.1.4.1 (libjpeg API level 6.2), the same decoder family
with the same defaults (JDCT_ISLOW, do_fancy_upsampling = TRUE), and is the pin: tests/test_jpeg.py checks this file against Pillow bit
for bit on every case and against tests/golden/jpeg_cases.npz (files + Pillow's decode, written by oracle/make_jpeg_golden.py), so
"parity with the reference's imread" is anchored on libjpeg-turbo's output of the same files.

The algorithm restated is the published libjpeg one (IJG libjpeg 6b, files named per function below; libjpeg-turbo's SIMD paths are
bit-exact with that C code for well-formed streams):
    jdmarker.c   marker parsing (SOI, DQT, DHT, SOF0/1/2, DRI, SOS)
    jdhuff.c     sequential Huffman entropy decoding, DC prediction, restart intervals
    jdphuff.c    progressive Huffman entropy decoding: DC / AC, first / refinement scans (spectral selection, successive approximation)
    jidctint.c   "islow" 8 x 8 inverse DCT: CONST_BITS 13, PASS1_BITS 2, columns then rows, + 128, clamp
    jdsample.c   h2v2_fancy_upsample: triangle filter 3/4 + 1/4 in both directions, + 8 / + 7 alternating rounding, edge replication;
                 h2v1_fancy_upsample (4:2:2): the same filter along the row only, + 1 / + 2;
                 h1v2_fancy_upsample (4:4:0; libjpeg-turbo): the same filter along the column only, + 1 / + 2
    EXIF         cv2.imread (OpenCV >= 3.1, default flags) turns the decoded image by the orientation tag 0x0112 of the APP1 segment
    jdcolor.c    YCbCr -> RGB in 16-bit fixed point (SCALEBITS 16)
The device side (text-detection-ctpn_amd/csrc/jpeg.hip) computes the last three as HIP kernels and the first two on the host pool; the
entropy half here is a pure-Python loop and is meant for small images only (the tests run the big ones through the library's host half
and this file's vectorised pixel half).
"""
import struct

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42,
                   49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class Unsupported(ValueError):
    pass


def exif_orientation(seg):
    """Orientation (tag 0x0112 of IFD0, a SHORT) of an APP1 segment body, 1 if there is none."""
    if len(seg) < 14 or seg[:6] != b"Exif\0\0":
        return 1
    t = seg[6:]
    if t[:2] not in (b"II", b"MM"):
        return 1
    e = "<" if t[:2] == b"II" else ">"
    if struct.unpack(e + "H", t[2:4])[0] != 42:
        return 1
    (ifd,) = struct.unpack(e + "I", t[4:8])
    if ifd + 2 > len(t):
        return 1
    (n,) = struct.unpack(e + "H", t[ifd:ifd + 2])
    for k in range(n):
        ent = t[ifd + 2 + 12 * k: ifd + 14 + 12 * k]
        if len(ent) < 12:
            return 1
        tag, typ = struct.unpack(e + "HH", ent[:4])
        if tag == 0x0112:
            (v,) = struct.unpack(e + "H", ent[8:10])
            return v if typ == 3 and 1 <= v <= 8 else 1
    return 1


def apply_orientation(img, o):
    """The stored image -> what cv2.imread returns for EXIF orientation o (2 mirrored left-right, 3 turned by 180 degrees, 4 mirrored
    top-bottom, 5 transposed, 6 a quarter turn clockwise, 7 transverse, 8 a quarter turn anti-clockwise)."""
    if o == 2:
        return img[:, ::-1]
    if o == 3:
        return img[::-1, ::-1]
    if o == 4:
        return img[::-1]
    if o == 5:
        return img.swapaxes(0, 1)
    if o == 6:
        return img.swapaxes(0, 1)[:, ::-1]
    if o == 7:
        return img.swapaxes(0, 1)[::-1, ::-1]
    if o == 8:
        return img.swapaxes(0, 1)[::-1]
    return img


# ---------------------------------------------------------------------------------------------------------------- jdmarker.c
def parse(data, resume=None):
    """-> dict(h, w, comps=[(id, hs, vs, tq)], qt={tq: (64,) natural order}, huff={(class, id): (counts, values)}, scan=[(id, td, ta)],
    band=(Ss, Se, Ah, Al), progressive, dri, pos = offset of the entropy-coded segment). resume = (a dict this function returned, offset):
    carry on behind a scan with the tables so far (progressive files); None when the file ends (EOI or no further marker)."""
    if resume is None:
        if data[:2] != b"\xff\xd8":
            raise ValueError("no SOI")
        i, qt, huff, frame, dri, prog, marks = 2, {}, {}, None, 0, False, {}
    else:
        f0, i = resume
        qt, huff, frame, dri, prog, marks = dict(f0["qt"]), dict(f0["huff"]), (f0["h"], f0["w"], f0["comps"]), f0["dri"], f0["progressive"], f0["marks"]
        while i + 1 < len(data) and not (data[i] == 0xFF and data[i + 1] not in (0x00, 0xFF) and not 0xD0 <= data[i + 1] <= 0xD7):
            i += 1                                        # what is left of the previous scan's bytes
        if i + 1 >= len(data):
            return None
    while True:
        if data[i] != 0xFF:
            raise ValueError("marker expected at %d" % i)
        m = data[i + 1]
        i += 2
        if m == 0xFF:
            i -= 1
            continue
        if m == 0xD8 or 0xD0 <= m <= 0xD7 or m == 0x01:
            continue
        if m == 0xD9:
            if resume is not None:
                return None
            raise ValueError("EOI before SOS")
        (L,) = struct.unpack(">H", data[i:i + 2])
        seg = data[i + 2:i + L]
        i += L
        if m == 0xDB:
            j = 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                j += 1
                if pq == 0:
                    tbl = list(seg[j:j + 64])
                    j += 64
                else:
                    tbl = list(struct.unpack(">64H", seg[j:j + 128]))
                    j += 128
                nat = np.zeros(64, np.int64)
                nat[ZIGZAG] = tbl
                qt[tq] = nat
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                counts = list(seg[j + 1:j + 17])
                n = sum(counts)
                huff[(tc, th)] = (counts, list(seg[j + 17:j + 17 + n]))
                j += 17 + n
        elif m in (0xC0, 0xC1, 0xC2):
            prog = m == 0xC2
            if seg[0] != 8:
                raise Unsupported("sample precision")
            h, w, nc = struct.unpack(">H", seg[1:3])[0], struct.unpack(">H", seg[3:5])[0], seg[5]
            frame = (h, w, [(seg[6 + 3 * k], seg[7 + 3 * k] >> 4, seg[7 + 3 * k] & 15, seg[8 + 3 * k]) for k in range(nc)])
        elif 0xC3 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported("lossless / hierarchical / arithmetic")
        elif m == 0xDD:
            (dri,) = struct.unpack(">H", seg[:2])
        elif m == 0xE0 and seg[:5] == b"JFIF\0":
            marks["jfif"] = True
        elif m == 0xEE and seg[:5] == b"Adobe" and len(seg) >= 12:
            marks["adobe"] = seg[11]
        elif m == 0xE1 and "exif" not in marks and seg[:6] == b"Exif\0\0":
            # the FIRST EXIF segment decides (one TIFF header is read, by OpenCV's ExifReader and by Pillow's getexif() alike)
            marks["exif"] = True
            o = exif_orientation(seg)
            if o != 1:
                marks["orientation"] = o
        elif m == 0xDA:
            ns = seg[0]
            scan = [(seg[1 + 2 * k], seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15) for k in range(ns)]
            band = (seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns] >> 4, seg[3 + 2 * ns] & 15)
            h, w, comps = frame
            return dict(h=h, w=w, comps=comps, qt=qt, huff=huff, scan=scan, band=band, progressive=prog, dri=dri, pos=i, marks=marks)


# ---------------------------------------------------------------------------------------------------------------- jdhuff.c
class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        while self.n <= 24:
            b = self.d[self.p] if self.p < len(self.d) else 0
            if b == 0xFF:
                nb = self.d[self.p + 1] if self.p + 1 < len(self.d) else 0xD9
                if nb == 0:
                    self.p += 2                       # stuffed 0xFF
                else:
                    b = 0                             # a marker: the decoder feeds zero bits (jdhuff.c jpeg_fill_bit_buffer)
            else:
                self.p += 1
            self.acc = (self.acc << 8) | b
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        v = (self.acc >> (self.n - k)) & ((1 << k) - 1)
        self.n -= k
        return v

    def restart(self):
        self.acc = self.n = 0
        while not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _code_table(counts, vals):
    codes, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            codes[(length, code)] = vals[k]
            k += 1
            code += 1
        code <<= 1
    return codes


def _decode(bits, codes):
    code = 0
    for length in range(1, 17):
        code = (code << 1) | bits.get(1)
        if (length, code) in codes:
            return codes[(length, code)]
    raise ValueError("bad Huffman code")


def _extend(v, t):
    """jdhuff.c HUFF_EXTEND"""
    if t == 0:
        return 0
    return v if v >= (1 << (t - 1)) else v - (1 << t) + 1


def coefficients(data):
    """File bytes -> (frame dict, [one (block rows, block columns, 64) int32 array per component]): quantised coefficients, natural order.
    Pure Python: small images only."""
    f = parse(data)
    if f["progressive"]:
        return _coefficients_progressive(data, f)
    comps = f["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    if len(comps) == 1:
        comps = [(comps[0][0], 1, 1, comps[0][3])]         # a single-component scan is non-interleaved: one block per MCU
        hmax = vmax = 1
    mw, mh = -(-f["w"] // (8 * hmax)), -(-f["h"] // (8 * vmax))
    tabs = {k: _code_table(*v) for k, v in f["huff"].items()}
    blocks = [np.zeros((mh * c[2], mw * c[1], 64), np.int32) for c in comps]
    sc = {cid: (td, ta) for cid, td, ta in f["scan"]}
    bits, pred, n = _Bits(data, f["pos"]), [0] * len(comps), 0
    for my in range(mh):
        for mx in range(mw):
            if f["dri"] and n and n % f["dri"] == 0:
                bits.restart()
                pred = [0] * len(comps)
            n += 1
            for ci, (cid, ch, cv, _) in enumerate(comps):
                td, ta = sc[cid]
                for by in range(cv):
                    for bx in range(ch):
                        blk = blocks[ci][my * cv + by, mx * ch + bx]
                        t = _decode(bits, tabs[(0, td)])
                        pred[ci] += _extend(bits.get(t), t)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = _decode(bits, tabs[(1, ta)])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                break
                            k += r
                            blk[ZIGZAG[k]] = _extend(bits.get(s), s)
                            k += 1
    f = dict(f, comps=comps)
    return f, blocks


# ---------------------------------------------------------------------------------------------------------------- jdphuff.c
def _refine(bits, blk, pos, p1, m1):
    """one correction bit for an already nonzero coefficient: its magnitude grows by the scan's bit, away from zero"""
    if bits.get(1) and (blk[pos] & p1) == 0:
        blk[pos] += p1 if blk[pos] >= 0 else m1


def _prog_block(bits, blk, tabs, td, ta, band, st):
    """One block of one progressive scan (decode_mcu_DC_first / _DC_refine / _AC_first / _AC_refine). st = [DC predictor, EOBRUN]."""
    ss, se, ah, al = band
    p1, m1 = 1 << al, -(1 << al)
    if ss == 0:
        if ah == 0:
            t = _decode(bits, tabs[(0, td)])
            st[0] += _extend(bits.get(t), t)
            blk[0] = st[0] * p1
        elif bits.get(1):
            blk[0] |= p1
        return
    if ah == 0:
        if st[1] > 0:
            st[1] -= 1
            return
        k = ss
        while k <= se:
            rs = _decode(bits, tabs[(1, ta)])
            r, s = rs >> 4, rs & 15
            if s:
                k += r
                blk[ZIGZAG[k]] = _extend(bits.get(s), s) * p1
            elif r == 15:
                k += 15
            else:
                st[1] = (1 << r) + (bits.get(r) if r else 0) - 1
                break
            k += 1
        return
    k = ss
    if st[1] == 0:
        while k <= se:
            rs = _decode(bits, tabs[(1, ta)])
            r, s = rs >> 4, rs & 15
            if s:
                s = p1 if bits.get(1) else m1
            elif r != 15:
                st[1] = (1 << r) + (bits.get(r) if r else 0)
                break
            while k <= se:
                pos = ZIGZAG[k]
                if blk[pos] != 0:
                    _refine(bits, blk, pos, p1, m1)
                else:
                    r -= 1
                    if r < 0:
                        break
                k += 1
            if s:
                blk[ZIGZAG[k]] = s
            k += 1
    if st[1] > 0:
        while k <= se:
            pos = ZIGZAG[k]
            if blk[pos] != 0:
                _refine(bits, blk, pos, p1, m1)
            k += 1
        st[1] -= 1


def _coefficients_progressive(data, f):
    comps = f["comps"]
    if len(comps) == 1:
        comps = [(comps[0][0], 1, 1, comps[0][3])]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mw, mh = -(-f["w"] // (8 * hmax)), -(-f["h"] // (8 * vmax))
    blocks = [np.zeros((mh * c[2], mw * c[1], 64), np.int32) for c in comps]
    index = {c[0]: i for i, c in enumerate(comps)}
    last = f
    while f is not None:
        last = f
        tabs = {k: _code_table(*v) for k, v in f["huff"].items()}
        bits, scan, band = _Bits(data, f["pos"]), f["scan"], f["band"]
        st = [[0, 0] for _ in scan]
        if len(scan) > 1:                                 # interleaved (DC scans only): MCU by MCU, like a sequential scan
            units = [(my, mx) for my in range(mh) for mx in range(mw)]
        else:                                             # one component: its own blocks in raster order, without the MCU padding
            ci = index[scan[0][0]]
            cw = -(-(-(-f["w"] * comps[ci][1] // hmax)) // 8)
            ch = -(-(-(-f["h"] * comps[ci][2] // vmax)) // 8)
            units = [(by, bx) for by in range(ch) for bx in range(cw)]
        for n, (uy, ux) in enumerate(units):
            if f["dri"] and n and n % f["dri"] == 0:
                bits.restart()
                st = [[0, 0] for _ in scan]
            if len(scan) > 1:
                for k, (cid, td, ta) in enumerate(scan):
                    ci = index[cid]
                    for by in range(comps[ci][2]):
                        for bx in range(comps[ci][1]):
                            _prog_block(bits, blocks[ci][uy * comps[ci][2] + by, ux * comps[ci][1] + bx], tabs, td, ta, band, st[k])
            else:
                cid, td, ta = scan[0]
                _prog_block(bits, blocks[index[cid]][uy, ux], tabs, td, ta, band, st[0])
        f = parse(data, resume=(f, bits.p))
    return dict(last, comps=comps), blocks


# ---------------------------------------------------------------------------------------------------------------- jidctint.c
_F = dict(F298=2446, F390=3196, F541=4433, F765=6270, F899=7373, F1175=9633, F1501=12299, F1847=15137, F1961=16069, F2053=16819, F2562=20995,
          F3072=25172)                                     # FIX(0.298631336) ... FIX(3.072711026) at CONST_BITS = 13


def _idct_1d(x, descale):
    F = _F
    z2, z3 = x[..., 2], x[..., 6]
    z1 = (z2 + z3) * F["F541"]
    tmp2, tmp3 = z1 - z3 * F["F1847"], z1 + z2 * F["F765"]
    z2, z3 = x[..., 0], x[..., 4]
    tmp0, tmp1 = (z2 + z3) << 13, (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F["F1175"]
    t0, t1, t2, t3 = t0 * F["F298"], t1 * F["F2053"], t2 * F["F3072"], t3 * F["F1501"]
    z1, z2, z3, z4 = -z1 * F["F899"], -z2 * F["F2562"], z5 - z3 * F["F1961"], z5 - z4 * F["F390"]
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    r = 1 << (descale - 1)
    return np.stack([(tmp10 + t3 + r) >> descale, (tmp11 + t2 + r) >> descale, (tmp12 + t1 + r) >> descale, (tmp13 + t0 + r) >> descale,
                     (tmp13 - t0 + r) >> descale, (tmp12 - t1 + r) >> descale, (tmp11 - t2 + r) >> descale, (tmp10 - t3 + r) >> descale], -1)


def idct_islow(coef):
    """(..., 64) dequantised coefficients, natural order -> (..., 8, 8) uint8 samples. jpeg_idct_islow: pass 1 over columns (descale by
    CONST_BITS - PASS1_BITS), pass 2 over rows (descale by CONST_BITS + PASS1_BITS + 3), + 128, range limit."""
    c = np.asarray(coef).reshape(np.shape(coef)[:-1] + (8, 8)).astype(np.int64)
    ws = np.swapaxes(_idct_1d(np.swapaxes(c, -1, -2), 13 - 2), -1, -2)
    return np.clip(_idct_1d(ws, 13 + 2 + 3) + 128, 0, 255).astype(np.uint8)


def component_planes(blocks, qts):
    """[(bh, bw, 64) quantised blocks], [(64,) quantisation table per component] -> [(8 bh, 8 bw) uint8 plane per component]."""
    out = []
    for b, q in zip(blocks, qts):
        px = idct_islow(np.asarray(b).astype(np.int64) * np.asarray(q).astype(np.int64))
        bh, bw = px.shape[:2]
        out.append(px.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    return out


# ---------------------------------------------------------------------------------------------------------------- jdsample.c
def upsample_h2v2_fancy(plane, dw, dh):
    """Chroma plane (padded to whole blocks), its real size dw x dh -> (2 dh, 2 dw) int64. h2v2_fancy_upsample: per output row the nearer
    input row weighs 3, the further one 1 (the row above the first / below the last real row is that row again); per output column the
    same 3 : 1 on those column sums, (.. + 8) >> 4 for even and (.. + 7) >> 4 for odd output columns; the first and the last column take
    (4 colsum + 8) >> 4 and (4 colsum + 7) >> 4. (Used for dw > 2 only: see pixels_from_coefficients.)"""
    c = np.asarray(plane)[:dh, :dw].astype(np.int64)
    up = np.vstack([c[:1], c, c[-1:]])
    out = np.zeros((2 * dh, 2 * dw), np.int64)
    for v in range(2):
        colsum = 3 * up[1:-1] + (up[:-2] if v == 0 else up[2:])
        last = np.concatenate([colsum[:, :1], colsum[:, :-1]], 1)
        nxt = np.concatenate([colsum[:, 1:], colsum[:, -1:]], 1)
        even, odd = (colsum * 3 + last + 8) >> 4, (colsum * 3 + nxt + 7) >> 4
        even[:, 0] = (colsum[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (colsum[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def upsample_h2v1_fancy(plane, dw, h):
    """4:2:2. Chroma plane (padded to whole blocks), its real size dw x h -> (h, 2 dw) int64. h2v1_fancy_upsample: within each row the
    nearer sample weighs 3, the further one 1, (.. + 1) >> 2 for even and (.. + 2) >> 2 for odd output columns; the first and the last
    output column are the edge samples themselves. (Used for dw > 2 only, like the h2v2 filter.)"""
    c = np.asarray(plane)[:h, :dw].astype(np.int64)
    left = np.concatenate([c[:, :1], c[:, :-1]], 1)
    right = np.concatenate([c[:, 1:], c[:, -1:]], 1)
    even, odd = (3 * c + left + 1) >> 2, (3 * c + right + 2) >> 2
    even[:, 0] = c[:, 0]
    odd[:, -1] = c[:, -1]
    out = np.zeros((h, 2 * dw), np.int64)
    out[:, 0::2] = even
    out[:, 1::2] = odd
    return out


def upsample_h1v2_fancy(plane, w, dh):
    """4:4:0. Chroma plane (padded to whole blocks), its real size w x dh -> (2 dh, w) int64. libjpeg-turbo's h1v2_fancy_upsample: within
    each column the nearer row weighs 3, the further one 1 (the row above the first / below the last real row is that row again),
    (.. + 1) >> 2 for the upper and (.. + 2) >> 2 for the lower output row of a pair. No narrow-image exception."""
    c = np.asarray(plane)[:dh, :w].astype(np.int64)
    up = np.vstack([c[:1], c, c[-1:]])
    out = np.zeros((2 * dh, w), np.int64)
    out[0::2] = (3 * up[1:-1] + up[:-2] + 1) >> 2
    out[1::2] = (3 * up[1:-1] + up[2:] + 2) >> 2
    return out


# ---------------------------------------------------------------------------------------------------------------- jdcolor.c
def _fix(x):
    return int(x * 65536 + 0.5)


def ycc_to_bgr(y, cb, cr):
    """build_ycc_rgb_table / ycc_rgb_convert, output in cv2's channel order."""
    y, xb, xr = np.asarray(y).astype(np.int64), np.asarray(cb).astype(np.int64) - 128, np.asarray(cr).astype(np.int64) - 128
    r = y + ((_fix(1.40200) * xr + 32768) >> 16)
    b = y + ((_fix(1.77200) * xb + 32768) >> 16)
    g = y + ((-_fix(0.34414) * xb + 32768 - _fix(0.71414) * xr) >> 16)
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------- whole pipeline
def pixels_from_coefficients(blocks, qts, h, w, hs, vs=None):
    """The device half: quantised blocks per component + tables -> (h, w, 3) BGR uint8. hs, vs = luma sampling factors: 1 x 1 (4:4:4),
    2 x 2 (4:2:0; vs defaults to hs), 2 x 1 (4:2:2), 1 x 2 (4:4:0)."""
    vs = hs if vs is None else vs
    pl = component_planes(blocks, qts)
    if len(pl) == 1:
        y = pl[0][:h, :w]
        return np.stack([y, y, y], -1)
    if hs == 1 and vs == 2:
        dh = (h + 1) // 2
        cb, cr = upsample_h1v2_fancy(pl[1], w, dh), upsample_h1v2_fancy(pl[2], w, dh)
    elif hs == 2 and vs == 1:
        dw = (w + 1) // 2
        if dw > 2:
            cb, cr = upsample_h2v1_fancy(pl[1], dw, h), upsample_h2v1_fancy(pl[2], dw, h)
        else:
            cb, cr = (np.repeat(p[:h, :dw], 2, 1).astype(np.int64) for p in pl[1:3])
    elif hs == 2:
        dw, dh = (w + 1) // 2, (h + 1) // 2
        if dw > 2:
            cb, cr = upsample_h2v2_fancy(pl[1], dw, dh), upsample_h2v2_fancy(pl[2], dw, dh)
        else:      # jdsample.c jinit_upsampler: the fancy filter needs downsampled_width > 2; narrower images get h2v2_upsample (replication)
            cb, cr = (np.repeat(np.repeat(p[:dh, :dw], 2, 0), 2, 1).astype(np.int64) for p in pl[1:3])
    else:
        cb, cr = pl[1], pl[2]
    return ycc_to_bgr(pl[0][:h, :w], cb[:h, :w], cr[:h, :w])


def imread_bgr(data):
    """cv2.imread(file, IMREAD_COLOR) of a sequential or progressive JPEG given as bytes."""
    f, blocks = coefficients(data)
    comps = f["comps"]
    if len(comps) == 3:
        # jdapimin.c default_decompress_parms: JFIF -> YCbCr; else Adobe transform 0 -> RGB, 1 -> YCbCr; else ids 'R' 'G' 'B' -> RGB
        marks = f["marks"]
        rgb = False if marks.get("jfif") else (marks["adobe"] == 0 if "adobe" in marks else [c[0] for c in comps] == [82, 71, 66])
        if rgb:
            raise Unsupported("RGB-coded file")
        if not (comps[1][1:3] == (1, 1) and comps[2][1:3] == (1, 1) and comps[0][1:3] in ((1, 1), (2, 2), (2, 1), (1, 2))):
            raise Unsupported("sampling factors")
    elif len(comps) != 1:
        raise Unsupported("component count")
    img = pixels_from_coefficients(blocks, [f["qt"][c[3]] for c in comps], f["h"], f["w"], comps[0][1], comps[0][2])
    return np.ascontiguousarray(apply_orientation(img, f["marks"].get("orientation", 1)))
