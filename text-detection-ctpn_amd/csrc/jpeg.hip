// Baseline JPEG -> BGR uint8 on the device: what cv2.imread does for the reference (ctpn/demo.py:59), split where the work splits.
//   host   : marker parsing and Huffman entropy decoding (sequential by nature: one image per worker thread of the ctx's pool) into
//            quantised DCT coefficient blocks, int16, natural order -- about as many bytes as the decoded pixels;
//   device : dequantisation + the 8 x 8 inverse DCT (jpeg_idct_kernel), chroma upsampling + YCbCr -> BGR (jpeg_color_kernel).
// The pixel arithmetic is libjpeg's, integer for integer, so that the result equals what Pillow / cv2 (both libjpeg-turbo, whose SIMD paths
// are bit-exact with its C code) return: jidctint.c's "islow" IDCT (CONST_BITS 13, PASS1_BITS 2), jdsample.c's h2v2 "fancy" (triangle)
// upsampling with its alternating + 8 / + 7 rounding and edge replication (4:2:2: h2v1, + 1 / + 2), jdcolor.c's 16-bit fixed-point YCbCr -> RGB. Restated from the
// published algorithms (libjpeg 6b API level, which libjpeg-turbo implements); checked bit for bit against Pillow on the CPU through
// oracle/jpeg_ref.py and on the GPU through the C ABI (tests/test_jpeg.py, tests/test_gpu_jpeg.py).
// Supported: 8-bit Huffman-coded files, baseline / extended sequential (SOF0 / SOF1) and progressive (SOF2: spectral selection and
// successive approximation, jdphuff.c's four scan kinds -- a progressive file differs from a sequential one in its entropy coding only, so
// it is the host half's business alone: the coefficient blocks it hands the device are the same); 1 component, or 3 components YCbCr
// with luma sampling 1x1 (4:4:4), 2x2 (4:2:0), 2x1 (4:2:2) or 1x2 (4:4:0: jdsample.c's h1v2 filter) and 1x1 chroma; restart intervals; an
// EXIF orientation is applied like cv2.imread applies it (an index map in the colour kernel). Anything else (CMYK, 4:1:1, arithmetic coding,
// lossless, 12-bit, RGB-coded files) and every INCOMPLETE file (truncated entropy data, a progressive file without its last scans: libjpeg
// has its own rules for those) returns CTPN_ERR_UNSUPPORTED: the caller decodes that file on the host (lib/utils/image.py) -- a different
// decoder, not a silent fallback of this one.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"
#include "jpeg_pixel.h"      // jidct_1d, jpeg_pixel: the per-sample arithmetic of the two kernels below

namespace ctpn {

static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
                                    57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---------------------------------------------------------------------------------------------
// host: parsing + entropy decoding
// ---------------------------------------------------------------------------------------------
struct JHuff {
  uint8_t fast_len[512];     // code length for a 9-bit prefix (0 = longer than 9 bits)
  uint8_t fast_sym[512];
  int32_t maxcode[18];       // largest code of length l (left-aligned comparisons are avoided: canonical decode), -1 if none
  int32_t valptr[17];
  int32_t mincode[17];
  uint8_t vals[256];
  bool present = false;
};

static bool jhuff_build(JHuff& h, const uint8_t counts[16], const uint8_t* vals, int nvals) {
  std::memset(h.fast_len, 0, sizeof(h.fast_len));
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    for (int i = 0; i < counts[l - 1]; ++i, ++k, ++code) {
      if (k >= nvals || k >= 256 || code >= (1 << l)) return false;      // more codes of length l than l bits hold (a damaged DHT): the lookahead fill below would run past its table
      h.vals[k] = vals[k];
      if (l <= 9) {
        const int lo = code << (9 - l), n = 1 << (9 - l);
        for (int j = 0; j < n; ++j) { h.fast_len[lo + j] = (uint8_t)l; h.fast_sym[lo + j] = vals[k]; }
      }
    }
    h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
    if (code > (1 << l)) return false;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  h.present = true;
  return true;
}

struct JBits {
  const uint8_t* p; const uint8_t* end;
  uint64_t acc = 0; int n = 0;
  long long pad = 0;        // zero bits fed behind the end of the segment's data so far (they are the LAST bits fed: consumed iff n < pad)
  inline void fill() {      // keep at least 32 bits; 0xFF00 is a stuffed 0xFF, any other marker feeds zeros (the scan is over or a restart follows)
    while (n <= 56) {
      uint32_t b = 0;
      bool real = false;
      if (p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0) { p += 2; real = true; }
          else b = 0;
        } else { ++p; real = true; }
      }
      if (!real) pad += 8;
      acc = (acc << 8) | b;
      n += 8;
    }
  }
  // has the decoder consumed bits that are not in the file? (a truncated file, or entropy data that ends before its last MCU: libjpeg
  // warns, pretends the rest is zero bits and -- for the blocks it never reaches -- leaves zeros; Huffman codes decoded FROM the padding
  // are not that. Such a file is the host decoder's, which has libjpeg's premature-end behaviour)
  inline bool overran() const { return n < pad; }
  inline void restart_at(const uint8_t* q) { p = q; acc = 0; n = 0; pad = 0; }
  inline uint32_t peek(int k) { if (n < k) fill(); return (uint32_t)((acc >> (n - k)) & ((1u << k) - 1u)); }
  inline void skip(int k) { n -= k; }
  inline uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek(k); n -= k; return v; }
};

static inline int jdecode(JBits& b, const JHuff& h) {
  const uint32_t look = b.peek(16);
  const int fl = h.fast_len[look >> 7];
  if (fl) { b.skip(fl); return h.fast_sym[look >> 7]; }
  for (int l = 10; l <= 16; ++l) {
    const int32_t code = (int32_t)(look >> (16 - l));
    if (code <= h.maxcode[l]) { b.skip(l); return h.vals[h.valptr[l] + code - h.mincode[l]]; }
  }
  return -1;
}
static inline int jextend(uint32_t v, int t) { return t == 0 ? 0 : ((int)v >= (1 << (t - 1)) ? (int)v : (int)v - (1 << t) + 1); }

struct JFrame {
  int h = 0, w = 0, ncomp = 0;
  int hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, id[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
  int mcux = 0, mcuy = 0;            // MCUs per row / column
  int orient = 1;                    // EXIF orientation (tag 0x0112 of IFD0), 1 if absent: cv2.imread returns the image turned accordingly
  int dri = 0;
  size_t scan = 0;                   // sequential: offset of the entropy-coded data; progressive: offset of the first SOS marker
  bool progressive = false;
  uint16_t qt[4][64];                // natural order
  bool qt_present[4] = {false, false, false, false};
  JHuff dc[4], ac[4];
};

// DQT / DHT / DRI segment bodies (jdmarker.c get_dqt, get_dht, get_dri): these may stand in front of ANY scan, so both the header parser and the
// progressive decoder's scan loop come here. lock_qt: a table already defined may not change any more (libjpeg latches a component's table at
// its first scan; a file that redefines one between scans is not taken)
static int jtables(int m, const uint8_t* s, size_t sl, JFrame& f, bool lock_qt, std::string& why) {
  if (m == 0xDB) {
    size_t j = 0;
    while (j < sl) {
      const int pq = s[j] >> 4, t = s[j] & 15;
      ++j;
      if (t > 3 || pq > 1 || j + (pq ? 128 : 64) > sl) { why = "bad DQT"; return CTPN_ERR_ARG; }
      uint16_t q[64];
      for (int k = 0; k < 64; ++k) q[kZigzag[k]] = pq ? (uint16_t)((s[j + 2 * k] << 8) | s[j + 2 * k + 1]) : s[j + k];
      j += pq ? 128 : 64;
      if (lock_qt && f.qt_present[t] && std::memcmp(q, f.qt[t], sizeof(q)) != 0) { why = "quantisation table redefined between scans"; return CTPN_ERR_UNSUPPORTED; }
      std::memcpy(f.qt[t], q, sizeof(q));
      f.qt_present[t] = true;
    }
  } else if (m == 0xC4) {
    size_t j = 0;
    while (j + 17 <= sl) {
      const int tc = s[j] >> 4, th = s[j] & 15;
      int nv = 0;
      for (int k = 0; k < 16; ++k) nv += s[j + 1 + k];
      if (tc > 1 || th > 3 || nv > 256 || j + 17 + nv > sl) { why = "bad DHT"; return CTPN_ERR_ARG; }
      if (!jhuff_build(tc ? f.ac[th] : f.dc[th], s + j + 1, s + j + 17, nv)) { why = "bad Huffman table"; return CTPN_ERR_ARG; }
      j += 17 + nv;
    }
  } else if (m == 0xDD) {
    if (sl >= 2) f.dri = (s[0] << 8) | s[1];
  }
  return CTPN_OK;
}

// EXIF orientation (tag 0x0112 of IFD0) from an APP1 segment body, 1 if there is none: cv2.imread turns the image accordingly (OpenCV >= 3.1,
// unless IMREAD_IGNORE_ORIENTATION); the colour kernel's index map (jpeg_orient) applies it. is_exif: the body carries the "Exif\0\0"
// signature -- the FIRST such segment decides (OpenCV's ExifReader and Pillow's getexif() both read one TIFF header, the first; a later
// EXIF APP1 that says something else is ignored by both, so it is ignored here)
static int jexif_orientation(const uint8_t* s, size_t sl, bool& is_exif) {
  is_exif = sl >= 6 && std::memcmp(s, "Exif\0\0", 6) == 0;
  if (sl < 14 || !is_exif) return 1;
  const uint8_t* t = s + 6;
  const size_t tl = sl - 6;
  const bool le = t[0] == 'I' && t[1] == 'I';
  if (!le && !(t[0] == 'M' && t[1] == 'M')) return 1;
  auto u16 = [&](size_t o) -> uint32_t { return le ? (uint32_t)(t[o] | (t[o + 1] << 8)) : (uint32_t)((t[o] << 8) | t[o + 1]); };
  auto u32 = [&](size_t o) -> uint32_t { return le ? (u16(o) | (u16(o + 2) << 16)) : ((u16(o) << 16) | u16(o + 2)); };
  if (u16(2) != 42) return 1;
  const size_t ifd = u32(4);
  if (ifd + 2 > tl) return 1;
  const uint32_t n = u16(ifd);
  for (uint32_t k = 0; k < n; ++k) {
    const size_t e = ifd + 2 + 12 * (size_t)k;
    if (e + 12 > tl) return 1;
    if (u16(e) == 0x0112) { const uint32_t v = u16(e + 8); return (u16(e + 2) == 3 && v >= 1 && v <= 8) ? (int)v : 1; }
  }
  return 1;
}

static int jparse(const uint8_t* d, size_t len, JFrame& f, std::string& why) {
  if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) { why = "not a JPEG (no SOI)"; return CTPN_ERR_ARG; }
  size_t i = 2;
  bool have_frame = false, saw_jfif = false, saw_adobe = false, saw_exif = false;
  int adobe_transform = 0, orientation = 1;
  while (i + 4 <= len) {
    if (d[i] != 0xFF) { why = "marker expected"; return CTPN_ERR_ARG; }
    const int m = d[i + 1];
    i += 2;
    if (m == 0xFF) { --i; continue; }                                  // fill bytes
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (m == 0xD9) break;
    const size_t L = ((size_t)d[i] << 8) | d[i + 1];
    if (L < 2 || i + L > len) { why = "segment runs past the end of the file"; return CTPN_ERR_ARG; }
    const uint8_t* s = d + i + 2;
    const size_t sl = L - 2;
    i += L;
    if (m == 0xDB || m == 0xC4 || m == 0xDD) {
      const int rc = jtables(m, s, sl, f, false, why);
      if (rc) return rc;
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (have_frame) { why = "second frame header"; return CTPN_ERR_ARG; }
      if (sl < 6 || s[0] != 8) { why = "only 8-bit samples"; return CTPN_ERR_UNSUPPORTED; }
      f.progressive = m == 0xC2;
      f.h = (s[1] << 8) | s[2]; f.w = (s[3] << 8) | s[4]; f.ncomp = s[5];
      if (f.ncomp != 1 && f.ncomp != 3) { why = "1 or 3 components only"; return CTPN_ERR_UNSUPPORTED; }
      if (sl < (size_t)(6 + 3 * f.ncomp) || f.h <= 0 || f.w <= 0) { why = "bad SOF"; return CTPN_ERR_ARG; }
      for (int k = 0; k < f.ncomp; ++k) { f.id[k] = s[6 + 3 * k]; f.hs[k] = s[7 + 3 * k] >> 4; f.vs[k] = s[7 + 3 * k] & 15; f.tq[k] = s[8 + 3 * k] & 3; }
      have_frame = true;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      why = "lossless / hierarchical / arithmetic-coded JPEG"; return CTPN_ERR_UNSUPPORTED;
    } else if (m == 0xE0) {
      if (sl >= 5 && std::memcmp(s, "JFIF", 5) == 0) saw_jfif = true;
    } else if (m == 0xEE) {
      if (sl >= 12 && std::memcmp(s, "Adobe", 5) == 0) { saw_adobe = true; adobe_transform = s[11]; }
    } else if (m == 0xE1) {
      if (!saw_exif) { bool is_exif = false; const int o = jexif_orientation(s, sl, is_exif); if (is_exif) { saw_exif = true; orientation = o; } }
    } else if (m == 0xDA) {
      if (!have_frame) { why = "SOS before SOF"; return CTPN_ERR_ARG; }
      if (sl < 1) { why = "bad SOS"; return CTPN_ERR_ARG; }
      const int ns = f.progressive ? 0 : s[0];      // a progressive frame's scan headers are read by jprogressive, scan after scan
      if (!f.progressive && (ns != f.ncomp || sl < (size_t)(1 + 2 * ns + 3))) { why = "multi-scan sequential files are not supported"; return CTPN_ERR_UNSUPPORTED; }
      for (int k = 0; k < ns; ++k) {
        int c = -1;
        for (int q = 0; q < f.ncomp; ++q) if (f.id[q] == s[1 + 2 * k]) c = q;
        if (c != k) { why = "scan component order"; return CTPN_ERR_UNSUPPORTED; }
        f.td[k] = s[2 + 2 * k] >> 4; f.ta[k] = s[2 + 2 * k] & 15;
        if (f.td[k] > 3 || f.ta[k] > 3 || !f.dc[f.td[k]].present || !f.ac[f.ta[k]].present || !f.qt_present[f.tq[k]]) { why = "scan refers to a missing table"; return CTPN_ERR_ARG; }
      }
      f.scan = f.progressive ? i - L - 2 : i;
      f.orient = orientation;          // applied by the colour kernel's index map (jpeg_orient)
      if (f.ncomp == 1) { f.hs[0] = f.vs[0] = 1; }
      else {
        // which colour space the three components are in, by libjpeg's rule (jdapimin.c default_decompress_parms): JFIF says YCbCr; else an
        // Adobe marker's transform flag (0 = RGB as stored, 1 = YCbCr); else the component ids ('R', 'G', 'B' = RGB, anything else YCbCr).
        // The device half converts YCbCr: a file that stores RGB is another decoder's
        const bool rgb = saw_jfif ? false : (saw_adobe ? adobe_transform == 0 : (f.id[0] == 'R' && f.id[1] == 'G' && f.id[2] == 'B'));
        if (rgb) { why = "RGB-coded JPEG (no YCbCr transform)"; return CTPN_ERR_UNSUPPORTED; }
        const bool c11 = f.hs[1] == 1 && f.vs[1] == 1 && f.hs[2] == 1 && f.vs[2] == 1;
        if (!c11 || !((f.hs[0] == 1 && (f.vs[0] == 1 || f.vs[0] == 2)) || (f.hs[0] == 2 && (f.vs[0] == 2 || f.vs[0] == 1)))) { why = "chroma subsampling other than 4:4:4 / 4:4:0 / 4:2:2 / 4:2:0"; return CTPN_ERR_UNSUPPORTED; }
      }
      f.mcux = (f.w + 8 * f.hs[0] - 1) / (8 * f.hs[0]);
      f.mcuy = (f.h + 8 * f.vs[0] - 1) / (8 * f.vs[0]);
      return CTPN_OK;
    }
  }
  why = "no scan found";
  return CTPN_ERR_ARG;
}

// blocks of component c: [mcuy * vs][mcux * hs][64] int16, component after component
static size_t jcoef_count(const JFrame& f) {
  size_t n = 0;
  for (int c = 0; c < f.ncomp; ++c) n += (size_t)f.mcuy * f.vs[c] * f.mcux * f.hs[c] * 64;
  return n;
}

static int jentropy(const uint8_t* d, size_t len, const JFrame& f, int16_t* coef, std::string& why) {
  std::memset(coef, 0, jcoef_count(f) * sizeof(int16_t));
  int16_t* base[3]; int bw[3];
  {
    size_t off = 0;
    for (int c = 0; c < f.ncomp; ++c) { base[c] = coef + off; bw[c] = f.mcux * f.hs[c]; off += (size_t)f.mcuy * f.vs[c] * bw[c] * 64; }
  }
  JBits b;
  b.p = d + f.scan; b.end = d + len;
  int pred[3] = {0, 0, 0};
  long long n = 0;
  for (int my = 0; my < f.mcuy; ++my)
    for (int mx = 0; mx < f.mcux; ++mx, ++n) {
      if (f.dri && n && n % f.dri == 0) {
        // byte-align, find RSTn
        if (b.overran()) { why = "entropy-coded data ends inside a restart interval"; return CTPN_ERR_UNSUPPORTED; }
        const uint8_t* p = b.p;
        while (p + 1 < b.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
        if (p + 1 >= b.end) { why = "restart marker missing"; return CTPN_ERR_ARG; }
        b.restart_at(p + 2);
        pred[0] = pred[1] = pred[2] = 0;
      }
      for (int c = 0; c < f.ncomp; ++c) {
        const JHuff& hd = f.dc[f.td[c]];
        const JHuff& ha = f.ac[f.ta[c]];
        for (int by = 0; by < f.vs[c]; ++by)
          for (int bx = 0; bx < f.hs[c]; ++bx) {
            int16_t* blk = base[c] + ((size_t)(my * f.vs[c] + by) * bw[c] + (mx * f.hs[c] + bx)) * 64;
            const int t = jdecode(b, hd);
            if (t < 0 || t > 11) { why = "corrupt DC code"; return CTPN_ERR_ARG; }
            pred[c] += jextend(b.get(t), t);
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              const int rs = jdecode(b, ha);
              if (rs < 0) { why = "corrupt AC code"; return CTPN_ERR_ARG; }
              const int r = rs >> 4, s = rs & 15;
              if (s == 0) {
                if (r != 15) break;
                k += 16;
                continue;
              }
              k += r;
              if (k > 63) { why = "AC run past the block"; return CTPN_ERR_ARG; }
              blk[kZigzag[k]] = (int16_t)jextend(b.get(s), s);
              ++k;
            }
          }
      }
    }
  if (b.overran()) { why = "entropy-coded data ends before the last MCU (truncated file)"; return CTPN_ERR_UNSUPPORTED; }
  return CTPN_OK;
}

// Progressive frames (SOF2), jdphuff.c: the coefficients arrive in several scans, each a band Ss..Se of the zigzag order at bit position
// Al -- a DC scan may interleave the components, an AC scan carries one -- first as the bits above Al ("first" scans, Ah = 0), then one bit at
// a time (refinement scans, Ah = Al + 1). Tables may change between scans. What the device gets is the finished coefficient array, the same
// layout as a sequential file's; libjpeg's output of a COMPLETE progressive file is the plain IDCT of it (its block smoothing only applies
// while AC bits are still missing). A file that ends early -- scans missing, or a scan cut short -- is CTPN_ERR_UNSUPPORTED: libjpeg's
// result for it (block smoothing, zero blocks behind a premature end) is not the plain IDCT of what was delivered.
struct JScan { int ns = 0, c[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0}, ss = 0, se = 0, ah = 0, al = 0; };

static inline void jrefine_nonzero(JBits& b, int16_t* coef, int p1, int m1) {      // a correction bit for a coefficient that is already nonzero
  if (b.get(1) && (*coef & p1) == 0) *coef = (int16_t)(*coef >= 0 ? *coef + p1 : *coef + m1);
}

// one block of one scan; eobrun and pred are the scan's running state
static inline int jprog_block(JBits& b, const JScan& sc, const JHuff* hd, const JHuff* ha, int16_t* blk, int& pred, int& eobrun) {
  const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
  if (sc.ss == 0) {
    if (sc.ah == 0) {                                  // decode_mcu_DC_first
      const int t = jdecode(b, *hd);
      if (t < 0 || t > 15) return -1;
      pred += jextend(b.get(t), t);
      blk[0] = (int16_t)(pred * (1 << sc.al));
    } else if (b.get(1)) blk[0] |= (int16_t)p1;        // decode_mcu_DC_refine
    return 0;
  }
  if (sc.ah == 0) {                                    // decode_mcu_AC_first
    if (eobrun > 0) { --eobrun; return 0; }
    for (int k = sc.ss; k <= sc.se; ++k) {
      const int rs = jdecode(b, *ha);
      if (rs < 0) return -1;
      const int r = rs >> 4, s = rs & 15;
      if (s) {
        k += r;
        if (k > 63) return -1;
        blk[kZigzag[k]] = (int16_t)(jextend(b.get(s), s) * (1 << sc.al));
      } else if (r == 15) k += 15;
      else {
        eobrun = 1 << r;
        if (r) eobrun += (int)b.get(r);
        --eobrun;                                      // this block is the first of the run
        break;
      }
    }
    return 0;
  }
  int k = sc.ss;                                       // decode_mcu_AC_refine
  if (eobrun == 0) {
    for (; k <= sc.se; ++k) {
      const int rs = jdecode(b, *ha);
      if (rs < 0) return -1;
      int r = rs >> 4, s = rs & 15;
      if (s) s = b.get(1) ? p1 : m1;                   // the size of a newly nonzero coefficient is always 1: only its sign is coded
      else if (r != 15) {
        eobrun = 1 << r;
        if (r) eobrun += (int)b.get(r);
        break;                                         // the rest of the band is handled below, as the first block of the run
      }
      // skip r coefficients that are still zero, giving every nonzero one on the way its correction bit; r = 15, s = 0: sixteen of them
      for (; k <= sc.se; ++k) {
        int16_t* co = blk + kZigzag[k];
        if (*co != 0) jrefine_nonzero(b, co, p1, m1);
        else if (--r < 0) break;
      }
      if (s) {
        if (k > 63) return -1;
        blk[kZigzag[k]] = (int16_t)s;
      }
    }
  }
  if (eobrun > 0) {
    for (; k <= sc.se; ++k) {
      int16_t* co = blk + kZigzag[k];
      if (*co != 0) jrefine_nonzero(b, co, p1, m1);
    }
    --eobrun;
  }
  return 0;
}

static int jprogressive(const uint8_t* d, size_t len, JFrame& f, int16_t* coef, std::string& why) {
  std::memset(coef, 0, jcoef_count(f) * sizeof(int16_t));
  int16_t* base[3]; int bw[3], cw[3], ch[3];           // padded block columns; the component's own size in blocks (a one-component scan covers these)
  {
    size_t off = 0;
    for (int c = 0; c < f.ncomp; ++c) {
      base[c] = coef + off; bw[c] = f.mcux * f.hs[c]; off += (size_t)f.mcuy * f.vs[c] * bw[c] * 64;
      cw[c] = ((f.w * f.hs[c] + f.hs[0] - 1) / f.hs[0] + 7) / 8;
      ch[c] = ((f.h * f.vs[c] + f.vs[0] - 1) / f.vs[0] + 7) / 8;
    }
  }
  size_t i = f.scan;
  int scans = 0;
  // libjpeg's coef_bits: the bit position Al every coefficient of every component has been delivered down to (-1: never). While any is not
  // 0 at the end of the file, libjpeg's output is NOT the plain IDCT of the coefficients (jdcoefct.c smooths the blocks from their
  // neighbours' DC values): such a file -- one cut at a scan boundary, say -- is the host decoder's
  int8_t cbits[3][64];
  std::memset(cbits, -1, sizeof(cbits));
  while (i + 4 <= len) {
    if (d[i] != 0xFF) { ++i; continue; }               // between scans: whatever is left of the entropy-coded segment
    const int m = d[i + 1];
    if (m == 0xFF) { ++i; continue; }
    if (m == 0x00 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { i += 2; continue; }
    if (m == 0xD9) break;
    i += 2;
    const size_t L = ((size_t)d[i] << 8) | d[i + 1];
    if (L < 2 || i + L > len) { if (scans) break; why = "segment runs past the end of the file"; return CTPN_ERR_ARG; }
    const uint8_t* s = d + i + 2;
    const size_t sl = L - 2;
    i += L;
    if (m == 0xDB || m == 0xC4 || m == 0xDD) {
      const int rc = jtables(m, s, sl, f, scans > 0, why);
      if (rc) return rc;
      continue;
    }
    if (m != 0xDA) {
      if (m >= 0xC0 && m <= 0xCF) { why = "second frame header"; return CTPN_ERR_ARG; }
      continue;                                        // APPn, COM between scans
    }
    JScan sc;
    if (sl < 1) { why = "bad SOS"; return CTPN_ERR_ARG; }
    // every scan walks every block of its components: a crafted file of thousands of tiny scans is hours of work (libjpeg-turbo's own
    // tools limit the scan count for the same reason); real progressive files have ~10, jpegtran's finest scripts a few dozen
    if (scans >= 256) { why = "more than 256 progressive scans"; return CTPN_ERR_UNSUPPORTED; }
    sc.ns = s[0];
    if (sc.ns < 1 || sc.ns > f.ncomp || sl < (size_t)(1 + 2 * sc.ns + 3)) { why = "bad SOS"; return CTPN_ERR_ARG; }
    for (int k = 0; k < sc.ns; ++k) {
      int c = -1;
      for (int q = 0; q < f.ncomp; ++q) if (f.id[q] == s[1 + 2 * k]) c = q;
      if (c < 0 || (k > 0 && c <= sc.c[k - 1])) { why = "scan component order"; return CTPN_ERR_ARG; }
      sc.c[k] = c; sc.td[k] = s[2 + 2 * k] >> 4; sc.ta[k] = s[2 + 2 * k] & 15;
      if (sc.td[k] > 3 || sc.ta[k] > 3) { why = "bad SOS"; return CTPN_ERR_ARG; }
    }
    sc.ss = s[1 + 2 * sc.ns]; sc.se = s[2 + 2 * sc.ns]; sc.ah = s[3 + 2 * sc.ns] >> 4; sc.al = s[3 + 2 * sc.ns] & 15;
    // jdphuff.c start_pass_phuff_decoder's validity rules
    const bool dc_scan = sc.ss == 0;
    if ((dc_scan && sc.se != 0) || (!dc_scan && (sc.se < sc.ss || sc.se > 63 || sc.ns != 1)) || sc.al > 13 || (sc.ah != 0 && sc.ah != sc.al + 1)) {
      why = "bad progressive scan parameters"; return CTPN_ERR_ARG;
    }
    for (int k = 0; k < sc.ns; ++k) {
      const bool need_dc = dc_scan && sc.ah == 0, need_ac = !dc_scan;
      if ((need_dc && !f.dc[sc.td[k]].present) || (need_ac && !f.ac[sc.ta[k]].present)) { why = "scan refers to a missing table"; return CTPN_ERR_ARG; }
    }
    // the scan's entropy-coded segment
    JBits b;
    b.p = d + i; b.end = d + len;
    int pred[3] = {0, 0, 0}, eobrun = 0;
    const bool inter = sc.ns > 1;                      // one component: its blocks in raster order, no MCU padding
    const int c0 = sc.c[0];
    const int ux = inter ? f.mcux : cw[c0], uy = inter ? f.mcuy : ch[c0];      // restart units: MCUs, or the one component's blocks
    long long n = 0;
    for (int my = 0; my < uy; ++my)
      for (int mx = 0; mx < ux; ++mx, ++n) {
        if (f.dri && n && n % f.dri == 0) {
          if (b.overran()) { why = "a progressive scan's data ends inside a restart interval"; return CTPN_ERR_UNSUPPORTED; }
          const uint8_t* p = b.p;
          while (p + 1 < b.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) {
            if (p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF) { p = b.end; break; }      // another marker: the scan ended early
            ++p;
          }
          if (p + 1 >= b.end) { why = "restart marker missing"; return CTPN_ERR_ARG; }
          b.restart_at(p + 2);
          pred[0] = pred[1] = pred[2] = 0; eobrun = 0;
        }
        if (inter) {
          for (int k = 0; k < sc.ns; ++k) {
            const int c = sc.c[k];
            for (int by = 0; by < f.vs[c]; ++by)
              for (int bx = 0; bx < f.hs[c]; ++bx) {
                int16_t* blk = base[c] + ((size_t)(my * f.vs[c] + by) * bw[c] + (mx * f.hs[c] + bx)) * 64;
                if (jprog_block(b, sc, &f.dc[sc.td[k]], &f.ac[sc.ta[k]], blk, pred[k], eobrun)) { why = "corrupt progressive scan"; return CTPN_ERR_ARG; }
              }
          }
        } else {
          int16_t* blk = base[c0] + ((size_t)my * bw[c0] + mx) * 64;
          if (jprog_block(b, sc, &f.dc[sc.td[0]], &f.ac[sc.ta[0]], blk, pred[0], eobrun)) { why = "corrupt progressive scan"; return CTPN_ERR_ARG; }
        }
      }
    if (b.overran()) { why = "a progressive scan's data ends before its last block (truncated file)"; return CTPN_ERR_UNSUPPORTED; }
    for (int k = 0; k < sc.ns; ++k)
      for (int z = sc.ss; z <= sc.se; ++z) cbits[sc.c[k]][z] = (int8_t)sc.al;
    ++scans;
    i = (size_t)(b.p - d);                             // the reader never passes a marker: the next one is at or behind it
  }
  if (!scans) { why = "no scan found"; return CTPN_ERR_ARG; }
  for (int c = 0; c < f.ncomp; ++c) if (!f.qt_present[f.tq[c]]) { why = "missing quantisation table"; return CTPN_ERR_ARG; }
  for (int c = 0; c < f.ncomp; ++c)
    for (int z = 0; z < 64; ++z)
      if (cbits[c][z] != 0) { why = "progressive file without its last scans (coefficient precision incomplete: libjpeg smooths such an image)"; return CTPN_ERR_UNSUPPORTED; }
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// device: dequantise + islow IDCT (8 threads per block: a column in pass 1, a row in pass 2, through LDS) -> planes;
// planes -> BGR. plane c of image i: [ph[c]][pw[c]] uint8 at plane_off[c] of the image's plane block
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const uint16_t* __restrict__ qt /* [n][3][64] */,
                                                         uint8_t* __restrict__ planes, JpegGeom g, int n_img) {
  __shared__ int ws[32][8][9];
  const int tid = threadIdx.x, lb = tid >> 3, t = tid & 7;
  const long long gb = (long long)blockIdx.x * 32 + lb;                 // global block index
  const bool live = gb < g.blocks_per_img * n_img;
  const int img = live ? (int)(gb / g.blocks_per_img) : 0;
  long long b = live ? gb - (long long)img * g.blocks_per_img : 0;
  int c = 0;
  while (c + 1 < g.ncomp && b >= (long long)g.bw[c] * g.bh[c]) { b -= (long long)g.bw[c] * g.bh[c]; ++c; }
  const int by = (int)(b / g.bw[c]), bx = (int)(b - (long long)by * g.bw[c]);
  const int16_t* blk = coef + (long long)img * g.coef_per_img + g.coef_off[c] + b * 64;
  const uint16_t* q = qt + ((long long)img * 3 + c) * 64;
  // pass 1: column t (elements t, t + 8, ...), dequantised
  int x[8], o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = live ? (int)blk[8 * k + t] * (int)q[8 * k + t] : 0;
  jidct_1d(x, o, 13 - 2);
#pragma unroll
  for (int k = 0; k < 8; ++k) ws[lb][k][t] = o[k];                      // ws[row][col]
  __syncthreads();
  // pass 2: row t
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = ws[lb][t][k];
  jidct_1d(x, o, 13 + 2 + 3);
  if (!live) return;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int v = o[k] + 128; v = v < 0 ? 0 : (v > 255 ? 255 : v); lo |= (uint32_t)v << (8 * k);
    int u = o[k + 4] + 128; u = u < 0 ? 0 : (u > 255 ? 255 : u); hi |= (uint32_t)u << (8 * k);
  }
  uint8_t* dst = planes + (long long)img * g.plane_per_img + g.plane_off[c] + ((long long)(by * 8 + t) * (g.bw[c] * 8) + bx * 8);
  *(uint2*)dst = make_uint2(lo, hi);
}

// four consecutive pixels of the batch's linear pixel order per thread: 12 output bytes = three aligned dwords (the group may straddle a row
// or an image; every pixel finds its own coordinates)
__global__ __launch_bounds__(256) void jpeg_color_kernel(const uint8_t* __restrict__ planes, uint8_t* __restrict__ out, JpegGeom g, int n_img) {
  const long long grp = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)g.oh * g.ow, total = per * n_img;
  const long long p0 = grp * 4;
  if (p0 >= total) return;
  int img = (int)(p0 / per);
  int rem = (int)(p0 - (long long)img * per);
  int y = rem / g.ow, x = rem - y * g.ow;            // in the turned image (EXIF orientation; the stored one for orientation 1)
  uint32_t px[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (p0 + k < total) {
      int sy, sx;
      jpeg_orient(g.orient, g.h, g.w, y, x, sy, sx);
      px[k] = jpeg_pixel(planes + (long long)img * g.plane_per_img, g, sy, sx);
    }
    if (++x == g.ow) { x = 0; if (++y == g.oh) { y = 0; ++img; } }
  }
  uint32_t* o = (uint32_t*)(out + p0 * 3);
  if (p0 + 4 <= total) {
    o[0] = px[0] | (px[1] << 24);
    o[1] = (px[1] >> 8) | (px[2] << 16);
    o[2] = (px[2] >> 16) | (px[3] << 8);
  } else {
    uint8_t* ob = out + p0 * 3;
    for (int k = 0; k < 4 && p0 + k < total; ++k) { ob[3 * k] = (uint8_t)px[k]; ob[3 * k + 1] = (uint8_t)(px[k] >> 8); ob[3 * k + 2] = (uint8_t)(px[k] >> 16); }
  }
}

// ---------------------------------------------------------------------------------------------
// entry points used by ctpn_api.hip
// ---------------------------------------------------------------------------------------------
int jpeg_probe(const uint8_t* data, size_t len, int* h, int* w, int* ncomp, int* luma_sampling) {
  JFrame f; std::string why;
  const int rc = jparse(data, len, f, why);
  if (rc) return fail(rc, "jpeg: " + why);
  // the size cv2.imread returns: orientations 5 .. 8 swap the stored height and width
  if (h) *h = f.orient >= 5 ? f.w : f.h;
  if (w) *w = f.orient >= 5 ? f.h : f.w;
  if (ncomp) *ncomp = f.ncomp;
  // 1, 2, 0x21 for 2 x 1, 0x12 for 1 x 2; | (orientation - 1) << 8: the files of one device batch share layout AND orientation
  if (luma_sampling) *luma_sampling = (f.hs[0] == f.vs[0] ? f.hs[0] : f.hs[0] * 16 + f.vs[0]) | ((f.orient - 1) << 8);
  return CTPN_OK;
}

static void jgeom(const JFrame& f, JpegGeom& g) {
  g.h = f.h; g.w = f.w; g.ncomp = f.ncomp; g.hs0 = f.hs[0]; g.vs0 = f.vs[0];
  g.orient = f.orient; g.oh = f.orient >= 5 ? f.w : f.h; g.ow = f.orient >= 5 ? f.h : f.w;
  long long co = 0, po = 0, nb = 0;
  for (int c = 0; c < 3; ++c) { g.bw[c] = g.bh[c] = 0; g.coef_off[c] = g.plane_off[c] = 0; }
  for (int c = 0; c < f.ncomp; ++c) {
    g.bw[c] = f.mcux * f.hs[c]; g.bh[c] = f.mcuy * f.vs[c];
    g.coef_off[c] = co; g.plane_off[c] = po;
    co += (long long)g.bw[c] * g.bh[c] * 64; po += (long long)g.bw[c] * g.bh[c] * 64; nb += (long long)g.bw[c] * g.bh[c];
  }
  g.coef_per_img = co; g.plane_per_img = po; g.blocks_per_img = nb;
}

size_t jpeg_coef_capacity(int h, int w) {      // int16 elements one image of h x w can need in any supported layout (MCU padding included)
  const long long mx = (w + 7) / 8, my = (h + 7) / 8, mx2 = (w + 15) / 16, my2 = (h + 15) / 16;
  const long long a = 3 * mx * my * 64, b = (4 + 2) * mx2 * my2 * 64;      // 4:4:4, 4:2:0
  // 4:2:2 and 4:4:0, each for the image stored as h x w or as w x h (EXIF orientations 5 .. 8: h, w are the TURNED image's)
  const long long c = (2 + 2) * mx2 * my * 64, d = (2 + 2) * mx * my2 * 64;
  return (size_t)std::max(std::max(a, b), std::max(c, d));
}

// host half: file bytes -> coefficient block + quantisation tables of ONE image; fills *g
int jpeg_entropy_decode(const uint8_t* data, size_t len, int16_t* coef, size_t coef_cap, uint16_t* qt3x64, JpegGeom* g) {
  JFrame f; std::string why;
  int rc = jparse(data, len, f, why);
  if (rc) return fail(rc, "jpeg: " + why);
  if (jcoef_count(f) > coef_cap) return fail(CTPN_ERR_CAPACITY, "jpeg: coefficient buffer too small");
  if ((rc = f.progressive ? jprogressive(data, len, f, coef, why) : jentropy(data, len, f, coef, why))) return fail(rc, "jpeg: " + why);
  for (int c = 0; c < 3; ++c) std::memcpy(qt3x64 + 64 * c, f.qt[f.tq[c < f.ncomp ? c : 0]], 64 * sizeof(uint16_t));
  jgeom(f, *g);
  return CTPN_OK;
}

// device half: n images of identical geometry g; coef_dev [n][coef_per_img] int16, qt_dev [n][3][64] uint16, planes_dev scratch
// [n][plane_per_img], out_dev [n][h][w][3] uint8 BGR
int launch_jpeg_pixels(const int16_t* coef_dev, const uint16_t* qt_dev, uint8_t* planes_dev, uint8_t* out_dev, const JpegGeom& g, int n, hipStream_t s) {
  const long long nblk = g.blocks_per_img * n;
  if (nblk <= 0 || (nblk + 31) / 32 > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "jpeg: grid out of range");
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblk + 31) / 32)), dim3(256), 0, s, coef_dev, qt_dev, planes_dev, g, n);
  const long long groups = ((long long)g.oh * g.ow * n + 3) / 4;
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, planes_dev, out_dev, g, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("jpeg launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
