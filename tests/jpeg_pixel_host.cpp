// TEST INFRASTRUCTURE (never part of the product library): the device half's per-sample source text -- csrc/jpeg_pixel.h, the functions
// jpeg_idct_kernel and jpeg_color_kernel are made of -- compiled for the host with the HIP qualifiers defined away, driven the way the two
// kernels drive it (column pass, row pass, clamp; then one jpeg_pixel call per output pixel). tests/test_jpeg.py builds this file with g++
// and compares its output with Pillow's decode, so the arithmetic of every layout is pinned on the CPU from the very text hipcc compiles.
#define __host__
#define __device__
#define __forceinline__ inline
#include "../text-detection-ctpn_amd/csrc/jpeg_pixel.h"

#include <stddef.h>
#include <vector>

extern "C" int jpeg_pixels_host(const int16_t* coef, const uint16_t* qt3x64, const int* layout8, uint8_t* out_bgr) {
  using namespace ctpn;
  JpegGeom g;
  g.h = layout8[0]; g.w = layout8[1]; g.ncomp = layout8[2]; g.hs0 = layout8[3] & 0xff;
  g.orient = (layout8[3] >> 8) + 1; g.oh = g.orient >= 5 ? g.w : g.h; g.ow = g.orient >= 5 ? g.h : g.w;
  long long co = 0;
  for (int c = 0; c < 3; ++c) { g.bw[c] = g.bh[c] = 0; g.coef_off[c] = g.plane_off[c] = 0; }
  for (int c = 0; c < g.ncomp; ++c) {
    g.bw[c] = layout8[c == 0 ? 4 : 5]; g.bh[c] = layout8[c == 0 ? 6 : 7];
    g.coef_off[c] = g.plane_off[c] = co;
    co += (long long)g.bw[c] * g.bh[c] * 64;
  }
  g.vs0 = g.ncomp == 3 ? g.bh[0] / g.bh[1] : 1;
  g.coef_per_img = g.plane_per_img = co;
  std::vector<uint8_t> planes((size_t)co);
  for (int c = 0; c < g.ncomp; ++c)
    for (int by = 0; by < g.bh[c]; ++by)
      for (int bx = 0; bx < g.bw[c]; ++bx) {
        const int16_t* blk = coef + g.coef_off[c] + ((long long)by * g.bw[c] + bx) * 64;
        const uint16_t* q = qt3x64 + 64 * c;
        int ws[8][8], x[8], o[8];
        for (int t = 0; t < 8; ++t) {                      // pass 1: thread t takes column t
          for (int k = 0; k < 8; ++k) x[k] = (int)blk[8 * k + t] * (int)q[8 * k + t];
          jidct_1d(x, o, 13 - 2);
          for (int k = 0; k < 8; ++k) ws[k][t] = o[k];
        }
        for (int t = 0; t < 8; ++t) {                      // pass 2: thread t takes row t
          for (int k = 0; k < 8; ++k) x[k] = ws[t][k];
          jidct_1d(x, o, 13 + 2 + 3);
          uint8_t* dst = planes.data() + g.plane_off[c] + ((long long)(by * 8 + t) * (g.bw[c] * 8) + bx * 8);
          for (int k = 0; k < 8; ++k) { int v = o[k] + 128; dst[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
        }
      }
  for (int y = 0; y < g.oh; ++y)                           // the colour kernel: one turned-image pixel at a time (jpeg_orient: EXIF orientation)
    for (int x = 0; x < g.ow; ++x) {
      int sy, sx;
      jpeg_orient(g.orient, g.h, g.w, y, x, sy, sx);
      const uint32_t p = jpeg_pixel(planes.data(), g, sy, sx);
      uint8_t* o = out_bgr + ((long long)y * g.ow + x) * 3;
      o[0] = (uint8_t)p; o[1] = (uint8_t)(p >> 8); o[2] = (uint8_t)(p >> 16);
    }
  return 0;
}
