// The per-sample arithmetic of the JPEG decoder's device half (jpeg.hip), in a header of its own so that ONE source serves two compilers:
// hipcc compiles it into jpeg_idct_kernel / jpeg_color_kernel; tests/test_jpeg.py compiles the same text with g++ (the HIP qualifiers
// defined away) into a small checker library and compares it with Pillow's decode on the CPU -- a new layout's arithmetic is pinned before
// it ever reaches a GPU. Nothing here is product host code: the library itself only ever calls these functions from its kernels.
#pragma once
#include <stdint.h>

namespace ctpn {

struct JpegGeom {
  int h, w, ncomp, hs0, vs0;         // luma sampling factors: 1 x 1 (4:4:4 / gray), 2 x 2 (4:2:0), 2 x 1 (4:2:2), 1 x 2 (4:4:0); the chroma planes' are 1 x 1
  int orient, oh, ow;                // EXIF orientation 1 .. 8 (cv2.imread applies it) and the size of the turned image: (h, w) for 1 .. 4, (w, h) for 5 .. 8
  int bw[3], bh[3];                  // blocks per row / column of every component
  long long coef_off[3];             // int16 offset of component c inside an image's coefficient block
  long long plane_off[3];            // byte offset of component c inside an image's plane block
  long long coef_per_img, plane_per_img;
  long long blocks_per_img;
};

__host__ __device__ __forceinline__ void jidct_1d(const int (&x)[8], int (&o)[8], int descale) {
  // jidctint.c: even part
  int z2 = x[2], z3 = x[6];
  int z1 = (z2 + z3) * 4433;
  const int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
  z2 = x[0]; z3 = x[4];
  const int tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  // odd part
  int t0 = x[7], t1 = x[5], t2 = x[3], t3 = x[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
  int z4 = t1 + t3;
  const int z5 = (z3 + z4) * 9633;
  t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 = z3 * (-16069) + z5; z4 = z4 * (-3196) + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  const int r = 1 << (descale - 1);
  o[0] = (tmp10 + t3 + r) >> descale; o[7] = (tmp10 - t3 + r) >> descale;
  o[1] = (tmp11 + t2 + r) >> descale; o[6] = (tmp11 - t2 + r) >> descale;
  o[2] = (tmp12 + t1 + r) >> descale; o[5] = (tmp12 - t1 + r) >> descale;
  o[3] = (tmp13 + t0 + r) >> descale; o[4] = (tmp13 - t0 + r) >> descale;
}

// pixel (oy, ox) of the image as cv2.imread returns it -> the stored pixel it is (EXIF orientation, tag 0x0112: 2 mirrored left-right,
// 3 turned by 180 degrees, 4 mirrored top-bottom, 5 transposed, 6 needs a quarter turn clockwise, 7 transverse, 8 a quarter turn
// anti-clockwise); h, w: the STORED size
__host__ __device__ __forceinline__ void jpeg_orient(int orient, int h, int w, int oy, int ox, int& y, int& x) {
  switch (orient) {
    case 2: y = oy; x = w - 1 - ox; break;
    case 3: y = h - 1 - oy; x = w - 1 - ox; break;
    case 4: y = h - 1 - oy; x = ox; break;
    case 5: y = ox; x = oy; break;
    case 6: y = h - 1 - ox; x = oy; break;
    case 7: y = h - 1 - ox; x = w - 1 - oy; break;
    case 8: y = ox; x = w - 1 - oy; break;
    default: y = oy; x = ox; break;
  }
}

// one pixel: chroma upsampling + colour conversion -> B | G << 8 | R << 16
__host__ __device__ __forceinline__ uint32_t jpeg_pixel(const uint8_t* __restrict__ P, const JpegGeom& g, int y, int x) {
  const int Y = P[g.plane_off[0] + (long long)y * (g.bw[0] * 8) + x];
  if (g.ncomp == 1) return (uint32_t)Y * 0x010101u;
  int cb, cr;
  if (g.hs0 == 1 && g.vs0 == 1) {
    cb = P[g.plane_off[1] + (long long)y * (g.bw[1] * 8) + x];
    cr = P[g.plane_off[2] + (long long)y * (g.bw[2] * 8) + x];
  } else if (g.hs0 == 1) {
    // jdsample.c h1v2_fancy_upsample (4:4:0; libjpeg-turbo >= 1.5): 3/4 nearer + 1/4 further ROW of the same column, + 1 for the upper and
    // + 2 for the lower output row of a pair; the row above the first / below the last real chroma row is that row again (the context rows
    // of jdmainct.c). Unlike the h2v1 / h2v2 filters it has no narrow-image exception (jinit_upsampler takes it whenever fancy upsampling is on)
    const int dh = (g.h + 1) >> 1;
    const int cy = y >> 1;
    int fy = (y & 1) ? cy + 1 : cy - 1;
    fy = fy < 0 ? 0 : (fy > dh - 1 ? dh - 1 : fy);
    const int bias = (y & 1) ? 2 : 1;
    int v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint8_t* C = P + g.plane_off[1 + k];
      const int pitch = g.bw[1 + k] * 8;
      v[k] = (3 * C[(long long)cy * pitch + x] + C[(long long)fy * pitch + x] + bias) >> 2;
    }
    cb = v[0]; cr = v[1];
  } else if (g.vs0 == 1) {
    // jdsample.c h2v1_fancy_upsample (4:2:2): 3/4 nearer + 1/4 further column of the SAME row, + 1 for even and + 2 for odd output columns;
    // the first and the last output column are the edge sample itself; plain replication where the downsampled width is <= 2
    const int dw = (g.w + 1) >> 1;
    const int cx = x >> 1;
    const int nx = (x & 1) ? cx + 1 : cx - 1;
    const bool edge = dw <= 2 || nx < 0 || nx > dw - 1;
    int v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint8_t* C = P + g.plane_off[1 + k] + (long long)y * (g.bw[1 + k] * 8);
      v[k] = edge ? C[cx] : (3 * C[cx] + C[nx] + ((x & 1) ? 2 : 1)) >> 2;
    }
    cb = v[0]; cr = v[1];
  } else {
    // jdsample.c h2v2_fancy_upsample: 3/4 nearer + 1/4 further in each direction; rows replicated at the top / bottom of the image,
    // the first / last column use (4 * colsum + 8 | 7) >> 4; + 8 for even output columns, + 7 for odd ones
    const int dw = (g.w + 1) >> 1, dh = (g.h + 1) >> 1;
    const int cy = y >> 1, cx = x >> 1;
    int v[2];
    if (dw > 2) {
      int fy = (y & 1) ? cy + 1 : cy - 1;
      fy = fy < 0 ? 0 : (fy > dh - 1 ? dh - 1 : fy);
      const int nx = (x & 1) ? cx + 1 : cx - 1;
      const bool edge = nx < 0 || nx > dw - 1;
      const int bias = (x & 1) ? 7 : 8;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint8_t* C = P + g.plane_off[1 + k];
        const int pitch = g.bw[1 + k] * 8;
        const int cs = 3 * C[(long long)cy * pitch + cx] + C[(long long)fy * pitch + cx];
        const int ns = edge ? 0 : 3 * C[(long long)cy * pitch + nx] + C[(long long)fy * pitch + nx];
        v[k] = edge ? (cs * 4 + bias) >> 4 : (cs * 3 + ns + bias) >> 4;
      }
    } else {      // jinit_upsampler takes the fancy filter only for downsampled_width > 2: narrower images get plain 2 x 2 replication
#pragma unroll
      for (int k = 0; k < 2; ++k) v[k] = P[g.plane_off[1 + k] + (long long)cy * (g.bw[1 + k] * 8) + cx];
    }
    cb = v[0]; cr = v[1];
  }
  // jdcolor.c: SCALEBITS 16, FIX(x) = (int)(x * 65536 + 0.5)
  const int xb = cb - 128, xr = cr - 128;
  int R = Y + ((91881 * xr + 32768) >> 16);
  int B = Y + ((116130 * xb + 32768) >> 16);
  int G = Y + ((-22554 * xb + 32768 - 46802 * xr) >> 16);
  R = R < 0 ? 0 : (R > 255 ? 255 : R); G = G < 0 ? 0 : (G > 255 ? 255 : G); B = B < 0 ? 0 : (B > 255 ? 255 : B);
  return (uint32_t)B | ((uint32_t)G << 8) | ((uint32_t)R << 16);          // BGR, like cv2.imread
}

}  // namespace ctpn
