// Input pipeline on device: cv2.resize(..., fx, fy, INTER_LINEAR) of the two places the reference
// rescales an image -- ctpn/demo.py:21-25 (resize_im, uint8 BGR) and lib/fast_rcnn/test.py:17-27 (_get_image_blob, float32
// after the mean subtraction). OpenCV (opencv_python==3.4.0.12, requirements.txt) is not in the reference tree, so this
// follows its published algorithm (modules/imgproc/src/resize.cpp, 3.4 branch) -- PARITY UNPINNED against the real cv2:
//   dsize = cvRound(src * f) (round half to even); sample position fx = (float)((dx + 0.5) / f - 0.5), sx = floor(fx);
//   sx < 0 -> sx = 0, fx = 0; sx >= w - 1 -> sx = w - 1, fx = 0 (columns only; rows are clamped instead);
//   uint8: 11-bit fixed-point weights a = cvRound(w * 2048) as short, horizontal sums in int,
//          dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
//   float: S = s0 * a0 + s1 * a1 per row, dst = S0 * b0 + S1 * b1, all in fp32 without contraction.
// One thread per output pixel (3 channels); HBM-bound and tiny next to the convolutions.
#include <cmath>

#include "common.h"

namespace ctpn {

__device__ __forceinline__ void rs_coord(int d, double inv_f, int n, int clamp_w, int& s, float& f) {
  f = (float)(((double)d + 0.5) * inv_f - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
  if (clamp_w) {
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n - 1) { s = n - 1; f = 0.f; }
  }
}

__device__ __forceinline__ int rs_short(float v) {   // saturate_cast<short>(float): round half to even, saturate
  const int r = (int)rintf(v);
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

template <typename T>
__global__ void resize_linear_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int h, int w, int dh, int dw, double inv_fx,
                                     double inv_fy) {
  const long long total = (long long)n * dh * dw;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
    const int dx = (int)(p % dw);
    const int dy = (int)((p / dw) % dh);
    const int img = (int)(p / ((long long)dw * dh));
    int sx, sy;
    float fx, fy;
    rs_coord(dx, inv_fx, w, 1, sx, fx);
    rs_coord(dy, inv_fy, h, 0, sy, fy);
    const int x1 = sx + 1 < w ? sx + 1 : w - 1;
    const int y0 = sy < 0 ? 0 : (sy < h ? sy : h - 1);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 < h ? sy + 1 : h - 1);
    const T* r0 = src + ((long long)img * h + y0) * w * 3;
    const T* r1 = src + ((long long)img * h + y1) * w * 3;
    T* o = dst + p * 3;
    if constexpr (sizeof(T) == 1) {
      const int a0 = rs_short((1.f - fx) * 2048.f), a1 = rs_short(fx * 2048.f);
      const int b0 = rs_short((1.f - fy) * 2048.f), b1 = rs_short(fy * 2048.f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = (int)r0[sx * 3 + c] * a0 + (int)r0[x1 * 3 + c] * a1;
        const int S1 = (int)r1[sx * 3 + c] * a0 + (int)r1[x1 * 3 + c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (T)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    } else {
      const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float S0 = __fadd_rn(__fmul_rn((float)r0[sx * 3 + c], a0), __fmul_rn((float)r0[x1 * 3 + c], a1));
        const float S1 = __fadd_rn(__fmul_rn((float)r1[sx * 3 + c], a0), __fmul_rn((float)r1[x1 * 3 + c], a1));
        o[c] = (T)__fadd_rn(__fmul_rn(S0, b0), __fmul_rn(S1, b1));
      }
    }
  }
}

// cvRound: round half to even
int resize_out_dim(int src, double f) { return (int)std::nearbyint((double)src * f); }

int launch_resize_linear(const void* src, void* dst, int is_f32, int n, int h, int w, int dh, int dw, double fx, double fy, hipStream_t s) {
  if (n <= 0 || h <= 0 || w <= 0 || dh <= 0 || dw <= 0 || !(fx > 0.0) || !(fy > 0.0)) return fail(CTPN_ERR_ARG, "resize: bad geometry");
  const long long total = (long long)n * dh * dw;
  const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 65536);
  if (is_f32)
    hipLaunchKernelGGL(resize_linear_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n, h, w, dh, dw, 1.0 / fx, 1.0 / fy);
  else
    hipLaunchKernelGGL(resize_linear_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (const uint8_t*)src, (uint8_t*)dst, n, h, w, dh, dw, 1.0 / fx, 1.0 / fy);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("resize launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
