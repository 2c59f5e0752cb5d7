// Bandwidth-bound layers of the VGG trunk and the one-time weight repack.
//   conv_first : _get_image_blob mean subtraction (reference lib/fast_rcnn/test.py:7-11) fused with
//                conv1_1 + bias + ReLU (lib/networks/VGGnet_test.py:21, network.py:160-183). K = 27 is
//                too thin for MFMA: direct VALU conv, uint8 image in, bordered NHWC out.
//   maxpool    : Network.max_pool 2x2 stride 2 'VALID' (network.py:189-196; odd trailing row/col dropped).
//   pack       : TF variable layout -> [out][k] rows used by igemm.hip (one-time, at weight load).
#include "common.h"

namespace ctpn {

__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// ---------------------------------------------------------------------------------------------
// conv1_1 (K = 27: too thin for MFMA, direct VALU conv).
// lut[c][v] = fp32(double(v) - PIXEL_MEANS[c]) is built on the host in double, exactly as numpy's
// in-place float32 -= float64 rounds it (reference lib/fast_rcnn/test.py:8-9).
// ---------------------------------------------------------------------------------------------
// Block = 64 x 4 output pixels; thread = 4 consecutive pixels of one row x 16 output channels, so every weight
// vector fetched from LDS feeds 4 pixels and the 6 x 66 x 3 input patch (mean-subtracted through the LUT once, zero
// outside the image = TF 'SAME' padding applied AFTER mean subtraction) is staged in LDS as fp32.
constexpr int CF_TW = 64, CF_TH = 4, CF_PW = CF_TW + 2, CF_PH = CF_TH + 2;

template <typename InT, typename OutT>
__global__ __launch_bounds__(256, 2) void conv_first_kernel(const InT* __restrict__ img, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ lut,
                                                         OutT* __restrict__ out, int N, int H, int W, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float sw[27 * 64];
  __shared__ float slut[3 * 256];
  __shared__ float sin_[CF_PH * CF_PW * 3];
  const int tid = threadIdx.x;
  for (int i = tid; i < 27 * 64; i += 256) sw[i] = w[i];
  if constexpr (sizeof(InT) == 1) {
    for (int i = tid; i < 768; i += 256) slut[i] = lut[i];
    __syncthreads();
  }
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int x0 = tx * CF_TW, y0 = ty * CF_TH;
  const InT* ib = img + (long long)n * H * W * 3;
  for (int i = tid; i < CF_PH * CF_PW * 3; i += 256) {
    const int c = i % 3;
    const int px = (i / 3) % CF_PW, py = i / (3 * CF_PW);
    const int yy = y0 + py - 1, xx = x0 + px - 1;
    float v = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const InT raw = ib[((long long)yy * W + xx) * 3 + c];
      if constexpr (sizeof(InT) == 1) v = slut[c * 256 + raw];
      else v = raw;   // already mean-subtracted fp32 blob (the reference's net.data feed)
    }
    sin_[i] = v;
  }
  __syncthreads();
  const int cg = tid & 3, gq = (tid >> 2) & 15, r = tid >> 6;
  const int y = y0 + r, xb = x0 + gq * 4;
  float acc[4][16];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = bias[cg * 16 + i];
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky) {
    // 6 pixels x 3 channels of patch row (r + ky), starting at patch column gq*4
    float in[18];
    const float* src = sin_ + ((r + ky) * CF_PW + gq * 4) * 3;
#pragma unroll
    for (int i = 0; i < 18; ++i) in[i] = src[i];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float4* wr = (const float4*)(sw + ((ky * 3 + kx) * 3 + c) * 64 + cg * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 wv = wr[q];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float v = in[(p + kx) * 3 + c];
            acc[p][4 * q + 0] = fmaf(v, wv.x, acc[p][4 * q + 0]);
            acc[p][4 * q + 1] = fmaf(v, wv.y, acc[p][4 * q + 1]);
            acc[p][4 * q + 2] = fmaf(v, wv.z, acc[p][4 * q + 2]);
            acc[p][4 * q + 3] = fmaf(v, wv.w, acc[p][4 * q + 3]);
          }
        }
      }
    }
  }
  if (y >= H) return;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = xb + p;
    if (x >= W) continue;
    const long long o = (((long long)n * (H + 2) + y + 1) * (W + 2) + x + 1) * 64 + cg * 16;
    if constexpr (sizeof(OutT) == 4) {
      float4* dst = (float4*)((float*)out + o);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[q] = make_float4(fmaxf(acc[p][4 * q], 0.f), fmaxf(acc[p][4 * q + 1], 0.f), fmaxf(acc[p][4 * q + 2], 0.f), fmaxf(acc[p][4 * q + 3], 0.f));
    } else {
      uint4* dst = (uint4*)((uint16_t*)out + o);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint4 v;
        v.x = ctpn_cvt_pk_bf16(fmaxf(acc[p][8 * q + 0], 0.f), fmaxf(acc[p][8 * q + 1], 0.f));
        v.y = ctpn_cvt_pk_bf16(fmaxf(acc[p][8 * q + 2], 0.f), fmaxf(acc[p][8 * q + 3], 0.f));
        v.z = ctpn_cvt_pk_bf16(fmaxf(acc[p][8 * q + 4], 0.f), fmaxf(acc[p][8 * q + 5], 0.f));
        v.w = ctpn_cvt_pk_bf16(fmaxf(acc[p][8 * q + 6], 0.f), fmaxf(acc[p][8 * q + 7], 0.f));
        dst[q] = v;
      }
    }
  }
}

static float* g_lut_dev[16] = {nullptr};  // per device, built on first use

static int get_lut(float** out) {
  int dev = 0;
  CTPN_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail(CTPN_ERR_ARG, "device id out of range");
  if (!g_lut_dev[dev]) {
    // PIXEL_MEANS, BGR (reference lib/fast_rcnn/config.py:200)
    const double means[3] = {102.9801, 115.9465, 122.7717};
    std::vector<float> h(768);
    for (int c = 0; c < 3; ++c)
      for (int v = 0; v < 256; ++v) h[c * 256 + v] = (float)((double)v - means[c]);
    float* d = nullptr;
    CTPN_HIP_TRY(hipMalloc(&d, 768 * sizeof(float)));
    CTPN_HIP_TRY(hipMemcpy(d, h.data(), 768 * sizeof(float), hipMemcpyHostToDevice));
    g_lut_dev[dev] = d;
  }
  *out = g_lut_dev[dev];
  return CTPN_OK;
}

int launch_conv_first(const void* img, int img_is_f32, const float* w27x64, const float* bias, void* out, DType out_t, int n,
                      int h, int w, hipStream_t s) {
  float* lut = nullptr;
  int rc = get_lut(&lut);
  if (rc) return rc;
  const int tiles_x = (w + CF_TW - 1) / CF_TW, tiles_y = (h + CF_TH - 1) / CF_TH;
  const unsigned grid = (unsigned)((long long)n * tiles_x * tiles_y);
  if (img_is_f32) {
    if (out_t == DType::F32)
      hipLaunchKernelGGL((conv_first_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)img, w27x64, bias, lut, (float*)out, n, h, w, tiles_x, tiles_y);
    else
      hipLaunchKernelGGL((conv_first_kernel<float, uint16_t>), dim3(grid), dim3(256), 0, s, (const float*)img, w27x64, bias, lut, (uint16_t*)out, n, h, w, tiles_x, tiles_y);
  } else {
    if (out_t == DType::F32)
      hipLaunchKernelGGL((conv_first_kernel<uint8_t, float>), dim3(grid), dim3(256), 0, s, (const uint8_t*)img, w27x64, bias, lut, (float*)out, n, h, w, tiles_x, tiles_y);
    else
      hipLaunchKernelGGL((conv_first_kernel<uint8_t, uint16_t>), dim3(grid), dim3(256), 0, s, (const uint8_t*)img, w27x64, bias, lut, (uint16_t*)out, n, h, w, tiles_x, tiles_y);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv_first launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// max-pool 2x2/2 VALID over bordered NHWC; thread = one output pixel x one 16-byte channel chunk
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint4 max4(const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ uint4 max4<float>(const uint4& a, const uint4& b) {
  uint4 r;
  r.x = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x)));
  r.y = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y)));
  r.z = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z)));
  r.w = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w)));
  return r;
}
__device__ __forceinline__ uint32_t maxbf2(uint32_t a, uint32_t b) {
  const float alo = bf2f((uint16_t)a), blo = bf2f((uint16_t)b);
  const float ahi = bf2f((uint16_t)(a >> 16)), bhi = bf2f((uint16_t)(b >> 16));
  const uint32_t lo = (alo >= blo) ? (a & 0xffffu) : (b & 0xffffu);
  const uint32_t hi = (ahi >= bhi) ? (a & 0xffff0000u) : (b & 0xffff0000u);
  return lo | hi;
}
template <>
__device__ __forceinline__ uint4 max4<uint16_t>(const uint4& a, const uint4& b) {
  uint4 r;
  r.x = maxbf2(a.x, b.x); r.y = maxbf2(a.y, b.y); r.z = maxbf2(a.z, b.z); r.w = maxbf2(a.w, b.w);
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int chunks = C * (int)sizeof(T) / 16;
  const long long total = (long long)N * Ho * Wo * chunks;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % chunks);
  long long p = idx / chunks;
  const int xo = (int)(p % Wo); p /= Wo;
  const int yo = (int)(p % Ho);
  const int n = (int)(p / Ho);
  const long long rowp = (long long)(W + 2) * C;
  const T* src = in + (((long long)n * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1) * C;
  const uint4 a = *((const uint4*)src + ch);
  const uint4 b = *((const uint4*)(src + C) + ch);
  const uint4 c = *((const uint4*)(src + rowp) + ch);
  const uint4 d = *((const uint4*)(src + rowp + C) + ch);
  const uint4 m = max4<T>(max4<T>(a, b), max4<T>(c, d));
  T* dst = out + (((long long)n * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * C;
  *((uint4*)dst + ch) = m;
}

int launch_maxpool(const void* in, void* out, DType t, int n, int h, int w, int c, hipStream_t s) {
  const int es = (t == DType::F32) ? 4 : 2;
  if ((c * es) % 16 != 0) return fail(CTPN_ERR_ARG, "maxpool: channel bytes must be a multiple of 16");
  const long long total = (long long)n * (h / 2) * (w / 2) * (c * es / 16);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (t == DType::F32)
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)in, (float*)out, n, h, w, c);
  else
    hipLaunchKernelGGL(maxpool_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (const uint16_t*)in, (uint16_t*)out, n, h, w, c);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("maxpool launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// pack: dst[c][r] = src[r][c]  (fp32 -> fp32 | bf16), 32x32 tiles through LDS
// ---------------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(256) void pack_transpose_kernel(const float* __restrict__ src, long long src_ld, OutT* __restrict__ dst,
                                                             long long dst_ld, int rows, int cols) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * src_ld + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) {
      const float v = tile[tx][i];
      if constexpr (sizeof(OutT) == 4) dst[(long long)c * dst_ld + r] = v;
      else dst[(long long)c * dst_ld + r] = f2bf(v);
    }
  }
}

__global__ void cvt_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int n, int hw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (2 * i + 1 >= n + 1) return;
  const float a = in[2 * i], b = (2 * i + 1 < n) ? in[2 * i + 1] : 0.f;
  unsigned int r = hw ? ctpn_cvt_pk_bf16(a, b) : ((unsigned int)f2bf(a) | ((unsigned int)f2bf(b) << 16));
  out[2 * i] = (uint16_t)r;
  if (2 * i + 1 < n) out[2 * i + 1] = (uint16_t)(r >> 16);
}
int launch_cvt_bf16(const float* in, uint16_t* out, int n, int hw, hipStream_t s) {
  hipLaunchKernelGGL(cvt_bf16_kernel, dim3((n / 2 + 256) / 256), dim3(256), 0, s, in, out, n, hw);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("cvt launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

int launch_pack_transpose(const float* src, long long src_ld, void* dst, long long dst_ld, DType dst_t, int rows,
                          int cols, hipStream_t s) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  if (dst_t == DType::F32)
    hipLaunchKernelGGL(pack_transpose_kernel<float>, grid, dim3(256), 0, s, src, src_ld, (float*)dst, dst_ld, rows, cols);
  else
    hipLaunchKernelGGL(pack_transpose_kernel<uint16_t>, grid, dim3(256), 0, s, src, src_ld, (uint16_t*)dst, dst_ld, rows, cols);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("pack launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
