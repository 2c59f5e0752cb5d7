// 3 x 3 convolution + bias + ReLU in bf16 through the 1-D Winograd transform F(2, 3) along x -- CORRECTNESS-FIRST reference kernel of a
// device mode that is not on the product path yet. It exists so that the arithmetic of oracle/winograd.py
// (conv3x3_relu_winograd_x) is pinned on the GPU, bit-compatible rounding points included, before the persistent conv kernel grows the
// mode: reachable only through ctpn_debug_conv3x3(impl = 2). No LDS, no pipelining: every MFMA operand comes straight from global memory.
//
// Replaces, when it is built out: tf.nn.conv2d + bias_add + relu of Network.conv (reference lib/networks/network.py:160-183).
//
//   U[co][(ky * 4 + f) * Ci + ci] = bf16( G[f][:] . w[ky][:][ci][co] )         (pack kernel: double -> float -> bf16, as the oracle)
//   V[r][t][f][ci]                = bf16( B^T[f][:] . d[r][2t .. 2t + 3][ci] )   (sum / difference of two bf16 values: exact in fp32)
//   m_f[y][t][co]                 = sum_ky sum_ci V[y + ky][t][f][ci] U[co][ky][f][ci]          (v_mfma_f32_32x32x16_bf16, fp32)
//   out[y][2t] = relu(m0 + m1 + m2 + b),  out[y][2t + 1] = relu(m1 - m2 - m3 + b)               -> bf16
//
// 12 Ci multiplies per output pair instead of 18. Activations: bordered NHWC bf16 like everywhere else; position t of output row y reads
// bordered columns 2t .. 2t + 3 of bordered rows y .. y + 2 -- for odd W the last position reads one pixel past its row, which is the
// next row's left border (zero) or the buffer's slack.
#include "common.h"

namespace ctpn {

typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(16))) float wg_f32x16;

// G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
__global__ __launch_bounds__(256) void winograd_x_pack_kernel(const float* __restrict__ w_hwio, uint16_t* __restrict__ u, int Ci, int Co, int co_pad) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;                  // over co_pad * 12 * Ci
  const long long total = (long long)co_pad * 12 * Ci;
  if (idx >= total) return;
  const int ci = (int)(idx % Ci);
  const int kf = (int)((idx / Ci) % 12), ky = kf >> 2, f = kf & 3;
  const int co = (int)(idx / ((long long)12 * Ci));
  float v = 0.f;
  if (co < Co) {
    const double g0 = w_hwio[((size_t)(ky * 3 + 0) * Ci + ci) * Co + co], g1 = w_hwio[((size_t)(ky * 3 + 1) * Ci + ci) * Co + co],
                 g2 = w_hwio[((size_t)(ky * 3 + 2) * Ci + ci) * Co + co];
    const double d = f == 0 ? g0 : f == 1 ? 0.5 * g0 + 0.5 * g1 + 0.5 * g2 : f == 2 ? 0.5 * g0 - 0.5 * g1 + 0.5 * g2 : g2;
    v = (float)d;
  }
  u[idx] = (uint16_t)(ctpn_cvt_pk_bf16(v, 0.f) & 0xffffu);
}

__device__ __forceinline__ float wg_lo(uint32_t p) { return ctpn_bf16_to_f32((unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float wg_hi(uint32_t p) { return ctpn_bf16_to_f32((unsigned short)(p >> 16)); }
// elementwise a + s * b on 8 packed bf16 values, fp32 arithmetic (exact for two bf16 operands), result RNE to bf16
__device__ __forceinline__ uint4 wg_axpb(const uint4& a, const uint4& b, float s) {
  uint4 r;
  r.x = ctpn_cvt_pk_bf16(wg_lo(a.x) + s * wg_lo(b.x), wg_hi(a.x) + s * wg_hi(b.x));
  r.y = ctpn_cvt_pk_bf16(wg_lo(a.y) + s * wg_lo(b.y), wg_hi(a.y) + s * wg_hi(b.y));
  r.z = ctpn_cvt_pk_bf16(wg_lo(a.z) + s * wg_lo(b.z), wg_hi(a.z) + s * wg_hi(b.z));
  r.w = ctpn_cvt_pk_bf16(wg_lo(a.w) + s * wg_lo(b.w), wg_hi(a.w) + s * wg_hi(b.w));
  return r;
}

// grid (ceil(tw / 32) * ceil(Co / 128), H, N); 4 waves: wave w owns channels cb * 128 + 32 w .. + 31 of 32 positions of one output row
__global__ __launch_bounds__(256) void winograd_x_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ u, const float* __restrict__ bias,
                                                         uint16_t* __restrict__ out, int H, int W, int Ci, int Co, int tblocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tb = blockIdx.x % tblocks, cb = blockIdx.x / tblocks;
  const int y = blockIdx.y, n = blockIdx.z;
  const int Hp = H + 2, Wp = W + 2, tw = (W + 1) >> 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int p_raw = tb * 32 + l31, p = p_raw < tw ? p_raw : tw - 1;
  const int co_a = cb * 128 + wave * 32 + l31;                                      // the channel whose U rows this lane feeds as the A operand
  const uint16_t* urow = u + (size_t)co_a * 12 * Ci + 8 * kh;
  wg_f32x16 acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const uint16_t* drow = in + (((size_t)n * Hp + y + ky) * Wp + 2 * p) * Ci + 8 * kh;
    for (int c0 = 0; c0 < Ci; c0 += 16) {
      const uint4 d0 = *(const uint4*)(drow + c0), d1 = *(const uint4*)(drow + Ci + c0), d2 = *(const uint4*)(drow + 2 * Ci + c0),
                  d3 = *(const uint4*)(drow + 3 * Ci + c0);
      uint4 v[4];
      v[0] = wg_axpb(d0, d2, -1.f);        // B^T = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
      v[1] = wg_axpb(d1, d2, 1.f);
      v[2] = wg_axpb(d2, d1, -1.f);
      v[3] = wg_axpb(d1, d3, -1.f);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const uint4 uf = *(const uint4*)(urow + (size_t)(ky * 4 + f) * Ci + c0);
        acc[f] = HalfOps<h_bf16>::mfma_32x32x16(uf, v[f], acc[f]);
      }
    }
  }
  // lane: position l31 (column of the MFMA result), channels 8 g4 + 4 kh + e of the wave's 32 (rows)
  if (p_raw >= tw) return;
  const int x0 = 2 * p_raw;
  uint16_t* o = out + (((size_t)n * Hp + y + 1) * Wp + x0 + 1) * Co;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = cb * 128 + wave * 32 + 8 * g4 + 4 * kh + e;
      if (co >= Co) continue;
      const int i = 4 * g4 + e;
      const float b = bias[co];
      const float o0 = fmaxf(((acc[0][i] + acc[1][i]) + acc[2][i]) + b, 0.f);
      const float o1 = fmaxf(((acc[1][i] - acc[2][i]) - acc[3][i]) + b, 0.f);
      o[co] = (uint16_t)(ctpn_cvt_pk_bf16(o0, 0.f) & 0xffffu);
      if (x0 + 1 < W) o[Co + co] = (uint16_t)(ctpn_cvt_pk_bf16(o1, 0.f) & 0xffffu);
    }
}

// in / out: bordered NHWC bf16 (the input with the usual slack behind it); w_hwio: fp32 [3][3][Ci][Co] on the device; bf16 only
int launch_conv3x3_winograd_x(const void* in, const float* w_hwio, const float* bias, void* out, int n, int h, int w, int ci, int co, hipStream_t s) {
  if (!in || !w_hwio || !bias || !out) return fail(CTPN_ERR_ARG, "winograd_x: null pointer");
  if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || ci % 16 != 0) return fail(CTPN_ERR_ARG, "winograd_x: Ci must be a positive multiple of 16");
  const int co_pad = (co + 127) / 128 * 128;
  uint16_t* u = nullptr;
  const size_t ucount = (size_t)co_pad * 12 * ci;
  CTPN_HIP_TRY(hipMalloc((void**)&u, ucount * 2));
  hipLaunchKernelGGL(winograd_x_pack_kernel, dim3((unsigned)((ucount + 255) / 256)), dim3(256), 0, s, w_hwio, u, ci, co, co_pad);
  const int tw = (w + 1) / 2, tblocks = (tw + 31) / 32, cblocks = co_pad / 128;
  hipLaunchKernelGGL(winograd_x_kernel, dim3((unsigned)(tblocks * cblocks), (unsigned)h, (unsigned)n), dim3(256), 0, s, (const uint16_t*)in, u, bias,
                     (uint16_t*)out, h, w, ci, co, tblocks);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(u);
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("winograd_x launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
