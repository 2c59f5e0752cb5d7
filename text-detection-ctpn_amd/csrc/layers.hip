// The network's first layer and the one-time weight repack.
//   conv_first : _get_image_blob mean subtraction (reference lib/fast_rcnn/test.py:7-11) fused with
//                conv1_1 + bias + ReLU (lib/networks/VGGnet_test.py:21, network.py:160-183), three forms:
//                  conv_first_kernel        fp32 mode: direct VALU conv, uint8 or float image in, bordered NHWC out
//                  conv_first_mfma_kernel   split precision, and the float-blob feed of the 16-bit modes: split-bf16 operands on the MFMAs
//                  image_to_q_kernel (+ conv_first_p_kernel)   uint8 feed of the 16-bit modes: bytes -> q-image (common.h); conv1_1 itself is
//                                           computed inside conv1_2's window stage (conv3x3_impl.h) and stored only for keep_acts
//   pack       : TF variable layout -> [out][k] rows used by the conv / GEMM kernels, conv1_1's MFMA fragments (one-time, at weight load).
//   (the 2x2 max-pools are fused into the conv epilogues, conv3x3_impl.h)
#include <cstring>
#include <type_traits>

#include <mutex>

#include "common.h"

namespace ctpn {

__device__ __forceinline__ uint16_t f2bf(float f) { return ctpn_f32_to_bf16(f); }

// ---------------------------------------------------------------------------------------------
// conv1_1 (K = 27: too thin for MFMA, direct VALU conv).
// lut[c][v] = fp32(double(v) - PIXEL_MEANS[c]) is built on the host in double, exactly as numpy's
// in-place float32 -= float64 rounds it (reference lib/fast_rcnn/test.py:8-9).
// ---------------------------------------------------------------------------------------------
// Block = 64 x 4 output pixels; thread = 4 consecutive pixels of one row x 16 output channels, so every weight
// vector fetched from LDS feeds 4 pixels and the 6 x 66 x 3 input patch (mean-subtracted through the LUT once, zero
// outside the image = TF 'SAME' padding applied AFTER mean subtraction) is staged in LDS as fp32.
constexpr int CF_TW = 64, CF_TH = 4, CF_PW = CF_TW + 2, CF_PH = CF_TH + 2;

// OutT: float | h_bf16 | h_f16
template <typename InT, typename OutT>
__global__ __launch_bounds__(256, 2) void conv_first_kernel(const InT* __restrict__ img, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ lut,
                                                         void* __restrict__ out, int N, int H, int W, int tiles_x, int tiles_y) {
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
    if constexpr (std::is_same<OutT, float>::value) {
      float4* dst = (float4*)((float*)out + o);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[q] = make_float4(fmaxf(acc[p][4 * q], 0.f), fmaxf(acc[p][4 * q + 1], 0.f), fmaxf(acc[p][4 * q + 2], 0.f), fmaxf(acc[p][4 * q + 3], 0.f));
    } else {
      uint4* dst = (uint4*)((uint16_t*)out + o);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint4 v;
        v.x = HalfOps<OutT>::cvt_pk(fmaxf(acc[p][8 * q + 0], 0.f), fmaxf(acc[p][8 * q + 1], 0.f));
        v.y = HalfOps<OutT>::cvt_pk(fmaxf(acc[p][8 * q + 2], 0.f), fmaxf(acc[p][8 * q + 3], 0.f));
        v.z = HalfOps<OutT>::cvt_pk(fmaxf(acc[p][8 * q + 4], 0.f), fmaxf(acc[p][8 * q + 5], 0.f));
        v.w = HalfOps<OutT>::cvt_pk(fmaxf(acc[p][8 * q + 6], 0.f), fmaxf(acc[p][8 * q + 7], 0.f));
        dst[q] = v;
      }
    }
  }
}

static uint16_t host_rne_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float host_bf16_to_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
static uint16_t host_rne_f16(float f) { const _Float16 h = (_Float16)f; uint16_t b; std::memcpy(&b, &h, 2); return b; }
static float host_f16_to_f(uint16_t b) { _Float16 h; std::memcpy(&h, &b, 2); return (float)h; }

// ---------------------------------------------------------------------------------------------
// conv1_1 on the matrix cores for the bf16 path, at fp32-class accuracy: every fp32 operand is split into two bf16
// terms (x = hi + lo to ~2^-16 relative, same for the weights) and  W*X ~= Whi*Xhi + Whi*Xlo + Wlo*Xhi  is three
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped lo*lo term is ~2^-16 of a product). The VALU kernel
// spends 1728 FMAs per pixel (1.09 ms per 32-image batch); this one is bound by its 2.2 GB write.
//
// im2col without a gather: the 27 inputs of a pixel are three runs of 9 consecutive patch elements (one per ky), so
// one K step of 16 per ky (m = 3 kx + c in slots 0..8, zero weights in 10..15, the bias in slot 9 of ky 0 against a
// constant 1.0) makes every lane's 8-element fragment 8 CONSECUTIVE bf16 of the LDS patch. The patch is kept as bf16
// planes (hi, lo), each in two copies one element apart, so that a run starting at an odd element is still a 4-byte
// aligned ds_read2_b32 pair; the zero-weighted slots read whatever finite data follows the run.
//
// One workgroup (4 waves) per 4 x 64 tile. Every global load of the workgroup is issued in ONE round before the first
// barrier (raw image dwords into registers, LUT into LDS, weight fragments into registers) and all address logic is
// branch-free: a branch per load makes hipcc wait for each load in turn (measured: 5 serialised byte-load rounds were
// 0.4 ms of a 0.86 ms kernel). A persistent variant with a dedicated loader wave (tile t+1 expanded while tile t is on
// the MFMAs) was built and measured SLOWER (1.11 ms): the expansion through the LUT is a serial LDS-latency chain in one
// wave, and on gfx9 loads and stores retire through one in-order counter, so a wave that prefetches also waits for the
// acknowledgement of its earlier stores. Weight fragments [co tile][ky][hi|lo][lane] are packed once at weight load.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 cf_bf16x8;
typedef __attribute__((ext_vector_type(16))) float cf_f32x16;

constexpr int CF_ROW_B = CF_PW * 3;                 // 198 bytes (u8) / elements per patch row
constexpr int CF_ROW_DW = (CF_ROW_B + 3 + 3) / 4;   // aligned dwords that cover a row at any byte alignment: 51
constexpr int CF_NEL = CF_PH * CF_ROW_B;            // 1188 elements per patch
constexpr int CF_PLANE = CF_NEL + 56;               // u16 per plane copy: zero tail for the 15-element over-read of the last run; (CF_PLANE / 2) % 32 == 14
                                                    // puts the odd lanes' dwords (copy B) on the 16 banks the even lanes' (copy A) leave free
constexpr int CF_BUF = 4 * CF_PLANE;                // [hi A | hi B | lo A | lo B]; copy B holds element i at index i + 1
constexpr int CF_DUMMY = CF_PLANE - 2;              // never read: target of bytes that belong to no patch position
static_assert(CF_PLANE % 2 == 0 && CF_ROW_B % 2 == 0, "run parity must be a per-lane constant");
static_assert((CF_PLANE / 2) % 32 == 14, "copy B must start 14 banks after copy A");

// One tile per workgroup: 2 / 4 / 8 tiles per workgroup (LUT + weight fragments staged once, next tile's dwords prefetched) measured
// SLOWER in round 2, 0.82 - 0.84 vs 0.665 ms -- many short independent workgroups hide the load -> LUT expansion -> MFMA -> store
// chain better than a loop inside one; the variant was removed in round 3.
// OM: how the fp32-class sums are stored -- 0: bf16, 1: fp16 (64 channels per pixel), 2: split precision, [hi(64) | lo(64)] bf16 planes
template <typename InT, int OM>
__global__ __launch_bounds__(256, 5) void conv_first_mfma_kernel(const InT* __restrict__ img, const uint4* __restrict__ wfrag,
                                                                 const float* __restrict__ lut, uint16_t* __restrict__ out,
                                                                 int N, int H, int W, int tiles_x, int tiles_y) {
  constexpr int OPITCH = OM == 2 ? 128 : 64;
  constexpr bool U8 = sizeof(InT) == 1;
  constexpr int NREG = U8 ? (CF_PH * CF_ROW_DW + 255) / 256 : (CF_NEL + 255) / 256;   // 2 dwords / 5 floats per thread
  __shared__ __attribute__((aligned(16))) uint16_t buf[CF_BUF];
  __shared__ uint32_t slut[U8 ? 768 : 1];
  __shared__ uint4 swf[12 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, fhalf = lane >> 5;
  const int ntiles = N * tiles_x * tiles_y;
  const unsigned long long ibase = (unsigned long long)img;
  const unsigned long long iend = ibase + (unsigned long long)N * H * W * 3 * sizeof(InT);
  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    n = t / tiles_y; y0 = ty * CF_TH; x0 = tx * CF_TW;
  };

  // ---- one round of global loads per tile ----
  // u8: element e = tid + 256 k is dword d of patch row `row`: the 4-byte ALIGNED dword (absolute address) at or below the
  // row's first byte + 4 d. A dword that holds at least one image byte lies in a mapped page; one that holds none (before
  // the first / after the last image byte, a row outside the image) is redirected to the image's first dword and masked
  // below. float: element e is one value of the patch.
  auto load_raw = [&](int t, uint32_t (&raw)[NREG]) {
    int n, y0, x0;
    tile_origin(t < ntiles ? t : ntiles - 1, n, y0, x0);
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int e = tid + 256 * k;
      if constexpr (U8) {
        const int row = e / CF_ROW_DW, d = e - row * CF_ROW_DW;
        const int yy = y0 + row - 1;
        const long long rs = (((long long)n * H + yy) * W + (x0 - 1)) * 3;        // byte offset of the patch row (may be < 0)
        const unsigned long long a = ((ibase + (unsigned long long)rs) & ~3ull) + 4ull * d;
        const bool ok = row < CF_PH && yy >= 0 && yy < H && a + 4 > ibase && a < iend;
        raw[k] = *(const uint32_t*)(ok ? a : (ibase & ~3ull));
      } else {
        const int row = e / CF_ROW_B, pos = e - row * CF_ROW_B;
        const int yy = y0 + row - 1, xx = x0 - 1 + pos / 3;
        const bool ok = row < CF_PH && yy >= 0 && yy < H && xx >= 0 && xx < W;
        raw[k] = __builtin_bit_cast(uint32_t, (float)img[ok ? (((long long)n * H + yy) * W + xx) * 3 + pos % 3 : 0]);
      }
    }
  };
  constexpr int TPW = 1;
  const int t_first = blockIdx.x * TPW;
  uint32_t raw[NREG];
  load_raw(t_first, raw);
  uint32_t lutv[3];
  if constexpr (U8) {
#pragma unroll
    for (int k = 0; k < 3; ++k) lutv[k] = __builtin_bit_cast(uint32_t, lut[768 + tid + 256 * k]);
  }
  // weight fragments [i = co tile][ky][part: 0 hi, 1 lo][lane] through LDS (12 KB): 48 VGPRs less, six workgroups per CU
  uint4 wload[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) wload[q] = wfrag[tid + 256 * q];
  if constexpr (U8) {
#pragma unroll
    for (int k = 0; k < 3; ++k) slut[tid + 256 * k] = lutv[k];
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) swf[tid + 256 * q] = wload[q];
  constexpr int TAIL_DW = (CF_PLANE - CF_NEL) / 2;
  if (tid < 4 * TAIL_DW) ((uint32_t*)buf)[(tid / TAIL_DW) * (CF_PLANE / 2) + CF_NEL / 2 + tid % TAIL_DW] = 0u;   // the four zero tails
  __syncthreads();

#pragma unroll 1
  for (int tt = 0; tt < TPW; ++tt) {
  const int t_cur = t_first + tt;
  if (t_cur >= ntiles) break;
  int n, y0, x0;
  tile_origin(t_cur, n, y0, x0);
  if (tt > 0) __syncthreads();                    // the previous tile's MFMA reads of `buf` are done
  // ---- registers -> bf16 planes; everything outside the image becomes 0 (SAME padding). Branch-free as well ----
  {
    const int pos_lo = x0 == 0 ? 3 : 0;                                        // patch-row positions that lie inside the image
    const int pos_hi = (W - x0 + 1) * 3 < CF_ROW_B ? (W - x0 + 1) * 3 : CF_ROW_B;
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int e = tid + 256 * k;
      if constexpr (U8) {
        const int row = e / CF_ROW_DW, d = e - row * CF_ROW_DW;
        const int yy = y0 + row - 1;
        const bool rowok = row < CF_PH && yy >= 0 && yy < H;
        const long long rs = (((long long)n * H + yy) * W + (x0 - 1)) * 3;
        const int pos0 = 4 * d - (int)((ibase + (unsigned long long)rs) & 3ull);   // position of the dword's first byte (-3 .. 203)
        const int c0 = (pos0 + 3) % 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pos = pos0 + q;
          const int c = (c0 + q) % 3;
          const bool inrow = row < CF_PH && pos >= 0 && pos < CF_ROW_B;
          const uint32_t keep = (rowok && pos >= pos_lo && pos < pos_hi) ? 0xffffffffu : 0u;   // a mask, not a select: the LUT read stays unconditional
          const uint32_t v = slut[c * 256 + ((raw[k] >> (8 * q)) & 0xffu)] & keep;
          const int i = inrow ? row * CF_ROW_B + pos : CF_DUMMY;
          buf[i] = (uint16_t)v;
          buf[CF_PLANE + i + 1] = (uint16_t)v;
          buf[2 * CF_PLANE + i] = (uint16_t)(v >> 16);
          buf[3 * CF_PLANE + i + 1] = (uint16_t)(v >> 16);
        }
      } else {
        const int row = e / CF_ROW_B, pos = e - row * CF_ROW_B;
        const int yy = y0 + row - 1;
        const bool ok = row < CF_PH && yy >= 0 && yy < H && pos >= pos_lo && pos < pos_hi;
        const float f = ok ? __builtin_bit_cast(float, raw[k]) : 0.f;
        const uint32_t hi = ctpn_cvt_pk_bf16(f, 0.f) & 0xffffu;
        const uint32_t lo = ctpn_cvt_pk_bf16(f - __builtin_bit_cast(float, hi << 16), 0.f) & 0xffffu;
        const int i = row < CF_PH ? e : CF_DUMMY;
        buf[i] = (uint16_t)hi;
        buf[CF_PLANE + i + 1] = (uint16_t)hi;
        buf[2 * CF_PLANE + i] = (uint16_t)lo;
        buf[3 * CF_PLANE + i + 1] = (uint16_t)lo;
      }
    }
  }
  if (TPW > 1 && tt + 1 < TPW) load_raw(t_cur + 1, raw);      // next tile's dwords fly under this tile's MFMAs and stores
  __syncthreads();

  // ---- MFMA: wave = tile row, two 32-pixel groups per wave ----
  // a lane's runs start at element s = (wave + ky) * 198 + 3 * (32 pt + l31) + 8 * fhalf: parity = parity of l31
  const int par = l31 & 1;
  const int s0 = wave * CF_ROW_B + 3 * l31 + 8 * fhalf;
  const uint32_t* p32 = (const uint32_t*)buf + ((par * CF_PLANE + s0 + par) >> 1);   // hi planes; lo planes: + CF_PLANE dwords
  const uint32_t keep0 = fhalf ? 0x0000ffffu : 0xffffffffu;         // slot 9 of ky 0 (lanes 32..63, element 1): constant 1.0 against the bias row
  const uint32_t one0 = fhalf ? 0x3f800000u : 0u;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    uint4 xhi[3], xlo[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int o = pt * 48 + ky * (CF_ROW_B / 2);
      xhi[ky] = make_uint4(p32[o], p32[o + 1], p32[o + 2], p32[o + 3]);
      xlo[ky] = make_uint4(p32[CF_PLANE + o], p32[CF_PLANE + o + 1], p32[CF_PLANE + o + 2], p32[CF_PLANE + o + 3]);
    }
    xhi[0].x = (xhi[0].x & keep0) | one0;
    xlo[0].x &= keep0;
    // store addressing (see conv3x3_p_kernel's store_pair): after the 32-lane exchange a lane holds two complete 16-byte pieces of ITS pixel,
    // v_permlane16_swap then trades piece 1 of lanes r with piece 0 of lanes r + 16: store A carries pixels 0..15 of the group, store B
    // pixels 16..31, FOUR lanes = 64 contiguous bytes per pixel -- a store instruction touches 16 lines instead of 32 (round 6: the split
    // mode's conv1_1 writes 4.4 GB per 32 images and was store-issue bound at 3.1 TB/s with 32-byte runs). Lane L stores piece
    // 2 ((L >> 4) & 1) + (L >> 5) of pixel (L & 15) [A] / 16 + (L & 15) [B].
    const int y = y0 + wave;
    const int xA = x0 + pt * 32 + (lane & 15), xB = xA + 16;
    uint16_t* opA = out + (((long long)n * (H + 2) + y + 1) * (W + 2) + xA + 1) * OPITCH + (2 * ((lane >> 4) & 1) + fhalf) * 8;
    uint16_t* opB = opA + 16 * OPITCH;
    const bool inA = y < H && xA < W, inB = y < H && xB < W;
    cf_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = cf_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int i = 0; i < 2; ++i)      // two independent accumulator chains
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cf_bf16x8, swf[((i * 3 + ky) * 2 + (term == 0 ? 1 : 0)) * 64 + lane]),
                                                           __builtin_bit_cast(cf_bf16x8, term == 1 ? xlo[ky] : xhi[ky]), acc[i], 0, 0, 0);
    // lanes l and l+32 hold channels 4*fhalf..+3 of each 8-channel group of the same pixel: v_permlane32_swap gives the low
    // half the whole even group and the high half the whole odd group. ReLU on the packed pair: a negative bf16 is a negative int16, so
    // max(x, 0) as int16 is exactly ReLU (-0 -> +0)
    typedef uint32_t cf_u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      cf_u32x4 vh[2], vl[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t pk[4], pl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (OM == 2) {
            ctpn_split_pk_bf16(fmaxf(acc[i][8 * q + 2 * j], 0.f), fmaxf(acc[i][8 * q + 2 * j + 1], 0.f), pk[j], pl[j]);      // ReLU in fp32, then (hi, lo)
          } else {
            const uint32_t u = OM == 1 ? ctpn_cvt_pk_f16(acc[i][8 * q + 2 * j], acc[i][8 * q + 2 * j + 1]) : ctpn_cvt_pk_bf16(acc[i][8 * q + 2 * j], acc[i][8 * q + 2 * j + 1]);
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            pk[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), s16x2{0, 0}));
          }
        }
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
        vh[q] = cf_u32x4{r0[0], r1[0], r0[1], r1[1]};
        if constexpr (OM == 2) {
          const auto s0_ = __builtin_amdgcn_permlane32_swap(pl[0], pl[2], false, false);
          const auto s1_ = __builtin_amdgcn_permlane32_swap(pl[1], pl[3], false, false);
          vl[q] = cf_u32x4{s0_[0], s1_[0], s0_[1], s1_[1]};
        }
      }
      cf_u32x4 a, b;
#pragma unroll
      for (int c = 0; c < 4; ++c) { const auto r = __builtin_amdgcn_permlane16_swap(vh[0][c], vh[1][c], false, false); a[c] = r[0]; b[c] = r[1]; }
      if (inA) *(cf_u32x4*)(opA + i * 32) = a;
      if (inB) *(cf_u32x4*)(opB + i * 32) = b;
      if constexpr (OM == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const auto r = __builtin_amdgcn_permlane16_swap(vl[0][c], vl[1][c], false, false); a[c] = r[0]; b[c] = r[1]; }
        if (inA) *(cf_u32x4*)(opA + 64 + i * 32) = a;
        if (inB) *(cf_u32x4*)(opB + 64 + i * 32) = b;
      }
    }
  }
  }  // tiles of this workgroup
}

// w27x64 / bias (device, fp32 [27][64] = HWIO flattened, [64]) -> MFMA A fragments [(i*3+ky)*2+part][64 lanes] of 8 bf16:
// lane (r = lane & 31, h = lane >> 5), element j is K slot m = 8 h + j of row ky: m < 9 the tap (ky, kx = m / 3, c = m % 3),
// m == 9 of ky 0 the bias (its data slot is the constant 1.0), everything else 0
int pack_conv1_frags(const float* w27x64_dev, const float* bias_dev, uint4* frags_dev) {
  std::vector<float> w(27 * 64), bv(64);
  CTPN_HIP_TRY(hipMemcpy(w.data(), w27x64_dev, w.size() * 4, hipMemcpyDeviceToHost));
  CTPN_HIP_TRY(hipMemcpy(bv.data(), bias_dev, bv.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint16_t> f((size_t)CF_FRAG_BYTES / 2, 0);
  for (int i = 0; i < 2; ++i)
    for (int ky = 0; ky < 3; ++ky)
      for (int ln = 0; ln < 64; ++ln) {
        const int r = ln & 31, h = ln >> 5;
        for (int j = 0; j < 8; ++j) {
          const int m = 8 * h + j;
          const float v = m < 9 ? w[(size_t)(ky * 9 + m) * 64 + i * 32 + r] : (m == 9 && ky == 0 ? bv[i * 32 + r] : 0.f);
          const uint16_t hi = host_rne_bf16(v), lo = host_rne_bf16(v - host_bf16_to_f(hi));
          f[((((size_t)(i * 3 + ky) * 2 + 0) * 64 + ln) * 8) + j] = hi;
          f[((((size_t)(i * 3 + ky) * 2 + 1) * 64 + ln) * 8) + j] = lo;
        }
      }
  const double dmean[3] = {103.0 - 102.9801, 116.0 - 115.9465, 123.0 - 122.7717};       // round(mean) - mean, BGR
  // conv1_1 over the q-image (conv_first_p_kernel, the producer inside conv3x3_wr_kernel): fragments [(i*3+ky)][64 lanes] x 8 halves behind the
  // split ones, a bf16 set and an fp16 set. The pixel side is EXACT in either type instead of split: p - mean_c = (p - m_c) + (m_c - mean_c)
  // with m = round(mean) = (103, 116, 123); q = p - m_c is an integer in [-123, 152]. The constant d_c = m_c - mean_c (|d| < 0.23) goes to the
  // weight side as G (below). What remains inexact is the 16-bit rounding of the 27 weights, as in every other layer of the 16-bit modes. K slot m = 8 h + j of row ky (h = lane half); the data side is two OVERLAPPING 16-byte
  // reads of the q-image row -- lanes 0..31 pixels (x - 1, x), lanes 32..63 pixels (x, x + 1) -- so pixel x appears twice:
  //   0..2   w[ky][0][c]        3   G[ky][0]                       4..6   w[ky][1][c]     7   ky == 1 ? V hi : G[ky][1]
  //   8..10  0                  11  ky == 1 ? V lo : 0             12..14 w[ky][2][c]     15  G[ky][2]
  // The pixels' fourth element P (1.0 inside the image, 0 in the zero border) turns G[ky][kx] = sum_c w d_c (d_c = round(mean_c) - mean_c,
  // rounded to the 16-bit type) into exactly the taps SAME padding keeps, and the centre pixel's P carries V = bias + G[1][1] + (what the
  // rounding of the other eight G dropped) as a (hi, lo) pair: interior pixels see the full constant to ~2^-17, border pixels miss the
  // dropped parts of their missing taps (< 2^-9 |G| each, |G| < 0.03: three orders below the rounding of the output itself).
  std::vector<uint16_t> fp((size_t)CFP_FRAG_BYTES, 0);
  for (int f16 = 0; f16 < 2; ++f16) {
    auto rne = [f16](float v) -> uint16_t { return f16 ? host_rne_f16(v) : ctpn::host_rne_bf16(v); };
    auto tof = [f16](uint16_t h) -> float { return f16 ? host_f16_to_f(h) : ctpn::host_bf16_to_f(h); };
    uint16_t* const fb = fp.data() + (size_t)f16 * (CFP_FRAG_BYTES / 2);
    for (int co = 0; co < 64; ++co) {
      double G[3][3], V = bv[co];
      uint16_t Gh[3][3];
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          G[ky][kx] = 0.0;
          for (int ch = 0; ch < 3; ++ch) G[ky][kx] += (double)tof(rne(w[(size_t)(ky * 9 + kx * 3 + ch) * 64 + co])) * dmean[ch];
          Gh[ky][kx] = rne((float)G[ky][kx]);
          V += (ky == 1 && kx == 1) ? G[ky][kx] : G[ky][kx] - (double)tof(Gh[ky][kx]);
        }
      const uint16_t Vh = rne((float)V), Vl = rne((float)(V - (double)tof(Vh)));
      const int i = co >> 5, r = co & 31;
      for (int ky = 0; ky < 3; ++ky) {
        uint16_t* lo8 = &fb[(((size_t)(i * 3 + ky)) * 64 + r) * 8];          // lane half 0: slots 0..7
        uint16_t* hi8 = &fb[(((size_t)(i * 3 + ky)) * 64 + 32 + r) * 8];     // lane half 1: slots 8..15
        for (int c = 0; c < 3; ++c) {
          lo8[c] = rne(w[(size_t)(ky * 9 + 0 + c) * 64 + co]);
          lo8[4 + c] = rne(w[(size_t)(ky * 9 + 3 + c) * 64 + co]);
          hi8[c] = 0;
          hi8[4 + c] = rne(w[(size_t)(ky * 9 + 6 + c) * 64 + co]);
        }
        lo8[3] = Gh[ky][0];
        lo8[7] = ky == 1 ? Vh : Gh[ky][1];
        hi8[3] = ky == 1 ? Vl : (uint16_t)0;
        hi8[7] = Gh[ky][2];
      }
    }
  }
  CTPN_HIP_TRY(hipMemcpy(frags_dev, f.data(), f.size() * 2, hipMemcpyHostToDevice));
  CTPN_HIP_TRY(hipMemcpy((char*)frags_dev + CF_FRAG_BYTES, fp.data(), fp.size() * 2, hipMemcpyHostToDevice));
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// uint8 feed of the 16-bit modes -> q-image (common.h): one workgroup = 1024 pixels of one image row, four per thread. The row segment's
// bytes are fetched as ALIGNED dwords (any byte alignment of the image pointer and of W * 3) and passed through LDS; a thread reads the
// four aligned dwords around its 12 bytes, shifts them into place (v_alignbyte, the shift is uniform per workgroup) and turns them
// into four pixels (q_B, q_G, q_R, 1.0): two 16-byte stores. Only image pixels are written; the zero frame around them is the buffer's
// initial state (ctpn_api.hip zeroes it when the geometry changes).
// ---------------------------------------------------------------------------------------------
template <typename HF>
__global__ __launch_bounds__(256) void image_to_q_kernel(const uint8_t* __restrict__ img, uint2* __restrict__ q, int N, int H, int W, int Hq, int Wq, int segs) {
  constexpr bool F16 = std::is_same<HF, h_f16>::value;
  constexpr uint32_t ONE = F16 ? 0x3c00u : 0x3f80u;
  __shared__ uint32_t sb[4 * 256 + 4];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int seg = t % segs; t /= segs;
  const int y = t % H, n = t / H;
  const int x0 = seg * 1024;
  const unsigned long long ibase = (unsigned long long)img;
  const unsigned long long iend = ibase + (unsigned long long)N * H * W * 3;
  const unsigned long long b0 = ibase + (((unsigned long long)n * H + y) * W + x0) * 3ull;      // first byte of the segment
  const unsigned long long a0 = b0 & ~3ull;
  const int npx = W - x0 < 1024 ? W - x0 : 1024;                  // pixels of this segment
  const int ndw = (npx * 3 + 3 + 3) / 4 + 1;                      // aligned dwords that cover them (+ one: a thread reads four)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 256 * k;
    if (i < ndw) {
      const unsigned long long a = a0 + 4ull * i;
      sb[i] = a < iend ? *(const uint32_t*)a : 0u;                // a dword that starts below iend holds at least one image byte: mapped
    }
  }
  if (tid < 4) sb[4 * 256 + tid] = 0u;
  __syncthreads();
  const int x = x0 + 4 * tid;
  if (x >= W) return;
  const int sh = (int)(b0 & 3ull);                                // uniform: position of the segment's first byte in its dword
  const uint32_t d0 = sb[3 * tid], d1 = sb[3 * tid + 1], d2 = sb[3 * tid + 2], d3 = sb[3 * tid + 3];
  const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh), w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
  // round(PIXEL_MEANS), BGR (reference lib/fast_rcnn/config.py:200); the differences are integers below 256 in magnitude: exact in either type
  auto bits = [](float f) -> uint32_t { return F16 ? (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f) : (__builtin_bit_cast(uint32_t, f) >> 16); };
  auto px = [&](uint32_t b, uint32_t g, uint32_t r) -> uint2 {
    uint2 v;
    v.x = bits((float)b - 103.f) | (bits((float)g - 116.f) << 16);
    v.y = bits((float)r - 123.f) | (ONE << 16);
    return v;
  };
  const uint2 p0 = px(w0 & 0xffu, (w0 >> 8) & 0xffu, (w0 >> 16) & 0xffu);
  const uint2 p1 = px(w0 >> 24, w1 & 0xffu, (w1 >> 8) & 0xffu);
  const uint2 p2 = px((w1 >> 16) & 0xffu, w1 >> 24, w2 & 0xffu);
  const uint2 p3 = px((w2 >> 8) & 0xffu, (w2 >> 16) & 0xffu, w2 >> 24);
  uint2* dst = q + ((size_t)n * Hq + y + 2) * Wq + x + 2;         // 16-byte aligned: x is a multiple of 4, Wq is even
  if (x + 3 < W) {
    *(uint4*)dst = make_uint4(p0.x, p0.y, p1.x, p1.y);
    *(uint4*)(dst + 2) = make_uint4(p2.x, p2.y, p3.x, p3.y);
  } else {
    dst[0] = p0;
    if (x + 1 < W) dst[1] = p1;
    if (x + 2 < W) dst[2] = p2;
  }
}

int launch_image_to_q(const uint8_t* img, void* q, DType t, int n, int h, int w, hipStream_t s) {
  if (!dtype_is_half(t)) return fail(CTPN_ERR_ARG, "image_to_q: 16-bit modes only");
  const int segs = (w + 1023) / 1024;
  const long long grid = (long long)n * h * segs;
  if (grid <= 0 || grid > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "image_to_q: grid out of range");
  const int hq = conv1_q_h(h), wq = conv1_q_w(w);
  if (t == DType::F16) hipLaunchKernelGGL((image_to_q_kernel<h_f16>), dim3((unsigned)grid), dim3(256), 0, s, img, (uint2*)q, n, h, w, hq, wq, segs);
  else hipLaunchKernelGGL((image_to_q_kernel<h_bf16>), dim3((unsigned)grid), dim3(256), 0, s, img, (uint2*)q, n, h, w, hq, wq, segs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("image_to_q launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// conv1_1 from the q-image, stand-alone: what keep_acts stores and what the ragged columns of conv1_2 (its edge kernel) read. One wave =
// one image row x 64 pixels (two MFMA pixel groups), operands straight from global memory: per tap row ky a lane reads 16 bytes of the
// q-image row -- lanes 0..31 pixels (x - 1, x), lanes 32..63 pixels (x, x + 1); K-slot order: pack_conv1_frags -- and the three MFMAs
// run ky = 0, 1, 2 from a zero accumulator: the sequence of the producer inside conv3x3_wr_kernel, operand for operand.
// ---------------------------------------------------------------------------------------------
template <typename HF>
__global__ __launch_bounds__(256) void conv_first_p_kernel(const uint2* __restrict__ q, const uint4* __restrict__ wfrag, uint16_t* __restrict__ out,
                                                           int N, int H, int W, int Hq, int Wq, int xb, int xe, int tiles_x, int tiles_y) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, fhalf = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y, y = ty * 4 + wave;
  uint4 wf[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) wf[k] = wfrag[k * 64 + lane];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int x = xb + tx * 64 + pt * 32 + l31;
    const bool inside = y < H && x < xe;
    cf_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = cf_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      // image pixel (y - 1 + ky, x - 1 + fhalf) = q pixel (y + 1 + ky, x + 1 + fhalf)
      const uint2* qp = q + ((size_t)n * Hq + (inside ? y + 1 + ky : 0)) * Wq + (inside ? x + 1 + fhalf : 0);
      const uint2 a = qp[0], b = qp[1];
      const uint4 xv = inside ? make_uint4(a.x, a.y, b.x, b.y) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = HalfOps<HF>::mfma_32x32x16(wf[i * 3 + ky], xv, acc[i]);
    }
    uint16_t* op = out + (((long long)n * (H + 2) + y + 1) * (W + 2) + x + 1) * 64 + 8 * fhalf;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        uint32_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t u = HalfOps<HF>::cvt_pk(acc[i][8 * qq + 2 * j], acc[i][8 * qq + 2 * j + 1]);
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          pk[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), s16x2{0, 0}));
        }
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
        if (inside) *(uint4*)(op + i * 32 + 16 * qq) = make_uint4(r0[0], r1[0], r0[1], r1[1]);      // channels 32 i + 16 qq + 8 fhalf .. + 8 of pixel x
      }
  }
}

int launch_conv_first_from_q(const void* q, const void* frags, void* out, DType t, int n, int h, int w, int xb, int xe, hipStream_t s) {
  if (!dtype_is_half(t)) return fail(CTPN_ERR_ARG, "conv_first_from_q: 16-bit modes only");
  if (xb < 0 || xe > w || xb >= xe) return fail(CTPN_ERR_ARG, "conv_first_from_q: empty column range");
  const int tiles_x = (xe - xb + 63) / 64, tiles_y = (h + 3) / 4;
  const long long grid = (long long)n * tiles_x * tiles_y;
  if (grid > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "conv_first_from_q: grid out of range");
  const int hq = conv1_q_h(h), wq = conv1_q_w(w);
  const uint4* fr = (const uint4*)frags;      // conv1_p_frags(...) of the ctx's fragment buffer
  if (t == DType::F16) hipLaunchKernelGGL((conv_first_p_kernel<h_f16>), dim3((unsigned)grid), dim3(256), 0, s, (const uint2*)q, fr, (uint16_t*)out, n, h, w, hq, wq, xb, xe, tiles_x, tiles_y);
  else hipLaunchKernelGGL((conv_first_p_kernel<h_bf16>), dim3((unsigned)grid), dim3(256), 0, s, (const uint2*)q, fr, (uint16_t*)out, n, h, w, hq, wq, xb, xe, tiles_x, tiles_y);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv_first_p launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

static float* g_lut_dev[CTPN_MAX_DEV] = {nullptr};  // per device, built on first use; [0,768): fp32 (v - mean), [768,1536): bits of (bf16 hi | bf16 lo << 16)
static std::mutex g_lut_mu;               // two ctxs may run their first forward from different host threads

static int get_lut(float** out) {
  int dev = 0, rc;
  if ((rc = current_device(dev))) return rc;
  std::lock_guard<std::mutex> lk(g_lut_mu);
  if (!g_lut_dev[dev]) {
    // PIXEL_MEANS, BGR (reference lib/fast_rcnn/config.py:200)
    const double means[3] = {102.9801, 115.9465, 122.7717};
    std::vector<float> h(1536);
    for (int c = 0; c < 3; ++c)
      for (int v = 0; v < 256; ++v) {
        const float f = (float)((double)v - means[c]);
        h[c * 256 + v] = f;
        const uint16_t hi = host_rne_bf16(f), lo = host_rne_bf16(f - host_bf16_to_f(hi));   // f = hi + lo to ~2^-16 relative
        const uint32_t packed = (uint32_t)hi | ((uint32_t)lo << 16);
        std::memcpy(&h[768 + c * 256 + v], &packed, 4);
      }
    float* d = nullptr;
    CTPN_HIP_TRY(hipMalloc(&d, 1536 * sizeof(float)));
    CTPN_HIP_TRY(hipMemcpy(d, h.data(), 1536 * sizeof(float), hipMemcpyHostToDevice));
    g_lut_dev[dev] = d;
  }
  *out = g_lut_dev[dev];
  return CTPN_OK;
}

int launch_conv_first(const void* img, int img_is_f32, const float* w27x64, const float* bias, void* out, DType out_t, int n,
                      int h, int w, hipStream_t s, const void* mfma_frags) {
  float* lut = nullptr;
  int rc = get_lut(&lut);
  if (rc) return rc;
  const int tiles_x = (w + CF_TW - 1) / CF_TW, tiles_y = (h + CF_TH - 1) / CF_TH;
  const unsigned grid = (unsigned)((long long)n * tiles_x * tiles_y);
  if (out_t == DType::SPLIT && !mfma_frags) return fail(CTPN_ERR_ARG, "conv_first: the split-precision output needs the MFMA fragments");
  if (mfma_frags && out_t != DType::F32) {
#define CFM_LAUNCH(IN, OM) hipLaunchKernelGGL((conv_first_mfma_kernel<IN, OM>), dim3(grid), dim3(256), 0, s, (const IN*)img, (const uint4*)mfma_frags, lut, (uint16_t*)out, n, h, w, tiles_x, tiles_y)
    if (img_is_f32) { if (out_t == DType::SPLIT) CFM_LAUNCH(float, 2); else if (out_t == DType::F16) CFM_LAUNCH(float, 1); else CFM_LAUNCH(float, 0); }
    else { if (out_t == DType::SPLIT) CFM_LAUNCH(uint8_t, 2); else if (out_t == DType::F16) CFM_LAUNCH(uint8_t, 1); else CFM_LAUNCH(uint8_t, 0); }
#undef CFM_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv_first_mfma launch: ") + hipGetErrorString(e));
    return CTPN_OK;
  }
#define CF_LAUNCH(IN, OUT) hipLaunchKernelGGL((conv_first_kernel<IN, OUT>), dim3(grid), dim3(256), 0, s, (const IN*)img, w27x64, bias, lut, out, n, h, w, tiles_x, tiles_y)
  if (img_is_f32) { if (out_t == DType::F32) CF_LAUNCH(float, float); else if (out_t == DType::F16) CF_LAUNCH(float, h_f16); else CF_LAUNCH(float, h_bf16); }
  else { if (out_t == DType::F32) CF_LAUNCH(uint8_t, float); else if (out_t == DType::F16) CF_LAUNCH(uint8_t, h_f16); else CF_LAUNCH(uint8_t, h_bf16); }
#undef CF_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv_first launch: ") + hipGetErrorString(e));
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
      if constexpr (std::is_same<OutT, float>::value) dst[(long long)c * dst_ld + r] = v;
      else ((uint16_t*)dst)[(long long)c * dst_ld + r] = HalfOps<OutT>::from_f32(v);
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
  else if (dst_t == DType::F16)
    hipLaunchKernelGGL(pack_transpose_kernel<h_f16>, grid, dim3(256), 0, s, src, src_ld, (h_f16*)dst, dst_ld, rows, cols);
  else if (dst_t == DType::BF16)
    hipLaunchKernelGGL(pack_transpose_kernel<h_bf16>, grid, dim3(256), 0, s, src, src_ld, (h_bf16*)dst, dst_ld, rows, cols);
  else
    return fail(CTPN_ERR_ARG, "pack_transpose: split precision packs through launch_pack_transpose_split");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("pack launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// split precision: src row k = tap * ci + c (TF HWIO flattened / [in][out]), column co -> dst[co][tap][hi(ci) | hi(ci) | lo(ci)] bf16:
// the K layout conv3x3's split kernels (and the LSTM projection GEMM over [hi | lo | hi] pixels) multiply against
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ src, long long src_ld, uint16_t* __restrict__ dst, int taps, int ci, int cols) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)cols * taps * ci;
  if (idx >= total) return;
  const int c = (int)(idx % ci);
  const int tap = (int)((idx / ci) % taps);
  const int co = (int)(idx / ((long long)ci * taps));
  const float v = src[((long long)tap * ci + c) * src_ld + co];
  const uint16_t hi = ctpn_f32_to_bf16(v);
  const uint16_t lo = ctpn_f32_to_bf16(v - ctpn_bf16_to_f32(hi));
  uint16_t* row = dst + ((long long)co * taps + tap) * 3 * ci;
  row[c] = hi; row[ci + c] = hi; row[2 * ci + c] = lo;
}
int launch_pack_transpose_split(const float* src, long long src_ld, void* dst, int taps, int ci, int cols, hipStream_t s) {
  const long long total = (long long)cols * taps * ci;
  hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, src_ld, (uint16_t*)dst, taps, ci, cols);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("pack (split) launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
