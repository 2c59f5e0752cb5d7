// Implicit-GEMM 3x3 convolution / plain GEMM on the gfx950 matrix cores.
//
// Replaces, for one launch: tf.nn.conv2d + bias_add + relu of Network.conv (reference
// lib/networks/network.py:160-183), and tf.matmul + biases of Bilstm / lstm_fc (network.py:110,157).
//
// GEMM view: rows = output pixels m = (n, y, x), K = 9*Ci, columns = Co.
//   * activations live in HBM as NHWC with a one-pixel zero border (n x (H+2) x (W+2) x C), so the
//     im2col gather is "row base + tap offset" with no bounds test and 'SAME' padding is the border;
//   * weights are pre-packed [Co][9*Ci] (k contiguous), the same 128-byte-row image as the pixels;
//   * a K step is one 128-byte strip per row (64 bf16 / 32 fp32 channels of one tap), staged into
//     LDS by global_load_lds_dwordx4 (16 B/lane, LDS image lane-linear) or through VGPRs;
//   * the 16-byte slot of a row is XOR-swizzled with (row>>1)&7 on the SOURCE side and on the
//     ds_read_b128 side, which makes the MFMA fragment reads bank-conflict free (checked by
//     tests/test_layouts.py against the gfx950 ds_read_b128 lane groups);
//   * MFMA operand roles are swapped (weights = A rows, pixels = B columns) so that each lane ends
//     up with 4 consecutive output channels of one pixel: the epilogue adds bias, applies ReLU,
//     converts, transposes through LDS and writes 16 B/lane, pixel-contiguous NHWC;
//   * bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate); fp32: v_mfma_f32_32x32x2_f32, which is an
//     exact fp32 fma chain (the correctness-gate path).
// Block -> tile order is remapped so that each XCD (private 4 MiB L2) walks a contiguous run of
// tiles: neighbouring tiles share input rows (3x3 halo) and all N tiles of one M tile share A.
#include <type_traits>

#include "common.h"

namespace ctpn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef h_bf16 bf16_s;

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  // inputs here are finite (post-ReLU sums); NaN is propagated as a quiet NaN
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <typename T>
__device__ __forceinline__ void mfma_step(f32x16& acc, const uint4& w, const uint4& x) {
  if constexpr (std::is_same<T, float>::value) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.x), __builtin_bit_cast(float, x.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.y), __builtin_bit_cast(float, x.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.z), __builtin_bit_cast(float, x.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.w), __builtin_bit_cast(float, x.w), acc, 0, 0, 0);
  } else {
    acc = HalfOps<T>::mfma_32x32x16(w, x, acc);
  }
}

template <typename T, typename OutT, int BM, int BN, int WGM, int WGN>
struct IGemmCfg {
  static constexpr int NW = WGM * WGN;
  static constexpr int NTHR = NW * 64;
  static constexpr int WM = BM / WGM, WN = BN / WGN;
  static constexpr int MT = WM / 32, NTL = WN / 32;
  static constexpr int BKE = 128 / (int)sizeof(T);
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_LOADS = BM / 8 / NW, B_LOADS = BN / 8 / NW;
  static constexpr int EP = BN * (int)sizeof(OutT) + 16;  // epilogue row pitch (bytes)
  static constexpr int LDS = (2 * STAGE > BM * EP) ? 2 * STAGE : BM * EP;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile is made of 32x32 MFMA tiles");
};

template <typename T, typename OutT, int BM, int BN, int WGM, int WGN, bool GLDS>
__global__ __launch_bounds__(WGM* WGN * 64) void igemm_kernel(IGemm g, int tiles_n) {
  using C = IGemmCfg<T, OutT, BM, BN, WGM, WGN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- block -> tile, XCD-contiguous (bijective for any grid size) ----
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int Wp = g.W + 2, Hp = g.H + 2;
  const int ksteps_per_tap = g.Ci / C::BKE;
  const int nk = g.ntaps * ksteps_per_tap;
  const long long ktot_bytes = (long long)g.ntaps * g.Ci * (long long)sizeof(T);

  // ---- per-lane staging sources ----
  const int srow = lane >> 3, sslot = lane & 7;
  long long a_off[C::A_LOADS];
  long long b_off[C::B_LOADS];
#pragma unroll
  for (int i = 0; i < C::A_LOADS; ++i) {
    const int grp = wave + i * C::NW;
    const int row = grp * 8 + srow;
    long long m = m0 + row;
    if (m > g.M - 1) m = g.M - 1;
    long long off;
    if (g.a_plain) {
      off = m * g.lda;
    } else {
      const int rw = g.rw > 0 ? g.rw : g.W;               // rows run over the column range [rx0, rx0 + rw) of every image row
      const long long hw = (long long)g.H * rw;
      const long long n = m / hw;
      const int rem = (int)(m - n * hw);
      const int y = rem / rw, x = g.rx0 + rem - y * rw;
      const int ty = (g.ntaps == 1) ? g.tap_base_y : 0, tx = (g.ntaps == 1) ? g.tap_base_x : 0;
      off = ((n * Hp + y + ty) * Wp + x + tx) * g.Ci;
    }
    const int srcslot = sslot ^ ((row >> 1) & 7);
    a_off[i] = off * (long long)sizeof(T) + srcslot * 16;
  }
#pragma unroll
  for (int i = 0; i < C::B_LOADS; ++i) {
    const int grp = wave + i * C::NW;
    const int row = grp * 8 + srow;
    const int srcslot = sslot ^ ((row >> 1) & 7);
    b_off[i] = (long long)(n0 + row) * ktot_bytes + srcslot * 16;
  }
  const char* a_base = (const char*)g.a;
  const char* b_base = (const char*)g.wt;

  f32x16 acc[C::NTL][C::MT];
#pragma unroll
  for (int i = 0; i < C::NTL; ++i)
#pragma unroll
    for (int j = 0; j < C::MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 areg[GLDS ? 1 : C::A_LOADS];
  uint4 breg[GLDS ? 1 : C::B_LOADS];

  auto koff_a = [&](int ks) -> long long {
    const int tap = ks / ksteps_per_tap;
    const int c0 = (ks - tap * ksteps_per_tap) * C::BKE;
    long long e = c0;
    if (g.ntaps != 1) {
      const int ky = tap / 3, kx = tap - ky * 3;
      e += (long long)(ky * Wp + kx) * g.Ci;
    }
    return e * (long long)sizeof(T);
  };

  auto issue = [&](int ks, int buf) {
    const long long ka = koff_a(ks);
    const long long kb = (long long)ks * 128;
    char* sb = smem + buf * C::STAGE;
#pragma unroll
    for (int i = 0; i < C::A_LOADS; ++i) {
      const int grp = wave + i * C::NW;
      if constexpr (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_off[i] + ka),
                                         (__attribute__((address_space(3))) void*)(sb + grp * 1024), 16, 0, 0);
      } else {
        areg[i] = *(const uint4*)(a_base + a_off[i] + ka);
      }
    }
#pragma unroll
    for (int i = 0; i < C::B_LOADS; ++i) {
      const int grp = wave + i * C::NW;
      if constexpr (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + b_off[i] + kb),
                                         (__attribute__((address_space(3))) void*)(sb + C::A_BYTES + grp * 1024), 16, 0, 0);
      } else {
        breg[i] = *(const uint4*)(b_base + b_off[i] + kb);
      }
    }
  };
  auto commit = [&](int buf) {  // register-staged variant only
    if constexpr (!GLDS) {
      char* sb = smem + buf * C::STAGE;
#pragma unroll
      for (int i = 0; i < C::A_LOADS; ++i) {
        const int grp = wave + i * C::NW;
        *(uint4*)(sb + grp * 1024 + lane * 16) = areg[i];
      }
#pragma unroll
      for (int i = 0; i < C::B_LOADS; ++i) {
        const int grp = wave + i * C::NW;
        *(uint4*)(sb + C::A_BYTES + grp * 1024 + lane * 16) = breg[i];
      }
    }
  };

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fsw = (frow >> 1) & 7;
  auto compute = [&](int buf) {
    const char* sa = smem + buf * C::STAGE + (wm * C::WM + frow) * 128;
    const char* sbb = smem + buf * C::STAGE + C::A_BYTES + (wn * C::WN + frow) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = ((2 * q + fhalf) ^ fsw) * 16;
      uint4 xf[C::MT], wf[C::NTL];
#pragma unroll
      for (int j = 0; j < C::MT; ++j) xf[j] = *(const uint4*)(sa + j * 32 * 128 + slot);
#pragma unroll
      for (int i = 0; i < C::NTL; ++i) wf[i] = *(const uint4*)(sbb + i * 32 * 128 + slot);
#pragma unroll
      for (int i = 0; i < C::NTL; ++i)
#pragma unroll
        for (int j = 0; j < C::MT; ++j) mfma_step<T>(acc[i][j], wf[i], xf[j]);
    }
  };

  // ---- main loop: double-buffered, one barrier per K step ----
  issue(0, 0);
  commit(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) issue(ks + 1, cur ^ 1);
    compute(cur);
    if (ks + 1 < nk) commit(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias + ReLU + convert, transpose through LDS, 16 B/lane pixel-contiguous stores ----
#pragma unroll
  for (int i = 0; i < C::NTL; ++i) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int co_l = wn * C::WN + i * 32 + 8 * g4 + 4 * fhalf;  // tile-local channel of reg 4*g4
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (g.bias) bv = *(const f32x4*)(g.bias + n0 + co_l);
#pragma unroll
      for (int j = 0; j < C::MT; ++j) {
        const int p = wm * C::WM + j * 32 + frow;
        float v0 = acc[i][j][4 * g4 + 0] + bv[0];
        float v1 = acc[i][j][4 * g4 + 1] + bv[1];
        float v2 = acc[i][j][4 * g4 + 2] + bv[2];
        float v3 = acc[i][j][4 * g4 + 3] + bv[3];
        if (g.relu) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        char* dst = smem + p * C::EP + co_l * (int)sizeof(OutT);
        if constexpr (sizeof(OutT) == 4) {
          f32x4 o = {v0, v1, v2, v3};
          *(f32x4*)dst = o;
        } else {
          uint2 o;
          o.x = HalfOps<OutT>::cvt_pk(v0, v1);
          o.y = HalfOps<OutT>::cvt_pk(v2, v3);
          *(uint2*)dst = o;
        }
      }
    }
  }
  __syncthreads();
  constexpr int CH = BN * (int)sizeof(OutT) / 16;      // 16-byte chunks per pixel row of the tile
  constexpr int EPC = 16 / (int)sizeof(OutT);          // channels per chunk
  char* out_base = (char*)g.out;
  for (int c = tid; c < BM * CH; c += C::NTHR) {
    const int p = c / CH, ch = c - p * CH;
    const long long m = m0 + p;
    const int co = n0 + ch * EPC;
    if (m < g.M && co < g.Co) {
      long long off;
      if (g.out_bordered) {
        const int rw = g.rw > 0 ? g.rw : g.W;
        const long long hw = (long long)g.H * rw;
        const long long n = m / hw;
        const int rem = (int)(m - n * hw);
        const int y = rem / rw, x = g.rx0 + rem - y * rw;
        off = ((n * Hp + y + 1) * Wp + x + 1) * g.ldc + co;
      } else {
        off = m * g.ldc + co;
      }
      *(uint4*)(out_base + off * (long long)sizeof(OutT)) = *(const uint4*)(smem + p * C::EP + ch * 16);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------
template <typename T, typename OutT, int BM, int BN, int WGM, int WGN>
static int launch_cfg(const IGemm& g, hipStream_t s) {
  using C = IGemmCfg<T, OutT, BM, BN, WGM, WGN>;
  const long long tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.Co + BN - 1) / BN;
  const long long nblk = tiles_m * tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "igemm: grid out of range");
  auto k = igemm_kernel<T, OutT, BM, BN, WGM, WGN, true>;      // operands by LDS-DMA (the register-staging variant lost its A/B in round 1)
  static bool attr_done[CTPN_MAX_DEV] = {false};      // per instantiation and device
  int dev = 0, rc;
  if ((rc = current_device(dev)) || (rc = raise_dynamic_lds((const void*)k, C::LDS, attr_done, dev))) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(C::NTHR), C::LDS, s, g, tiles_n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("igemm launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

template <typename T, typename OutT>
static int launch_typed(const IGemm& g, hipStream_t s) {
  // narrow outputs (Co <= 64: conv1_2, heads) use a tall 256x64 tile, everything else 128x128
  // ... and few rows (the heads of one or two images: 9 tall tiles for 256 CUs) 64 x 64 tiles on four waves of 32 x 32: a quarter of the work
  // per wave, four times the workgroups; every output is the same K-ordered sum in either shape
  if (g.Co <= 64 && g.M <= 4096) return launch_cfg<T, OutT, 64, 64, 2, 2>(g, s);
  if (g.Co <= 64) return launch_cfg<T, OutT, 256, 64, 4, 1>(g, s);
  return launch_cfg<T, OutT, 128, 128, 2, 2>(g, s);
}

int launch_igemm(const IGemm& g, DType in_t, DType out_t, hipStream_t s) {
  const int bke = (in_t == DType::F32) ? 32 : 64;
  if (g.Ci <= 0 || g.Ci % bke != 0) return fail(CTPN_ERR_ARG, "igemm: Ci must be a multiple of the 128-byte K strip");
  if (g.ntaps != 1 && g.ntaps != 9) return fail(CTPN_ERR_ARG, "igemm: ntaps must be 1 or 9");
  if (g.M <= 0 || g.Co <= 0) return fail(CTPN_ERR_ARG, "igemm: empty problem");
  const int epc = (out_t == DType::F32) ? 4 : 8;
  if (g.Co % epc != 0 || g.ldc % epc != 0) return fail(CTPN_ERR_ARG, "igemm: Co/ldc must be multiples of a 16-byte chunk");
  if (in_t == DType::F32 && out_t == DType::F32) return launch_typed<float, float>(g, s);
  if (in_t == DType::BF16 && out_t == DType::BF16) return launch_typed<h_bf16, h_bf16>(g, s);
  if (in_t == DType::BF16 && out_t == DType::F32) return launch_typed<h_bf16, float>(g, s);
  if (in_t == DType::F16 && out_t == DType::F16) return launch_typed<h_f16, h_f16>(g, s);
  if (in_t == DType::F16 && out_t == DType::F32) return launch_typed<h_f16, float>(g, s);
  if (in_t == DType::BF16 && out_t == DType::F16) return launch_typed<h_bf16, h_f16>(g, s);
  return fail(CTPN_ERR_ARG, "igemm: unsupported dtype pair");
}

}  // namespace ctpn
