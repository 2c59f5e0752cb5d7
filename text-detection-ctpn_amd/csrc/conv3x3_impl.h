// 3x3 convolution with tap reuse out of LDS (the VGG trunk + rpn_conv/3x3), bias + ReLU (+ 2x2 max-pool) fused -- kernels and launch
// templates. Included by one translation unit per arithmetic type (conv3x3_f32.hip, conv3x3_bf16.hip, conv3x3_f16.hip,
// conv3x3_split.hip: they compile in parallel) and by conv3x3.hip (launch_conv3x3: layer-level decisions).
//
// Replaces tf.nn.conv2d + bias_add + relu of Network.conv (reference lib/networks/network.py:160-183) and, when POOL,
// the Network.max_pool that follows it (network.py:189-196; VGGnet_test.py:23,26,30,34).
//
// igemm.hip treats the conv as im2col GEMM and therefore moves every input pixel L2 -> LDS nine times (once per
// tap): at a 128x128 tile that is 64 B/clk/CU, i.e. 39 TB/s at MFMA peak -- above what the L2s deliver -- and it is
// why that kernel sits at ~28 % of the bf16 roofline. Here a workgroup owns 256 output pixels x BN channels and, per
// 64-channel chunk (one 128-byte strip per pixel), stages the INPUT window those pixels need ONCE into LDS; the nine
// taps are nine shifted views of that window (LDS row + ky*pitch + kx), so only the weight strip changes per K step:
//     2D mode   : window = (8+2) x (32+2) pixel patch of one image (any W; needed for the 2x2 pool fusion)
//     flat mode : window = 256 + 2*(W+2) + 2 CONSECUTIVE pixels of the bordered NHWC buffer (M runs over bordered
//                 positions, border outputs are computed and dropped) -- no tile quantisation on the small
//                 75x112 / 37x56 maps, perfectly contiguous staging
// L2 -> LDS traffic drops ~3x (20 B/clk/CU at BN = 128). Everything else follows igemm.hip: 128-byte rows with the
// 16-byte slot XOR-swizzled by (row>>1)&7 (source side for global_load_lds, read side for ds_read_b128: conflict-free
// for ANY 32 consecutive rows, tests/test_layouts.py), swapped MFMA operand roles (weights = A rows) so a lane owns
// 4 consecutive channels of one pixel, epilogue through LDS with 16 B/lane stores, XCD-contiguous block order.
//
// Arithmetic types (template parameter T): float (exact-fp32 MFMA 32x32x2), h_bf16, h_f16 (32x32x16, common.h).
// SPLIT (CTPN_PREC_SPLIT; T = h_bf16): every activation and weight is a (hi, lo) pair of bf16 and a product is three MFMAs,
//     x w ~= x_hi w_hi + x_lo w_hi + x_hi w_lo      (fp32 accumulate; the dropped x_lo w_lo is ~2^-17 of the product)
// laid out so that the K loop does not change at all: a pixel of a C-channel map is [hi(C) | lo(C)] (2 C bf16), a weight row is
// [w_hi(Ci) | w_hi(Ci) | w_lo(Ci)] per tap, and the kernel runs a plain bf16 convolution over K = 9 x 3 Ci whose 64-channel input
// chunk c is chunk (c < 2 Ci / 64 ? c : c - 2 Ci / 64) of the pixel (`a_wrap`: the third K block re-reads the hi plane -- from LDS-DMA's
// point of view just another chunk of the same pixel, served by L2). Only the epilogue differs: ReLU in fp32, then
// hi = RNE_bf16(v), lo = RNE_bf16(v - hi) into the two planes (and, for the layer that feeds the LSTM projection GEMM, hi once more:
// [hi | lo | hi], so that GEMM is a plain K = 3 C product as well).
#pragma once
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"

namespace ctpn {

constexpr int C3_MAX_DEV = CTPN_MAX_DEV;     // launch state is per device, see common.h
static inline int c3_device(int& dev) { return current_device(dev); }
static inline int c3_cu_count(int dev, int& ncu) { return device_cu_count(dev, ncu); }
static inline int c3_raise_lds(const void* kern, bool (&done)[C3_MAX_DEV], int dev) { return raise_dynamic_lds(kern, 160 * 1024, done, dev); }

typedef ctpn_f32x16 c3_f32x16;
typedef __attribute__((ext_vector_type(4))) float c3_f32x4;
typedef uint32_t c3_u32x4 __attribute__((ext_vector_type(4)));   // native vector: inline-asm register operands ("v", tied "+v") need one

template <typename T>
__device__ __forceinline__ void c3_mfma(c3_f32x16& acc, const uint4& w, const uint4& x) {
  if constexpr (std::is_same<T, float>::value) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.x), __builtin_bit_cast(float, x.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.y), __builtin_bit_cast(float, x.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.z), __builtin_bit_cast(float, x.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, w.w), __builtin_bit_cast(float, x.w), acc, 0, 0, 0);
  } else {
    acc = HalfOps<T>::mfma_32x32x16(w, x, acc);
  }
}
// two fp32 -> one packed pair of the 16-bit output type
template <typename OutT>
__device__ __forceinline__ uint32_t c3_cvt_pk(float lo, float hi) { return HalfOps<OutT>::cvt_pk(lo, hi); }

template <int N>
__device__ __forceinline__ void c3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Raw s_barrier. It orders NOTHING by itself on gfx950: no vmcnt, no lgkmcnt wait is implied, and hipcc moves register-only work (MFMAs and
// the s_waitcnt for their LDS operands) across it freely. Every buffer hand-over in these kernels therefore states both waits explicitly in
// front of it: `s_waitcnt lgkmcnt(0)` (my LDS reads of the buffer the next LDS-DMA recycles have executed) and the counted vmcnt (my DMA
// pieces of the buffer the next step reads have landed). tools/scan_barrier_reads.py checks the compiled code for LDS reads in flight
// across a barrier.
__device__ __forceinline__ void c3_barrier() { __builtin_amdgcn_s_barrier(); }

// LDS-DMA issued from inline asm: hipcc does not count it, so it neither drains it with vmcnt(0) at the next
// barrier / ds_read nor waits for it at all -- every wait is the kernel's own counted s_waitcnt (cdna guide 5.7).
// lds_dst: wave-uniform LDS byte address (the hardware adds lane * 16); gsrc: this lane's 16 source bytes.
__device__ __forceinline__ void c3_glds16_asm(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// LDS-DMA, scalar base + per-lane 32-bit offset: lds_dst is the wave-uniform LDS byte address (hardware adds lane * 16)
// (m0 is declared clobbered instead of being saved and restored around every piece: two SALU fewer per KiB in the K loops)
// clang warns about every reserved register on a clobber list (-Winline-asm: "may not be preserved across the asm statement"). That is the
// contract wanted here: nothing else in these kernels keeps a value in m0 across the statement (the compiler's own LDS-DMA / ds_*_addtid /
// s_movrel uses would; there are none, and tests/test_gpu_round6.py::test_lds_dma_helper_forms_agree compares this form with the
// save / restore form c3_glds16_asm tile for tile on the device, so a compiler that starts to keep state in m0 is caught). The
// diagnostic is silenced for THIS statement only; the build fails on any other warning (__graft_entry__.build()).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void c3_glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory", "m0");
}
#pragma clang diagnostic pop

// LDS fragment read issued from inline asm (cdna guide 5.7 form iii): program order is pinned by `volatile`, completion
// is the kernel's own counted s_waitcnt lgkmcnt + sched_barrier(0) in front of the first consumer.
__device__ __forceinline__ void c3_ds_read_b128_asm(uint4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr));
}
template <int N>
__device__ __forceinline__ void c3_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <typename OutT>
__device__ __forceinline__ uint4 c3_max4(const uint4& a, const uint4& b) {
  uint4 r;
  if constexpr (sizeof(OutT) == 4) {
    r.x = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x)));
    r.y = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y)));
    r.z = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z)));
    r.w = __builtin_bit_cast(uint32_t, fmaxf(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w)));
  } else {
    auto mx = [](uint32_t p, uint32_t q) -> uint32_t {
      const uint32_t lo = (HalfOps<OutT>::to_f32((uint16_t)p) >= HalfOps<OutT>::to_f32((uint16_t)q)) ? (p & 0xffffu) : (q & 0xffffu);
      const uint32_t hi = (HalfOps<OutT>::to_f32((uint16_t)(p >> 16)) >= HalfOps<OutT>::to_f32((uint16_t)(q >> 16))) ? (p & 0xffff0000u) : (q & 0xffff0000u);
      return lo | hi;
    };
    r.x = mx(a.x, b.x); r.y = mx(a.y, b.y); r.z = mx(a.z, b.z); r.w = mx(a.w, b.w);
  }
  return r;
}

// SPLIT epilogues: four fp32 values (ReLU already applied) of channels co .. co + 3 of one pixel -> the hi and lo planes ([hi | lo | hi] with dup)
__device__ __forceinline__ void c3_split_store4(char* pix_base, int co, const uint4& v, int Co, int dup) {
  uint2 hi, lo;
  ctpn_split_pk_bf16(__builtin_bit_cast(float, v.x), __builtin_bit_cast(float, v.y), hi.x, lo.x);
  ctpn_split_pk_bf16(__builtin_bit_cast(float, v.z), __builtin_bit_cast(float, v.w), hi.y, lo.y);
  *(uint2*)(pix_base + co * 2) = hi;
  *(uint2*)(pix_base + (Co + co) * 2) = lo;
  if (dup) *(uint2*)(pix_base + (2 * Co + co) * 2) = hi;
}

struct Conv3 {
  const void* in;      // bordered NHWC, T
  const void* wt;      // [co_pad][9*Ci] T
  const float* bias;
  void* out;           // bordered NHWC, OutT (may be null when POOL and the full-resolution output is not kept)
  void* pool_out;      // bordered NHWC of the pooled map (POOL only)
  int N, H, W, Ci, Co, relu;
  int tiles_x, tiles_y;       // 2D mode
  long long m_total;          // flat mode: N*(H+2)*(W+2)
  int a_rows;                 // LDS rows of one A window (multiple of 8)
  long long ptiles_total;     // persistent kernel: pixel tiles x tiles_n
  int w_cover;                // 2D mode: columns [0, w_cover) are this launch's (0 = all W); the rest belongs to a strip launch
  int abl;                    // persistent kernel, timing only and only in -DCTPN_ABLATION builds (`make ablation`; CTPN_C3_P_ABL): 1 = skip the epilogue,
                              // 2 = its arithmetic without the stores (WRONG results; the product library ignores the field)
  int tiles_n;
  // persistent kernel, flat windows: half-tile tail (see conv3x3_p_kernel). Tiles [0, ht_full) are walked whole; the ht_r tiles behind them
  // are split into two halves of 128 consecutive pixels: 2 * ht_r work items for the first 2 * ht_r workers of the tail round. 0: no split.
  long long ht_full;
  int ht_r;
  // persistent kernel, 2D patches without a fused pool: tile rows run over the bordered rows of the WHOLE batch (tiles_y counts them)
  // instead of per image -- see c3_launch_p
  int stacked;
  // SPLIT kernels (see the file comment): Ci above is the K width per tap (3 x the layer's input channels); in_pitch = bf16 elements per
  // input pixel (2 x), a_wrap = first 64-channel K chunk that re-reads the hi plane (chunk c -> pixel chunk c - a_wrap), out_pitch = bf16
  // elements per output pixel (2 Co, or 3 Co with dup_hi: [hi | lo | hi]). Non-split launches: in_pitch = Ci, out_pitch = Co.
  int in_pitch, a_wrap, out_pitch, dup_hi;
  // tuning options (ctpn_set_option; same results either way): -1 = the kernel family's default
  int opt_ahead;
  int opt_small;              // flat windows at one image per call: 0 = half tiles of 128 pixels (round 3), else 64-pixel x 128-channel items (round 6)
  int opt_p64;                // Co = 64 layers outside the weights-in-registers kernel: 0 = the non-persistent kernel, else conv3x3_p_kernel<.., BN_T = 64>
  // conv1_2 with conv1_1 computed in its window stage (conv3x3_wr_kernel FUSE): the batch's q-image and conv1_1's fragments; `in` is unused
  const void* q1; const void* q1_frags;
};

constexpr int C3_BM = 256;

// 16 x 16 patches: a 32-pixel MFMA tile is two patch rows of 16. Lanes 16..31 take the second row ROTATED by two columns
// (lane 16 + k owns column (k + 14) & 15): with the 18-pixel LDS row pitch that makes the LDS row of lane l congruent to l
// mod 16 again, which is what keeps every ds_read_b128 lane group on 16 distinct bank quads (un-rotated: 1.3-1.45 x the
// busy cycles in SQ_LDS_BANK_CONFLICT on the conv4 layers).
__device__ __forceinline__ int c3_tw16_col(int l31) { return (l31 & 16) ? ((l31 - 2) & 15) : l31; }

// TW: width of the 2D output patch (32 -> 8 x 32, 16 -> 16 x 16; a 32-pixel MFMA tile is one row of 32 or two rows of 16).
// The launcher picks the shape that wastes fewer pixels on the layer's map (e.g. 74 x 112 pooled: 19 % -> 7.5 %).
// SPLIT: T = h_bf16, OutT = float (the LDS staging of the epilogue holds the fp32 sums; the global stores split them)
template <typename T, typename OutT, int BN, int WGM, int WGN, bool FLAT, bool POOL, int ABUF, int NBUF, int TW = 32, bool SPLIT = false>
__global__ __launch_bounds__(WGM* WGN * 64) void conv3x3_kernel(Conv3 g) {
  static_assert(!SPLIT || (std::is_same<T, h_bf16>::value && std::is_same<OutT, float>::value), "split kernels run bf16 MFMAs and stage fp32 sums");
  constexpr int C3_TW = TW, C3_TH = C3_BM / TW, C3_PW2D = C3_TW + 2;
  static_assert(TW == 32 || TW == 16, "2D patch is 8 x 32 or 16 x 16");
  constexpr int NW = WGM * WGN, NTHR = NW * 64;
  constexpr int MT = (C3_BM / 32) / WGM;      // pixel tiles (32 px) per wave
  constexpr int NTL = (BN / 32) / WGN;        // channel tiles per wave
  constexpr int BKE = 128 / (int)sizeof(T);
  constexpr int B_BYTES = BN * 128;
  constexpr int B_LOADS = BN / 8 / NW;        // 1 KB wave-loads of B per wave per K step
  constexpr int AG_MAX = FLAT ? (61 + NW - 1) / NW : (43 + NW - 1) / NW;   // A groups (8 rows each) per wave, upper bound
  constexpr int EP = BN * (int)sizeof(OutT) + 16;
  static_assert(BN % (8 * NW) == 0 && (C3_BM / 32) % WGM == 0 && (BN / 32) % WGN == 0, "bad wave split");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int tn = lid % g.tiles_n;
  const int pt = lid / g.tiles_n;
  const int n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int Wp = g.W + 2, Hp = g.H + 2;
  const int PW = FLAT ? Wp : C3_PW2D;          // LDS-window pixel pitch of one image row
  const int a_bytes = g.a_rows * 128;
  char* const sA = smem;                        // ABUF windows
  char* const sB = smem + ABUF * a_bytes;       // NBUF weight strips
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // tile origin
  int img = 0, y0 = 0, x0 = 0;
  long long q0 = 0;
  if constexpr (FLAT) {
    q0 = (long long)pt * C3_BM;
  } else {
    const int per_img = g.tiles_x * g.tiles_y;
    img = pt / per_img;
    const int rem = pt - img * per_img;
    const int tyi = rem / g.tiles_x;
    y0 = tyi * C3_TH;
    x0 = (rem - tyi * g.tiles_x) * C3_TW;
  }

  // ---- staging sources ----
  const int srow = lane >> 3, sslot = lane & 7;
  const int a_groups = g.a_rows >> 3;
  long long a_off[AG_MAX];
#pragma unroll
  for (int i = 0; i < AG_MAX; ++i) {
    int grp = wave + i * NW;
    if (NBUF == 3 && grp > a_groups - 1) grp = a_groups - 1;   // counted-vmcnt pipeline: every wave issues every slot
    const int r = grp * 8 + srow;
    long long pix;
    if constexpr (FLAT) {
      long long q = q0 - PW - 1 + r;
      q = q < 0 ? 0 : (q > g.m_total - 1 ? g.m_total - 1 : q);
      pix = q;
    } else {
      const int i2 = r / C3_PW2D, j2 = r - i2 * C3_PW2D;
      int yy = y0 + i2, xx = x0 + j2;
      yy = yy > Hp - 1 ? Hp - 1 : yy;
      xx = xx > Wp - 1 ? Wp - 1 : xx;
      pix = ((long long)img * Hp + yy) * Wp + xx;
    }
    a_off[i] = pix * g.in_pitch * (long long)sizeof(T) + ((sslot ^ ((r >> 1) & 7)) << 4);
  }
  const long long ktot_bytes = 9LL * g.Ci * (long long)sizeof(T);
  long long b_off[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int row = (wave + i * NW) * 8 + srow;
    b_off[i] = (long long)(n0 + row) * ktot_bytes + ((sslot ^ ((row >> 1) & 7)) << 4);
  }
  const char* a_base = (const char*)g.in;
  const char* b_base = (const char*)g.wt;

  auto issue_a_group = [&](int i, int chunk, int buf) {   // i-th group of this wave
    if constexpr (SPLIT) chunk = chunk >= g.a_wrap ? chunk - g.a_wrap : chunk;      // third K block: the hi plane again
    int grp = wave + i * NW;
    if (NBUF == 3 && grp > a_groups - 1) grp = a_groups - 1;   // duplicate of the last group: same bytes, same place
    if constexpr (NBUF == 3) {
      c3_glds16_asm(a_base + a_off[i] + (long long)chunk * 128, __builtin_amdgcn_readfirstlane(lds0 + buf * a_bytes + grp * 1024));
    } else {
      if (grp < a_groups)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_off[i] + (long long)chunk * 128),
                                         (__attribute__((address_space(3))) void*)(sA + buf * a_bytes + grp * 1024), 16, 0, 0);
    }
  };
  auto issue_b = [&](int chunk, int tap, int buf) {
    const long long kb = ((long long)tap * g.Ci + (long long)chunk * BKE) * (long long)sizeof(T);
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      if constexpr (NBUF == 3)
        c3_glds16_asm(b_base + b_off[i] + kb, __builtin_amdgcn_readfirstlane(lds0 + ABUF * a_bytes + buf * B_BYTES + (wave + i * NW) * 1024));
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + b_off[i] + kb),
                                         (__attribute__((address_space(3))) void*)(sB + buf * B_BYTES + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };

  c3_f32x16 acc[NTL][MT];
#pragma unroll
  for (int i = 0; i < NTL; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, fhalf = lane >> 5;
  const int fswB = (l31 >> 1) & 7;
  int tilebase[MT];   // LDS row of (pixel tile j, lane) at tap (0,0)
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    if constexpr (FLAT) tilebase[j] = (wm * MT + j) * 32 + l31;
    else if constexpr (TW == 32) tilebase[j] = (wm * MT + j) * C3_PW2D + l31;
    else tilebase[j] = (2 * (wm * MT + j) + (l31 >> 4)) * C3_PW2D + c3_tw16_col(l31);
  }

  auto compute = [&](int abuf, int bbuf, int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int rowoff = ky * PW + kx;
    const char* sa = sA + abuf * a_bytes;
    const char* sb = sB + bbuf * B_BYTES + (wn * (BN / WGN) + l31) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = 2 * q + fhalf;
      uint4 xf[MT], wf[NTL];
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int r = tilebase[j] + rowoff;
        xf[j] = *(const uint4*)(sa + r * 128 + ((slot ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < NTL; ++i) wf[i] = *(const uint4*)(sb + i * 32 * 128 + ((slot ^ fswB) << 4));
#pragma unroll
      for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) c3_mfma<T>(acc[i][j], wf[i], xf[j]);
    }
  };

  // ---- main loop: chunk-major, tap-minor ----
  const int nchunks = g.Ci / BKE;
  {
    // Three weight-strip buffers, prefetch distance 2, COUNTED vmcnt + raw s_barrier: the strip for step s+2 (and
    // the next chunk's window slices) stay in flight across the barrier; only what step s+1 needs is waited for.
    // Step s = 9*chunk + tap uses strip buffer s % 3 = tap % 3. Every wave issues the same number of loads per
    // step (padded with duplicates), so the vmcnt immediates are compile-time constants.
    static_assert(NBUF == 3, "pipeline is written for three strip buffers");
#pragma unroll
    for (int i = 0; i < AG_MAX; ++i) issue_a_group(i, 0, 0);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    c3_wait_vm<B_LOADS>();
    c3_barrier();
    auto step = [&](auto tc, auto lastc, int c, int ab) {
      constexpr int t = decltype(tc)::value;
      constexpr bool last = decltype(lastc)::value;
      // the next chunk's window slices go out in steps 0..7 ONLY: what step 8 issues is still in flight when the next chunk
      // starts (its wait leaves this step's loads pending), and with 4 waves (11 groups per wave) a slice issued there was
      // read before it had landed -- a rare wrong pixel row in the fp32 conv1_2
      constexpr int nA = (ABUF == 2 && !last && t < 8 && AG_MAX > t) ? (AG_MAX - t + 7) / 8 : 0;
      constexpr bool has_b = (t + 2 < 9) || !last;
      if constexpr (has_b) {
        if constexpr (t + 2 < 9) issue_b(c, t + 2, (t + 2) % 3);
        else issue_b(c + 1, t + 2 - 9, (t + 2) % 3);
      }
      if constexpr (nA > 0) {
#pragma unroll
        for (int i = t; i < AG_MAX; i += 8) issue_a_group(i, c + 1, ab ^ 1);
      }
      compute(ab, t % 3, t);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // no fragment read in flight across the barrier: the next step's DMA recycles the strip just read (see conv3x3_p_kernel)
      c3_wait_vm<(has_b ? B_LOADS : 0) + nA>();
      c3_barrier();
    };
    auto chunk = [&](auto lastc, int c) {
      const int ab = (ABUF == 2) ? (c & 1) : 0;
      step(std::integral_constant<int, 0>{}, lastc, c, ab);
      step(std::integral_constant<int, 1>{}, lastc, c, ab);
      step(std::integral_constant<int, 2>{}, lastc, c, ab);
      step(std::integral_constant<int, 3>{}, lastc, c, ab);
      step(std::integral_constant<int, 4>{}, lastc, c, ab);
      step(std::integral_constant<int, 5>{}, lastc, c, ab);
      step(std::integral_constant<int, 6>{}, lastc, c, ab);
      step(std::integral_constant<int, 7>{}, lastc, c, ab);
      step(std::integral_constant<int, 8>{}, lastc, c, ab);
    };
    for (int c = 0; c + 1 < nchunks; ++c) chunk(std::false_type{}, c);
    chunk(std::true_type{}, nchunks - 1);
    __syncthreads();
  }

  // ---- epilogue ----
#pragma unroll
  for (int i = 0; i < NTL; ++i) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int co_l = wn * (BN / WGN) + i * 32 + 8 * g4 + 4 * fhalf;
      c3_f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (g.bias) bv = *(const c3_f32x4*)(g.bias + n0 + co_l);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int p = (wm * MT + j) * 32 + ((!FLAT && TW == 16) ? (l31 & 16) + c3_tw16_col(l31) : l31);   // tile-local pixel, row-major
        float v0 = acc[i][j][4 * g4 + 0] + bv[0];
        float v1 = acc[i][j][4 * g4 + 1] + bv[1];
        float v2 = acc[i][j][4 * g4 + 2] + bv[2];
        float v3 = acc[i][j][4 * g4 + 3] + bv[3];
        if (g.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        char* dst = smem + p * EP + co_l * (int)sizeof(OutT);
        if constexpr (sizeof(OutT) == 4) {
          c3_f32x4 o = {v0, v1, v2, v3};
          *(c3_f32x4*)dst = o;
        } else {
          uint2 o;
          o.x = c3_cvt_pk<OutT>(v0, v1);
          o.y = c3_cvt_pk<OutT>(v2, v3);
          *(uint2*)dst = o;
        }
      }
    }
  }
  __syncthreads();
  constexpr int CH = BN * (int)sizeof(OutT) / 16;
  constexpr int EPC = 16 / (int)sizeof(OutT);
  if (g.out) {
    char* out_base = (char*)g.out;
    for (int c = tid; c < C3_BM * CH; c += NTHR) {
      const int p = c / CH, ch = c - p * CH;
      const int co = n0 + ch * EPC;
      if (co >= g.Co) continue;
      long long opix;
      bool ok;
      if constexpr (FLAT) {
        const long long q = q0 + p;
        const long long per = (long long)Hp * Wp;
        const long long im = q / per;
        const int rem = (int)(q - im * per);
        const int yb = rem / Wp, xb = rem - yb * Wp;
        ok = q < g.m_total && yb >= 1 && yb <= g.H && xb >= 1 && xb <= g.W;
        opix = q;
      } else {
        const int y = y0 + p / C3_TW, x = x0 + p % C3_TW;
        ok = y < g.H && x < g.W;
        opix = ((long long)img * Hp + y + 1) * Wp + x + 1;
      }
      if constexpr (SPLIT) { if (ok) c3_split_store4(out_base + opix * g.out_pitch * 2, co, *(const uint4*)(smem + p * EP + ch * 16), g.Co, g.dup_hi); }
      else if (ok) *(uint4*)(out_base + (opix * g.Co + co) * (long long)sizeof(OutT)) = *(const uint4*)(smem + p * EP + ch * 16);
    }
  }
  if constexpr (POOL && !FLAT) {
    const int Ho = g.H >> 1, Wo = g.W >> 1;
    char* pool_base = (char*)g.pool_out;
    for (int c = tid; c < (C3_BM / 4) * CH; c += NTHR) {
      const int pp = c / CH, ch = c - pp * CH;
      const int co = n0 + ch * EPC;
      if (co >= g.Co) continue;
      const int py = pp / (C3_TW / 2), px = pp % (C3_TW / 2);   // (TH/2) x (TW/2) pooled pixels
      const int Y = (y0 >> 1) + py, X = (x0 >> 1) + px;
      if (Y >= Ho || X >= Wo) continue;
      const int p00 = (2 * py) * C3_TW + 2 * px;
      const uint4 a = *(const uint4*)(smem + p00 * EP + ch * 16);
      const uint4 b = *(const uint4*)(smem + (p00 + 1) * EP + ch * 16);
      const uint4 cc = *(const uint4*)(smem + (p00 + C3_TW) * EP + ch * 16);
      const uint4 d = *(const uint4*)(smem + (p00 + C3_TW + 1) * EP + ch * 16);
      const uint4 m = c3_max4<OutT>(c3_max4<OutT>(a, b), c3_max4<OutT>(cc, d));
      const long long opix = ((long long)img * (Ho + 2) + Y + 1) * (Wo + 2) + X + 1;
      if constexpr (SPLIT) c3_split_store4(pool_base + opix * g.out_pitch * 2, co, m, g.Co, g.dup_hi);     // max of the fp32 sums, then split: pooling commutes with the monotone rounding
      else *(uint4*)(pool_base + (opix * g.Co + co) * (long long)sizeof(OutT)) = m;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Persistent form of conv3x3_kernel (256 pixels x 128 channels, 8 waves, three weight strips, double window):
// one workgroup per CU walks tiles w, w + G, ... and its load pipeline never drains -- while a tile's last chunk is on
// the MFMAs, the window slices and the first two weight strips of the NEXT tile stream in exactly like the next chunk
// of the same tile would. The epilogue is register-only (bias from LDS, ReLU, v_cvt_pk, v_permlane32_swap -> 16-byte
// stores; the 2x2 pool is a max over the wave's two pixel rows + lane^1, + lane^16 for 16 x 16 patches), because the
// LDS is busy receiving the next tile. conv3x3_kernel pays per tile: an exposed prologue (first window + two strips,
// ~1.5 us), the LDS-staged epilogue (~1.5 us) and the workgroup launch -- 15 % of a K = 1152 tile, 9 % at K = 2304,
// 5 % at K = 4608, which is the order the layers' TFLOP/s were in (conv2_2 951 ... conv4_2 1236).
// ---------------------------------------------------------------------------------------------
// SPLIT: T = OutT = h_bf16 over (hi, lo) planes, see the file comment
// BN_T = 64 (Co = 64: conv1_2 outside the 16-bit modes' weights-in-registers kernel, i.e. split precision and fp32): the same walk, windows,
// strips and pipeline on 256 pixels x 64 channels -- 8 waves = 4 (pixel groups of 64) x 2 (channel halves of 32), ONE channel tile per wave
// (two MFMAs per k-slice group; 2 x + 1 w fragment reads, the carried row fragment saves a third of the x reads), 8 x 32 patches only. The
// non-persistent kernel this replaces for that layer paid an exposed prologue and an LDS-staged epilogue per 256 x 64 tile of a K = 1728
// loop: 4.53 ms = 33.7 % of the bf16 peak for split conv1_2 at batch 32 (VERDICT r5 "weak" 1).
// BM_T = 64 (flat windows, round 6): 64 pixels x 128 channels per workgroup, 8 waves = 2 pixel tiles x 4 channel tiles, ONE 32 x 32 MFMA tile per
// wave. For the ONE-IMAGE call: conv5_x / rpn_conv of a 600 x 900 image are 36 tiles of 256 pixels x 128 channels -- 72 half-tile items on 256 CUs,
// each walking 72 K steps of 16 MFMAs per wave with one wave per SIMD (30 us per layer, 13 % of peak); as 144 items of 64 x 128 every wave is
// busy with 4 MFMAs per K step and the step is bounded by the LDS (two fragment reads per MFMA) instead of by one wave's latency chain.
// Every output is still summed by one wave in the usual K order: the bits do not depend on the form (tools/r6_lone_vs_batch.py).
// HT32 (round 6): the half-tile tail for 8 x 32 patches as well (a half = the tile shifted by four patch rows, computed by the waves of pixel
// groups 0 and 1), for one or two images per call only: there the ragged-column edge kernels run in the stream, not beside this kernel, so the
// ~30 registers the idle path costs are available.
template <typename T, typename OutT, bool FLAT, bool POOL, int TW, bool AHEAD = false, bool SPLIT = false, int BN_T = 128, int BM_T = 256, bool HT32 = false>
__global__ __launch_bounds__(512) void conv3x3_p_kernel(Conv3 g) {
  static_assert(!HT32 || (TW == 32 && !FLAT && BN_T == 128 && BM_T == 256), "half-tile tails for 8 x 32 patches: the plain 128-channel form");
  static_assert(!SPLIT || std::is_same<T, h_bf16>::value, "split kernels run bf16 MFMAs");
  static_assert(BN_T == 128 || (BN_T == 64 && !FLAT && TW == 32), "the 64-channel form exists for 8 x 32 patches");
  static_assert(BM_T == 256 || (BM_T == 64 && FLAT && BN_T == 128), "the 64-pixel form exists for flat windows");
  constexpr int BM = BM_T;
  constexpr int BN = BN_T, WGN = BM == 64 ? 4 : 2;         // 8 waves = 4 (pixel tiles) x 2 (channel halves); 64-pixel form: 2 x 4
  constexpr int C3_TW = TW, C3_PW2D = C3_TW + 2;
  constexpr int NW = 8;
  constexpr int MT = BM == 64 ? 1 : 2, NTL = BN / (32 * WGN);
  constexpr int BKE = 128 / (int)sizeof(T);
  constexpr int B_BYTES = BN * 128;
  constexpr int B_LOADS = BN / 8 / NW;        // 2
  constexpr int AG_MAX = FLAT ? ((BM == 64 ? 37 : 61) + NW - 1) / NW : (43 + NW - 1) / NW;
  static_assert(sizeof(T) == sizeof(OutT), "in and out types match");
  static_assert(!(FLAT && POOL), "the pool fusion needs 2D patches");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int G = gridDim.x, bid = blockIdx.x;
  const int xq = G >> 3, xr = G & 7, xcd = bid & 7;
  const int w0 = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);   // XCD-mates walk neighbouring tiles
  const long long total = g.ptiles_total;
  if (w0 >= g.ht_full + 2LL * g.ht_r) return;                  // work items: whole tiles + two halves per split tile (= total when nothing is split)
  // Half-tile tail (flat windows: conv5_x / rpn_conv; 16 x 16 patches: conv4_x). With G workers and total = q G + r tiles the last round keeps r workers busy and
  // G - r idle: 1132 tiles on 256 CUs pay 5 rounds for 4.42 of work. For r <= G / 2 the r tail tiles are split into two halves of 128
  // PIXELS (2 r work items on 2 r workers): a half is the tile shifted by 128 flat pixels / 8 patch rows, computed by the workgroup's waves
  // 0..3 only (pixel groups wm = 0, 1: one wave per SIMD, so each SIMD's MFMA pipe belongs to one wave and the K loop takes about half as
  // long), while waves 4..7 keep issuing their share of the LDS-DMA and meeting the barriers. Every output is still computed by ONE wave in
  // the usual K order, so results do not depend on where a tile falls in the walk (a split of K would: the sums of a batch and of its
  // images run alone would differ in the last bit; that variant, with a fence-free partial-sum exchange, was built and dropped in round 3).
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int Wp = g.W + 2, Hp = g.H + 2;
  const int PW = FLAT ? Wp : C3_PW2D;
  const int a_bytes = g.a_rows * 128;
  char* const sA = smem;                                   // 2 windows
  char* const sB = smem + 2 * a_bytes;                     // 3 weight strips
  float* const sbias = (float*)(smem + 2 * a_bytes + 3 * B_BYTES);   // the whole bias vector (padded to the N tile)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int srow = lane >> 3, sslot = lane & 7;
  const int a_groups = g.a_rows >> 3;
  const int l31 = lane & 31, fhalf = lane >> 5;
  const int fswB = (l31 >> 1) & 7;
  const long long ktot_bytes = 9LL * g.Ci * (long long)sizeof(T);
  const char* a_base = (const char*)g.in;
  const char* b_base = (const char*)g.wt;
  const int nchunks = g.Ci / BKE;

  for (int i = tid; i < g.tiles_n * BN; i += 512) sbias[i] = (g.bias && i < g.Co) ? g.bias[i] : 0.f;

  // Window and weight-strip staging: `global_load_lds_dwordx4 voff, s[base]` -- the per-lane 32-bit source offsets of a wave's
  // groups are tile-independent (computed once), the tile / chunk / tap enters through a scalar base: one VMEM instruction per KiB and
  // no VALU in the K loop (per-lane 64-bit pixel arithmetic and clamping cost ~5 VALU per step and 12 live VGPRs). Windows are
  // fetched WITHOUT clamping: reads past the bordered image (edge tiles) or before / behind the buffer (flat mode's first and last
  // tiles) land in the slack the ctx allocates around every activation buffer and only feed outputs that are never stored.
  struct Tile { int n0, img, y0, x0; long long q0; const char* ab; const char* bb;          // ab / bb: scalar bases of the window / the weight rows
                int rbase; };       // stacked tile rows: bordered row, inside its image, of the tile's first output row
  auto usg = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  auto spin = [&](const char* base, long long byte_off) -> const char* {                    // uniform pointer pinned to an SGPR pair
    const unsigned long long a = (unsigned long long)(uintptr_t)base + (unsigned long long)byte_off;
    const unsigned lo = usg((unsigned)a), hi = usg((unsigned)(a >> 32));
    return (const char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  };
  const int pix_bytes = g.in_pitch * (int)sizeof(T);
  uint32_t aoff[AG_MAX];
#pragma unroll
  for (int i = 0; i < AG_MAX; ++i) {
    int grp = wave + i * NW;
    if (grp > a_groups - 1) grp = a_groups - 1;            // every wave issues every slot (duplicates of the last group)
    const int r = grp * 8 + srow;
    int pix;
    if constexpr (FLAT) pix = r;
    else { const int i2 = r / C3_PW2D, j2 = r - i2 * C3_PW2D; pix = i2 * Wp + j2; }
    aoff[i] = (uint32_t)(pix * pix_bytes + ((sslot ^ ((r >> 1) & 7)) << 4));
  }
  uint32_t boff[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int row = (wave + i * NW) * 8 + srow;
    boff[i] = (uint32_t)((long long)row * ktot_bytes + ((sslot ^ ((row >> 1) & 7)) << 4));
  }
  auto setup = [&](long long lid, int half, Tile& t) {       // half: -1 = the whole tile, 0 / 1 = its first / second 128 pixels (flat only)
    int tn = (int)(lid % g.tiles_n);
    long long pt = lid / g.tiles_n;
#ifdef CTPN_ABLATION
    // CTPN_C3_P_ABL | 0x100 / 0x200 (measurement builds; CORRECT results: a bijection on the whole rounds of the walk): channel-slice-major
    // tile order per XCD for the Co = 512 layers. Default: the 32 workers of an XCD take 8 pixel tiles x all 4 slices per
    // round (a window is fetched by ONE XCD, the 4.7 MB of weights by every XCD every round). 0x100: an XCD takes ONE slice (two XCDs per
    // slice) x 32 pixel tiles; 0x200: TWO slices x 16 pixel tiles. tools/r5_order_ab.sh measures FETCH_SIZE and images/s of the three.
    if ((g.abl & 0x300) && g.tiles_n == 4 && G == 256 && lid < (g.ht_full / G) * G) {
      const int r = (int)(lid >> 8), p = (int)(lid & 255), xc = p >> 5, j = p & 31;
      if (g.abl & 0x100) { tn = xc & 3; pt = (long long)r * 64 + (xc >> 2) * 32 + j; }
      else { tn = 2 * (xc & 1) + (j & 1); pt = (long long)r * 64 + (xc >> 1) * 16 + (j >> 1); }
    }
#endif
    t.n0 = tn * BN; t.img = 0; t.y0 = 0; t.x0 = 0; t.q0 = 0; t.rbase = 0;
    long long pix0;
    if constexpr (FLAT) {
      t.q0 = pt * BM + (half > 0 ? BM / 2 : 0);
      pix0 = t.q0 - PW - 1;
    } else {
      const int per_img = g.stacked ? 0x7fffffff : g.tiles_x * g.tiles_y;       // stacked: one "image" = the whole bordered batch
      t.img = (int)(pt / per_img);
      const int rem = (int)(pt - (long long)t.img * per_img);
      const int tyi = rem / g.tiles_x;
      t.y0 = tyi * (C3_BM / TW) + (half > 0 ? (C3_BM / TW) / 2 : 0);      // (half items: 16 x 16 patches only, see HT)
      t.x0 = (rem - tyi * g.tiles_x) * C3_TW;
      pix0 = ((long long)t.img * Hp + t.y0) * Wp + t.x0;
      t.rbase = g.stacked ? (t.y0 + 1) % Hp : 0;
    }
    t.ab = spin(a_base, pix0 * pix_bytes);
    t.bb = spin(b_base, (long long)t.n0 * ktot_bytes);
  };
  auto issue_a_group = [&](int i, const char* ab, int chunk, int buf) {
    if constexpr (SPLIT) chunk = chunk >= g.a_wrap ? chunk - g.a_wrap : chunk;      // third K block: the hi plane again (scalar select)
    int grp = wave + i * NW;
    if (grp > a_groups - 1) grp = a_groups - 1;
    c3_glds16_saddr(ab + chunk * 128, aoff[i], __builtin_amdgcn_readfirstlane(lds0 + buf * a_bytes + grp * 1024));
  };
  // K steps of a chunk run KX-MAJOR: step t = (kx = t / 3, ky = t % 3), i.e. weight tap ky * 3 + kx. Consecutive steps then differ by one
  // input ROW, and the pixel fragments of the row two steps share stay in registers (see compute): the LDS read stream -- which at one
  // ds_read_b128 per MFMA and wave runs exactly at the CU's 128 B/clk -- loses a sixth (8 x 32 patches) / a twelfth (16 x 16) of its bytes.
  auto issue_b = [&](const char* bb, int chunk, int step_t, int buf) {
    const int tap = (step_t % 3) * 3 + step_t / 3;
    const char* sb = bb + ((long long)tap * g.Ci + (long long)chunk * BKE) * (long long)sizeof(T);
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      c3_glds16_saddr(sb, boff[i], __builtin_amdgcn_readfirstlane(lds0 + 2 * a_bytes + buf * B_BYTES + (wave + i * NW) * 1024));
  };

  c3_f32x16 acc[NTL][MT];
  int tilebase[MT];   // LDS row of (pixel tile j, lane) at tap (0,0)
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    if constexpr (FLAT) tilebase[j] = (wm * MT + j) * 32 + l31;
    else if constexpr (TW == 32) tilebase[j] = (wm * MT + j) * C3_PW2D + l31;
    else tilebase[j] = (2 * (wm * MT + j) + (l31 >> 4)) * C3_PW2D + c3_tw16_col(l31);
  }
  // Pixel fragments carried between the steps of a kx triple (4 k-slices x 16 bytes per lane):
  //   8 x 32 patches : pixel tile j of tap ky is window row 2 wm + j + ky, so tile 0 of step ky + 1 IS tile 1 of step ky: every step after
  //                    the first of a triple reads one pixel fragment per k-slice instead of two (4 row reads per triple instead of 6);
  //   16 x 16 patches: a tile is two window rows, tile 0 of ky = 2 IS tile 1 of ky = 0 (held across the ky = 1 step): 5 instead of 6;
  //   flat windows   : tiles are 32 consecutive pixels, a row shift of W + 2 pixels maps no tile onto another: nothing to carry.
  uint4 xcar[4];
  auto compute = [&](int abuf, int bbuf, auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int kx = t / 3, ky = t % 3;
    constexpr bool reuse0 = !FLAT && ((TW == 32 && ky > 0) || (TW == 16 && ky == 2));     // tile 0's fragments are the carried ones
    constexpr bool save1 = !FLAT && ((TW == 32 && ky < 2) || (TW == 16 && ky == 0));      // tile 1's fragments are carried on
    const int rowoff = ky * PW + kx;
    const char* sa = sA + abuf * a_bytes;
    const char* sb = sB + bbuf * B_BYTES + (wn * (BN / WGN) + l31) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = 2 * q + fhalf;
      uint4 xf[MT], wf[NTL];
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        if (reuse0 && j == 0) { xf[0] = xcar[q]; continue; }
        const int r = tilebase[j] + rowoff;
        xf[j] = *(const uint4*)(sa + r * 128 + ((slot ^ ((r >> 1) & 7)) << 4));
      }
      if constexpr (save1) xcar[q] = xf[1];
#pragma unroll
      for (int i = 0; i < NTL; ++i) wf[i] = *(const uint4*)(sb + i * 32 * 128 + ((slot ^ fswB) << 4));
#pragma unroll
      for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) c3_mfma<T>(acc[i][j], wf[i], xf[j]);
    }
  };

  // AHEAD: the fragment reads run ONE k-slice group ahead of the MFMAs through a second register set. hipcc on its own issues a group's
  // four reads right behind the previous group's last MFMA and makes that group's first MFMA wait for them: with the two waves of a SIMD
  // kept in step by the barriers, both sit in that LDS latency together (MFMA pipe 60 - 73 % busy). Here group (t, q) is READ while the
  // MFMAs of the group before it (the last group of step t - 1 for q = 0: it runs behind the barrier, on fragments read in front of it)
  // are issued, and its own MFMAs come one group later; sched_group_barriers order every [reads of group n + 1][MFMAs of group n] block, a sched_barrier closes it.
  struct Frag { uint4 x[MT], w[NTL]; };
  auto load_group = [&](int abuf, int bbuf, auto tc, auto qc, Frag& f) {
    constexpr int t = decltype(tc)::value, q = decltype(qc)::value;
    constexpr int kx = t / 3, ky = t % 3;
    constexpr bool reuse0 = !FLAT && ((TW == 32 && ky > 0) || (TW == 16 && ky == 2));
    constexpr bool save1 = !FLAT && ((TW == 32 && ky < 2) || (TW == 16 && ky == 0));
    const int rowoff = ky * PW + kx;
    const char* sa = sA + abuf * a_bytes;
    const char* sb = sB + bbuf * B_BYTES + (wn * (BN / WGN) + l31) * 128;
    const int slot = 2 * q + fhalf;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      if (reuse0 && j == 0) { f.x[0] = xcar[q]; continue; }
      const int r = tilebase[j] + rowoff;
      f.x[j] = *(const uint4*)(sa + r * 128 + ((slot ^ ((r >> 1) & 7)) << 4));
    }
    if constexpr (save1) xcar[q] = f.x[1];
#pragma unroll
    for (int i = 0; i < NTL; ++i) f.w[i] = *(const uint4*)(sb + i * 32 * 128 + ((slot ^ fswB) << 4));
  };
  auto mma_group = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) c3_mfma<T>(acc[i][j], f.w[i], f.x[j]);
  };
  Frag pend;

  // k-th work item of this worker: whole tiles w0 + k G below ht_full, then (flat windows only) at most one half of a tail tile; selects
  // on wave-uniform scalars and ONE setup() per item. Past the end the current item is returned again (its window is prefetched once more:
  // every step issues the same loads).
  Tile cur, nxt;
  long long lid = w0;             // index of the current item in the walk w0, w0 + G, ...
  long long cur_tile = w0; int cur_half = -1;
  bool active = true;             // does this wave compute in the current item? (waves 4..7 sit out the half items)
  // HT: kernels that split their tail tiles. Flat windows (a half = 128 consecutive pixels) and 16 x 16 patches (a half = 8 rows x 16: the
  // tile origin moves down by 8 rows, pixel groups 0 and 1 are exactly those rows). NOT the 8 x 32-patch kernels: they share their CUs
  // with the one-wave edge kernel, and the idle path costs this kernel ~30 registers (2 x 235 + 74 > a SIMD's 512).
  constexpr bool HT = ((FLAT || TW == 16) && BM == 256) || HT32;
  auto pick = [&](long long l, long long& tile, int& half) -> bool {
    if constexpr (HT) {
      const long long o = l - g.ht_full;
      const bool whole = l < g.ht_full, half_item = !whole && o < 2LL * g.ht_r;
      tile = whole ? l : (half_item ? g.ht_full + (o >> 1) : cur_tile);
      half = (int)usg((unsigned)(whole ? -1 : (half_item ? (int)(o & 1) : cur_half)));
      return whole || half_item;
    } else {
      const bool ok = l < total;
      tile = ok ? l : cur_tile; half = -1;
      return ok;
    }
  };
  (void)pick(w0, cur_tile, cur_half);
  setup(cur_tile, cur_half, cur);
  if constexpr (HT) active = cur_half < 0 || wm < 2;
  // the only exposed prologue of the launch
#pragma unroll
  for (int i = 0; i < AG_MAX; ++i) issue_a_group(i, cur.ab, 0, 0);
  issue_b(cur.bb, 0, 0, 0);
  issue_b(cur.bb, 0, 1, 1);
  c3_wait_vm<B_LOADS>();
  __syncthreads();            // also publishes sbias
  int wpar = 0;               // window buffer of the current chunk

  // One K step. Step s = 9 * chunk + tap reads strip buffer tap % 3; the strip for step s + 2 and the slices of the NEXT
  // chunk's window are issued first. `last` = last chunk of the tile: "next chunk" is chunk 0 of the next tile.
  auto step = [&](auto tc, auto lastc, int c) {
    constexpr int t = decltype(tc)::value;
    constexpr bool last = decltype(lastc)::value;
    constexpr int nA = (t < 8 && AG_MAX > t) ? (AG_MAX - t + 7) / 8 : 0;   // slices in steps 0..7 only (see conv3x3_kernel)
    if constexpr (t + 2 < 9) issue_b(cur.bb, c, t + 2, (t + 2) % 3);
    else if constexpr (last) issue_b(nxt.bb, 0, t + 2 - 9, (t + 2) % 3);
    else issue_b(cur.bb, c + 1, t + 2 - 9, (t + 2) % 3);
#pragma unroll
    for (int i = t; i < (t < 8 ? AG_MAX : 0); i += 8) {
      if constexpr (last) issue_a_group(i, nxt.ab, 0, wpar ^ 1);
      else issue_a_group(i, cur.ab, c + 1, wpar ^ 1);
    }
    if constexpr (!AHEAD) {
      compute(wpar, t % 3, tc);
    } else {
      constexpr int nrd = (MT + NTL) - ((!FLAT && ((TW == 32 && t % 3 > 0) || (TW == 16 && t % 3 == 2))) ? 1 : 0);   // LDS reads of one group of this step
      constexpr int nmf = ((int)sizeof(T) == 2 ? 1 : 4) * MT * NTL;                                                  // MFMA instructions of one group
      Frag nf0, nf1, nf2, nf3;
      load_group(wpar, t % 3, tc, std::integral_constant<int, 0>{}, nf0);
      if (t > 0 || c > 0) mma_group(pend);             // the last group of the previous step (none in front of a tile's first step)
      __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, nmf, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_group(wpar, t % 3, tc, std::integral_constant<int, 1>{}, nf1);
      mma_group(nf0);
      __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, nmf, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_group(wpar, t % 3, tc, std::integral_constant<int, 2>{}, nf2);
      mma_group(nf1);
      __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, nmf, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_group(wpar, t % 3, tc, std::integral_constant<int, 3>{}, nf3);
      mma_group(nf2);
      __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, nmf, 0);
      __builtin_amdgcn_sched_barrier(0);
      pend = nf3;
    }
    // INVARIANT (round 6: EVERY path): no LDS read may be in flight across the barrier below. Behind it, step t + 1 aims its LDS-DMA
    // at the strip buffer (t % 3) and -- in the chunk's last step -- at the window this step read; gfx950's barrier does not imply an lgkmcnt
    // wait. AHEAD: nf3's ds_reads (-> pend) were issued in front of the step's last four MFMAs and have had ~128 clk to land: free. The plain
    // path: hipcc sinks the last k-slice group's MFMAs (and the wait for their operands) below the barrier, so without this wait that group's
    // reads were ordered against the DMA by latency only -- and lost: with another stream's kernel loading the memory system, a strip landed
    // before the reads of the step three earlier had executed (profiles/r06_barrier_war.txt: the last k-slice of one step computed with the
    // weights of the step three later, bit for bit).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    c3_wait_vm<B_LOADS + nA>();
    c3_barrier();
  };
  auto chunk = [&](auto lastc, int c) {
    step(std::integral_constant<int, 0>{}, lastc, c);
    step(std::integral_constant<int, 1>{}, lastc, c);
    step(std::integral_constant<int, 2>{}, lastc, c);
    step(std::integral_constant<int, 3>{}, lastc, c);
    step(std::integral_constant<int, 4>{}, lastc, c);
    step(std::integral_constant<int, 5>{}, lastc, c);
    step(std::integral_constant<int, 6>{}, lastc, c);
    step(std::integral_constant<int, 7>{}, lastc, c);
    step(std::integral_constant<int, 8>{}, lastc, c);
    wpar ^= 1;
  };
  // A chunk of a wave that sits out a half item (flat windows): its share of the LDS-DMA, the counted wait and the barrier of every step --
  // exactly the loads `step` issues, no MFMAs. A compact runtime loop ON PURPOSE: guarding compute() inside `step` instead put a branch
  // into every K step, which stopped hipcc from pipelining across the step boundaries (measured: the flat kernel +9 %, and that build
  // failed the run-twice-same-bytes test -- the per-step control flow let fragment reads move relative to the counted waits).
  auto idle_chunk = [&](bool last, int c) {
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      const int t2 = t + 2 < 9 ? t + 2 : t + 2 - 9;
      const char* const bsrc = (t + 2 < 9 || !last) ? cur.bb : nxt.bb;
      const int bchunk = t + 2 < 9 ? c : (last ? 0 : c + 1);
      issue_b(bsrc, bchunk, t2, (t + 2) % 3);
#pragma unroll
      for (int i = 0; i < AG_MAX; ++i)
        if (t < 8 && (i & 7) == t) {
          if (last) issue_a_group(i, nxt.ab, 0, wpar ^ 1);
          else issue_a_group(i, cur.ab, c + 1, wpar ^ 1);
        }
      static_assert(AG_MAX <= 8, "at most one window slice per step, in steps 0 .. AG_MAX - 1");
      if (t < 8 && t < AG_MAX) c3_wait_vm<B_LOADS + 1>(); else c3_wait_vm<B_LOADS>();
      c3_barrier();
    }
    wpar ^= 1;
  };

  for (;;) {
    const long long nlid = lid + G;
    long long nxt_tile; int nxt_half;
    const bool has_next = pick(nlid, nxt_tile, nxt_half);
    setup(nxt_tile, nxt_half, nxt);
    // accumulators start from the bias (read from LDS straight into the accumulator registers): no bias add in the epilogue, and
    // max-pooling the sums commutes with it
    {
      const float* bl = sbias + cur.n0 + wn * (BN / WGN) + 4 * fhalf;
#pragma unroll
      for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const c3_f32x4 bv = *(const c3_f32x4*)(bl + i * 32 + 8 * g4);
#pragma unroll
          for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * g4 + e] = bv[e];
        }
    }
    bool idle_done = false;
    if constexpr (HT) {
      if (!active) {
        for (int c = 0; c + 1 < nchunks; ++c) idle_chunk(false, c);
        idle_chunk(true, nchunks - 1);
        idle_done = true;
      }
    }
    if (!idle_done) {
      for (int c = 0; c + 1 < nchunks; ++c) chunk(std::false_type{}, c);
      chunk(std::true_type{}, nchunks - 1);
      if constexpr (AHEAD) mma_group(pend);      // the tile's last group
    }

    // ---- epilogue from registers: lane owns channels 8 g4 + 4 fhalf .. + 3 of pixel l31 of each (i, j) tile ----
    // ReLU (always on in this network: the launcher sends relu == 0 to the non-persistent kernel) on the packed bf16 pairs as an
    // integer max (sign bit set <=> negative), fp32: fmaxf. Addresses: 64-bit tile base on the scalar unit + a 32-bit lane
    // offset (per-lane 64-bit pixel arithmetic with quarter-rate v_mul_lo_u32 / v_mad_u64_u32 was a third of this epilogue).
#ifdef CTPN_ABLATION
    // timing-only ablations (CTPN_C3_P_ABL, -DCTPN_ABLATION builds only). CAVEAT (round 3): a layer that does not store feeds ZEROS to the
    // next one (the activation buffers start zeroed), and MFMAs on zero operands draw less power -- the whole stack then clocks ~15 %
    // higher. Compare CYCLES (GRBM_GUI_ACTIVE, tools/r3_pmc_epi.sh), not time: by cycles the stores cost 0-3 % of a layer, not the
    // 17 % of the stack round 2 read off the clock.
    const bool do_epi = g.abl != 1;
    const bool st_on = g.abl != 2;
#else
    constexpr bool st_on = true, do_epi = true;
#endif
    typedef short c3_s16x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) char* c3_gptr;
#ifdef CTPN_ABLATION
    const c3_gptr hot = (c3_gptr)(uintptr_t)(g.out ? g.out : g.pool_out);
#endif
    auto relu_pk = [](uint32_t p) -> uint32_t {
      const c3_s16x2 z = {0, 0};
      return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(c3_s16x2, p), z));
    };
    // fp32 sums -> one packed 16-bit pair. PL = 0: the 16-bit modes (convert, ReLU as an integer max on the pair); SPLIT: ReLU in fp32,
    // then PL = 1: the hi plane RNE_bf16(v), PL = 2: the lo plane RNE_bf16(v - hi) (may be negative: no ReLU on the packed pair)
    auto pack = [&](auto plc, float x, float y) -> uint32_t {
      constexpr int PL = decltype(plc)::value;
      if constexpr (PL == 0) {
        if constexpr (sizeof(OutT) == 2) return relu_pk(c3_cvt_pk<OutT>(x, y)); else return 0u;      // (fp32 kernels never pack)
      } else {
        x = __builtin_fmaxf(x, 0.f); y = __builtin_fmaxf(y, 0.f);
        uint32_t hi, lo;
        ctpn_split_pk_bf16(x, y, hi, lo);
        return PL == 1 ? hi : lo;
      }
    };
    const int opitch = SPLIT ? g.out_pitch : g.Co;       // 16-bit elements per output pixel
    // 16-byte stores of 8 (bf16) / 4 (fp32) consecutive channels of accumulator tile a; dst = this lane's pixel, first channel of the tile
    auto store_tile_pl = [&](auto plc, c3_gptr dst, bool ok, const c3_f32x16& a) {
      if constexpr (sizeof(OutT) == 4) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const c3_f32x4 o = {__builtin_fmaxf(a[4 * g4], 0.f), __builtin_fmaxf(a[4 * g4 + 1], 0.f), __builtin_fmaxf(a[4 * g4 + 2], 0.f), __builtin_fmaxf(a[4 * g4 + 3], 0.f)};
          if (ok) *(__attribute__((address_space(1))) c3_f32x4*)(dst + (8 * g4 + 4 * fhalf) * 4) = o;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t e0 = pack(plc, a[8 * q + 0], a[8 * q + 1]), e1 = pack(plc, a[8 * q + 2], a[8 * q + 3]);
          const uint32_t o0 = pack(plc, a[8 * q + 4], a[8 * q + 5]), o1 = pack(plc, a[8 * q + 6], a[8 * q + 7]);
          const auto r0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);   // low lanes: even group complete, high lanes: odd group
          const auto r1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
          const c3_u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
#ifdef CTPN_ABLATION
          if (g.abl == 5) dst = hot;            // timing only: the same store instructions, all aimed at one hot KiB (no write stream)
#endif
          if (ok && st_on) *(__attribute__((address_space(1))) c3_u32x4*)(dst + (16 * q + 8 * fhalf) * 2) = v;
        }
      }
    };
    // bf16, two pixels per lane: after the two 16-byte pieces (q = 0, 1) of a lane's pixel are formed, v_permlane16_swap exchanges
    // piece 1 of lanes r with piece 0 of lanes r + 16, so that store A carries pixels 0..15 of the tile row and store B pixels 16..31,
    // FOUR lanes (64 contiguous bytes) per pixel instead of two: a store instruction touches 16 lines instead of 32 (the texture
    // path's cost per store is its number of distinct lines -- conv1_1 went from 0.61 to 0.45 ms on exactly that). Lane L stores
    // piece 2 * ((L >> 4) & 1) + (L >> 5) of pixel (L & 15) [A] / 16 + (L & 15) [B]; dstA / dstB: that pixel, first channel of tile a.
    const int piece_off = (2 * ((lane >> 4) & 1) + fhalf) * 16;
    auto store_pair_pl = [&](auto plc, c3_gptr dstA, bool okA, c3_gptr dstB, bool okB, bool haveB, const c3_f32x16& a) {
      c3_u32x4 v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint32_t e0 = pack(plc, a[8 * q + 0], a[8 * q + 1]), e1 = pack(plc, a[8 * q + 2], a[8 * q + 3]);
        const uint32_t o0 = pack(plc, a[8 * q + 4], a[8 * q + 5]), o1 = pack(plc, a[8 * q + 6], a[8 * q + 7]);
        const auto r0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
        v[q] = c3_u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
      c3_u32x4 va, vb;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const auto r = __builtin_amdgcn_permlane16_swap(v[0][c], v[1][c], false, false);
        va[c] = r[0]; vb[c] = r[1];
      }
#ifdef CTPN_ABLATION
      if (g.abl == 5) { dstA = hot; dstB = hot; }
#endif
      if (okA && st_on) *(__attribute__((address_space(1))) c3_u32x4*)(dstA + piece_off) = va;
      if (haveB && okB && st_on) *(__attribute__((address_space(1))) c3_u32x4*)(dstB + piece_off) = vb;
    };
    // one store per plane: the 16-bit modes write their single plane; SPLIT writes hi at the channel, lo at Co + channel (and hi once more
    // at 2 Co + channel when the consumer is the LSTM projection GEMM)
    const size_t plane_b = (size_t)g.Co * 2;
    auto store_tile = [&](c3_gptr dst, bool ok, const c3_f32x16& a) {
      if constexpr (!SPLIT) store_tile_pl(std::integral_constant<int, 0>{}, dst, ok, a);
      else {
        store_tile_pl(std::integral_constant<int, 1>{}, dst, ok, a);
        store_tile_pl(std::integral_constant<int, 2>{}, dst + plane_b, ok, a);
        if (g.dup_hi) store_tile_pl(std::integral_constant<int, 1>{}, dst + 2 * plane_b, ok, a);
      }
    };
    auto store_pair = [&](c3_gptr dstA, bool okA, c3_gptr dstB, bool okB, bool haveB, const c3_f32x16& a) {
      if constexpr (!SPLIT) store_pair_pl(std::integral_constant<int, 0>{}, dstA, okA, dstB, okB, haveB, a);
      else {
        store_pair_pl(std::integral_constant<int, 1>{}, dstA, okA, dstB, okB, haveB, a);
        store_pair_pl(std::integral_constant<int, 2>{}, dstA + plane_b, okA, dstB + plane_b, okB, haveB, a);
        if (g.dup_hi) store_pair_pl(std::integral_constant<int, 1>{}, dstA + 2 * plane_b, okA, dstB + 2 * plane_b, okB, haveB, a);
      }
    };
    auto usgpr = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    auto gbase = [&](const void* base, unsigned pix, int ch) -> c3_gptr {     // uniform: base + (pix * Co + ch) * sizeof(OutT), pinned to SGPRs
      const unsigned long long a = (unsigned long long)(uintptr_t)base + ((unsigned long long)pix * (unsigned)opitch + (unsigned)ch) * sizeof(OutT);
      const unsigned lo = usgpr((unsigned)a), hi = usgpr((unsigned)(a >> 32));
      return (c3_gptr)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    const int ch0 = cur.n0 + wn * (BN / WGN);                                  // first channel of this wave
    // lane-dependent address terms are recomputed per tile ON PURPOSE: as loop invariants hipcc hoists them out of the tile loop,
    // where they stay live across the K loop -- at 256 VGPRs that means scratch reloads (and their vmcnt(0)) inside the load pipeline
    int lq = l31;
    asm volatile("" : "+v"(lq));
    if (g.out && do_epi && (!HT || active)) {
      if constexpr (FLAT) {
        char* ob = (char*)g.out;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const long long q = cur.q0 + (wm * MT + j) * 32 + l31;
          const long long per = (long long)Hp * Wp;
          const long long im = q / per;
          const int rem = (int)(q - im * per);
          const int yb = rem / Wp, xb = rem - yb * Wp;
          const bool ok = q < g.m_total && yb >= 1 && yb <= g.H && xb >= 1 && xb <= g.W;
#pragma unroll
          for (int i = 0; i < NTL; ++i) {
            const int co = ch0 + i * 32;
            store_tile((c3_gptr)(uintptr_t)(ob + (q * opitch + co) * (long long)sizeof(OutT)), ok && co < g.Co, acc[i][j]);
          }
        }
      } else {
        // is output row `prow` of the tile an interior row? per image: below H. Stacked tile rows (a tile spans at most two images:
        // the launcher stacks only for H >= 16): the bordered row inside its image must be 1 .. H, and the row must lie inside the batch
        auto rowok = [&](int prow) -> bool {
          if (!g.stacked) return cur.y0 + prow < g.H;
          int rr = cur.rbase + prow;
          rr = rr >= Hp ? rr - Hp : rr;
          return rr >= 1 && rr <= g.H && cur.y0 + 1 + prow < g.N * Hp;
        };
        // tile origin on the scalar unit; this lane's pixel inside the tile: row prow(j), column lcol
        const unsigned tpix = usgpr((unsigned)((cur.img * Hp + cur.y0 + 1) * Wp + cur.x0 + 1));
        const c3_gptr tb = gbase(g.out, tpix, ch0);
        if constexpr (sizeof(OutT) == 2) {
          // pixels (lq & 15) [store A] and 16 + (lq & 15) [store B] of tile row j: 8 x 32 patches: same row, 16 columns apart;
          // 16 x 16 patches: rows 2 jj and 2 jj + 1 (the second row's lane order is rotated, c3_tw16_col)
          const int l15 = lq & 15;
          const int colA = l15, colB = TW == 32 ? 16 + l15 : c3_tw16_col(16 + l15);
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const int prowA = TW == 32 ? wm * MT + j : 2 * (wm * MT + j), prowB = TW == 32 ? prowA : prowA + 1;
            const bool okA = rowok(prowA) && cur.x0 + colA < g.W, okB = rowok(prowB) && cur.x0 + colB < g.W;
            const uint32_t offA = (uint32_t)((prowA * Wp + colA) * opitch) * 2u, offB = (uint32_t)((prowB * Wp + colB) * opitch) * 2u;
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
              const bool cok = ch0 + i * 32 < g.Co;
              store_pair(tb + (size_t)offA + i * 64, okA && cok, tb + (size_t)offB + i * 64, okB && cok, true, acc[i][j]);
            }
          }
        } else {
        const int lcol = TW == 32 ? lq : c3_tw16_col(lq);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int prow = TW == 32 ? wm * MT + j : 2 * (wm * MT + j) + (lq >> 4);
          const bool ok = rowok(prow) && cur.x0 + lcol < g.W;
          const uint32_t loff = (uint32_t)((prow * Wp + lcol) * opitch) * (uint32_t)sizeof(OutT);
#pragma unroll
          for (int i = 0; i < NTL; ++i)
            store_tile(tb + (size_t)loff + i * 32 * (int)sizeof(OutT), ok && ch0 + i * 32 < g.Co, acc[i][j]);
        }
        }
      }
    }
    if constexpr (POOL) if (do_epi && (!HT || active)) {
      // max commutes with the bias (already in the sums), ReLU and the rounding: pool the sums. Vertical partner: the wave's other pixel
      // row (8 x 32 patches: same lane of tile j = 1) or a ds_bpermute partner (16 x 16 patches: a tile is two rows of 16); horizontal
      // partner: lane ^ 1. Lanes 2k / 2k+1 share a pooled pixel: the even one keeps channel tile 0, the odd one tile 1.
      const int Ho = g.H >> 1, Wo = g.W >> 1;
      const bool odd = (lq & 1) != 0;
      const unsigned ppix = usgpr((unsigned)((cur.img * (Ho + 2) + (cur.y0 >> 1) + 1) * (Wo + 2) + (cur.x0 >> 1) + 1));
      const c3_gptr pb = gbase(g.pool_out, ppix, ch0);
      auto hpool = [&](const c3_f32x16& v0, const c3_f32x16& v1, int Yl, int Xl, bool keep) {
        // v0 / v1: vertically pooled sums of channel tile 0 / 1 for this lane's column; (Yl, Xl): pooled pixel inside the tile
        c3_f32x16 mine;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // own / send exactly as before the rewrite: ONE cross-lane move per element, made opaque right away. (Written as
          // max(v, dpp(v)) for both channel tiles with the select afterwards, hipcc sank the whole computation into the store's
          // exec-masked block and kept a single DPP move for all 16 elements: wrong pooled values, caught by the parity tests.)
          const float own = odd ? v1[r] : v0[r], send = odd ? v0[r] : v1[r];
          float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
          asm volatile("" : "+v"(recv));
          mine[r] = __builtin_fmaxf(own, recv);
        }
        const int co = ch0 + (odd ? 32 : 0);
        if constexpr (sizeof(OutT) == 2) {
          // lanes 2k / 2k+1 hold channel tile 0 / 1 of pooled pixel k: with the 16-lane exchange (store_pair) store A writes pooled pixels
          // 0..7 of the row as full 128-byte lines (8 lanes each), store B pixels 8..15 (8 x 32 patches; a 16 x 16 patch has 8 pooled
          // pixels per row pair and its lanes 16..31 hold nothing to store: ONE store instead of two half-empty ones)
          (void)Xl; (void)keep;
          const int xa = (lq & 15) >> 1, xb = 8 + xa;
          const bool rowok = (cur.y0 >> 1) + Yl < Ho && co < g.Co;
          const uint32_t offA = (uint32_t)((Yl * (Wo + 2) + xa) * opitch + (odd ? 32 : 0)) * 2u;
          store_pair(pb + (size_t)offA, rowok && (cur.x0 >> 1) + xa < Wo, pb + (size_t)offA + (size_t)(8 * opitch) * 2u,
                     rowok && (cur.x0 >> 1) + xb < Wo, TW == 32, mine);
        } else {
          const uint32_t loff = (uint32_t)((Yl * (Wo + 2) + Xl) * opitch + (odd ? 32 : 0)) * (uint32_t)sizeof(OutT);
          store_tile(pb + (size_t)loff, keep && (cur.y0 >> 1) + Yl < Ho && (cur.x0 >> 1) + Xl < Wo && co < g.Co, mine);
        }
      };
      if constexpr (NTL == 1) {
        // ONE channel tile per wave: lanes 2k / 2k + 1 end up with the same 32 channels of pooled pixel k; the even one stores them (store A:
        // pooled pixels 0..7 of the row, four lanes = 64 contiguous bytes each; store B: pixels 8..15)
        c3_f32x16 mine;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = __builtin_fmaxf(acc[0][0][r], acc[0][1][r]);
          float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
          asm volatile("" : "+v"(recv));          // (see hpool: one cross-lane move per element, opaque right away)
          mine[r] = __builtin_fmaxf(v, recv);
        }
        if constexpr (sizeof(OutT) == 2) {
          const int xa = (lq & 15) >> 1, xb = 8 + xa;
          const bool rok = (cur.y0 >> 1) + wm < Ho && ch0 < g.Co && !odd;
          const uint32_t offA = (uint32_t)((wm * (Wo + 2) + xa) * opitch) * 2u;
          store_pair(pb + (size_t)offA, rok && (cur.x0 >> 1) + xa < Wo, pb + (size_t)offA + (size_t)(8 * opitch) * 2u, rok && (cur.x0 >> 1) + xb < Wo, true, mine);
        } else {
          const int Xl = lq >> 1;
          const uint32_t loff = (uint32_t)((wm * (Wo + 2) + Xl) * opitch) * (uint32_t)sizeof(OutT);
          store_tile(pb + (size_t)loff, !odd && (cur.y0 >> 1) + wm < Ho && (cur.x0 >> 1) + Xl < Wo && ch0 < g.Co, mine);
        }
      } else if constexpr (TW == 32) {
        c3_f32x16 v0, v1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { v0[r] = __builtin_fmaxf(acc[0][0][r], acc[0][1][r]); v1[r] = __builtin_fmaxf(acc[NTL - 1][0][r], acc[NTL - 1][1][r]); }
        hpool(v0, v1, wm, lq >> 1, true);
      } else {
        // vertical partner under the rotated lane order (c3_tw16_col): row 0 lane c <-> row 1 lane 16 + ((c + 2) & 15)
        const int vpart = ((lane & 32) | ((lq & 16) ? ((lq - 2) & 15) : 16 + ((lq + 2) & 15))) << 2;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          c3_f32x16 v0, v1;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float a0 = acc[0][j][r], a1 = acc[NTL - 1][j][r];
            float b0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(vpart, __builtin_bit_cast(int, a0)));   // same column, other row
            float b1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(vpart, __builtin_bit_cast(int, a1)));
            asm volatile("" : "+v"(b0), "+v"(b1));          // cross-lane results pinned outside the store's exec-masked block (see hpool)
            v0[r] = __builtin_fmaxf(a0, b0); v1[r] = __builtin_fmaxf(a1, b1);
          }
          hpool(v0, v1, wm * MT + j, (lq & 15) >> 1, (lq & 16) == 0);
        }
      }
    }
    if (!has_next) break;
    lid = nlid;
    cur = nxt;
    cur_tile = nxt_tile; cur_half = nxt_half;
    if constexpr (HT) active = cur_half < 0 || wm < 2;
  }
  c3_wait_vm<0>();   // the dummy prefetch of the last tile
}

// ---------------------------------------------------------------------------------------------
// Weights-in-REGISTERS persistent kernel for the Ci = 64 layers in bf16 (conv1_2: 64 -> 64 + pool, conv2_1: 64 -> 128).
// Round 1's weights-stationary kernel (nine weight strips in LDS; removed in round 3) spent ~21 instructions per MFMA (address
// arithmetic for 144 swizzled fragment reads and 11 window pieces per tile, 186 accvgpr copies): issue-bound at 51 % MFMA busy. Here:
//   * a workgroup (4 waves, one per SIMD, 512 registers each) owns 64 output channels and walks 8 x 32-pixel tiles; wave
//     (ph, ch) computes pixel rows 4 ph .. 4 ph + 3 x channels 32 ch .. + 31: all 36 weight fragments of its 32 channels
//     (9 taps x 4 k-slices x 16 B per lane = 144 VGPRs) are loaded ONCE and stay in registers -- no weight traffic in LDS at all;
//   * the LDS holds only input windows, three of them, with a PADDED 144-byte pixel pitch instead of the XOR swizzle: bank
//     group = (9 row + slot) mod 16 is conflict-free for the ds_read_b128 lane groups and, unlike the swizzle, AFFINE -- every
//     fragment read of a tile is `ds_read_b128 v, vbase offset:imm` off ONE address register;
//   * the window pieces are `global_load_lds_dwordx4 voff, s[base]`: the per-lane source offsets of a wave's 12 pieces are
//     tile-independent (computed once), the tile enters through a scalar base -- one VMEM instruction per KiB, no VALU. Windows
//     are fetched WITHOUT clamping at the image edge: reads past the last bordered row / image run into the next rows / the slack
//     the ctx allocates behind every activation buffer; those window pixels only feed outputs that are never stored;
//   * the K loop is ordered by INPUT row: fragment (row r, kx, k-slice q) is read once and feeds every (output row j, ky) with
//     j + ky = r: 72 reads for 144 MFMAs per wave and tile (was 144), issued PD = 8 slots ahead through a register ring with
//     counted lgkmcnt, across tile boundaries;
//   * the epilogue of tile k (bias, ReLU, 2 x 2 pool via DPP, bf16 pack, 16-byte stores) runs on a second accumulator set,
//     interleaved piece by piece with the MFMAs of tile k + 1; window k + 2 is issued inside the same stream;
//   * tiles are CLAIMED, not statically partitioned: a workgroup's first five tiles are fixed (worker + i * nworkers), every later
//     one comes from a device-scope atomic counter, fetched by one lane five tiles ahead and handed to the other waves through
//     an LDS word behind the regular tile barrier. The proposal-stream kernels of the previous batch (sort, NMS: one 1024-thread
//     workgroup per image for up to a millisecond) share the GPU with conv1_2 / conv2_1 of the next batch, and a persistent
//     workgroup that needs a whole CU (147 KB of LDS, 432 registers per lane) cannot start on a CU an NMS workgroup occupies:
//     with a static partition those late starters still had their full share to do and the launch ended ~0.3 ms late;
//   * ONE s_barrier per tile, PD slots into it: by then every wave has drained its reads of window k - 1 (buffer of k + 2)
//     and `vmcnt(0)` there covers window k + 1 (issued a whole tile earlier) -- no counted vmcnt, no dump page.
// ---------------------------------------------------------------------------------------------
struct Conv3WR {
  const void* in; const void* wt; const float* bias; void* out; void* pool_out;
  int N, H, W, Co;
  int tiles_x, tiles_y, tiles_n;
  unsigned ptiles;                 // N * tiles_x * tiles_y
  unsigned groups, per_group;      // tile ranges: workgroup b belongs to group b % groups (8 = one per XCD: the hardware deals consecutive
                                   // workgroup ids round-robin over the XCDs) and walks tiles [grp * per_group, min(.. + per_group, ptiles))
  unsigned magic_img, magic_row;   // floor(2^32 / d) + 1 for d = tiles_x * tiles_y and d = tiles_x (exact for pt * d < 2^32)
  char* dump;                      // 4 KB per workgroup: where lanes outside the image store, so that every wave issues the same number of stores
  unsigned* claim;                 // [groups][tiles_n][2] = {tiles handed out beyond the static ones, workgroups that have finished}; zero between launches
  // FUSE (conv1_2 with conv1_1 computed in its window stage): the q-image of the batch (common.h), its geometry, conv1_1's fragments
  const void* q; const void* wfq;
  int Hq, Wq;
};

constexpr int WR_PITCH = 144, WR_PW = 34, WR_ROWS = 10 * WR_PW, WR_PIECES = 48, WR_WIN = WR_PIECES * 1024, WR_NBUF = 3, WR_PD = 8;
static_assert(WR_ROWS * WR_PITCH <= WR_WIN, "window must fit its pieces");

template <int... I, typename F>
__device__ __forceinline__ void c3_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void c3_static_for(F&& f) { c3_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// "+v" / "+a": the destination is declared read-write although the instruction only writes it. That ties every new value to
// the register of the old one, so ring slots and accumulators stay IN PLACE across the tile loop's back edge; as plain
// outputs the register allocator gave each definition a fresh register and glued the loop together with 128 v_accvgpr_mov +
// 32 v_mov per iteration.
template <int OFF, bool XA = false>
__device__ __forceinline__ void c3_ds_read_b128_off(c3_u32x4& dst, uint32_t lds_addr) {
  if constexpr (XA) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+a"(dst) : "v"(lds_addr), "n"(OFF));
  else asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(lds_addr), "n"(OFF));
}
// One K slot of the weights-in-registers kernel as ONE asm block: wait for the ring's oldest fragment x, run the slot's 1..3 MFMAs
// on it (one per output row it feeds), refill the ring slot with the fragment PD slots ahead. Register files: the 36 weight
// fragments live in AGPRs (MFMA A operand), the two accumulator sets (128), the bias vector and the ring in VGPRs, tied in place
// ("+v") -- so the epilogue is plain C++ on accumulator elements. (Accumulators in AGPRs needed a v_accvgpr_read per element from
// asm, whose "a" input hipcc sometimes fed with a v_accvgpr_write right in front of it: a hazard it cannot see into the asm for.) INIT = index
// of the MFMA that starts its accumulator's chain for this tile (C operand = the bias vector), -1 = none.
// Hazards the compiler no longer sees, all satisfied by construction: a dependent MFMA on exactly the same accumulator (same
// opcode) is interlocked by the hardware; x comes from LDS behind the block's own s_waitcnt, the weights were loaded once at
// kernel start; the ds_read overwrites x, an A/B operand of MFMAs issued before it (in-order issue; only SrcC has a WAR window);
// the VALU reads an accumulator set (v_accvgpr_read in the epilogue pieces) no earlier than PD + 1 slots after its last MFMA
// and no later than 9 slots before its next one.
// XC: register file of the ring fragments x -- "+v", or "+a" in the fused kernel (ds_read_b128 loads accumulation registers as well, and
// an MFMA takes either operand from them): its 32 ring registers move out of the VGPR file to make room for the producer
#define C3_DEFINE_SLOTS(SFX, MN, XC) \
template <int OFF, int WAIT, int INIT> \
__device__ __forceinline__ void c3_slot1##SFX(c3_f32x16& a0, const c3_u32x4& w0, c3_u32x4& x, uint32_t xaddr, const c3_f32x16& bias) { \
  if constexpr (INIT == 0) \
    asm volatile("s_waitcnt lgkmcnt(%6)\n\t" MN "%0, %2, %1, %3\n\tds_read_b128 %1, %4 offset:%5" \
                 : "+v"(a0), XC(x) : "a"(w0), "v"(bias), "v"(xaddr), "n"(OFF), "n"(WAIT)); \
  else \
    asm volatile("s_waitcnt lgkmcnt(%5)\n\t" MN "%0, %2, %1, %0\n\tds_read_b128 %1, %3 offset:%4" \
                 : "+v"(a0), XC(x) : "a"(w0), "v"(xaddr), "n"(OFF), "n"(WAIT)); \
} \
template <int OFF, int WAIT, int INIT> \
__device__ __forceinline__ void c3_slot2##SFX(c3_f32x16& a0, c3_f32x16& a1, const c3_u32x4& w0, const c3_u32x4& w1, c3_u32x4& x, uint32_t xaddr, \
                                         const c3_f32x16& bias) { \
  if constexpr (INIT == 0) \
    asm volatile("s_waitcnt lgkmcnt(%8)\n\t" MN "%0, %3, %2, %5\n\t" MN "%1, %4, %2, %1\n\tds_read_b128 %2, %6 offset:%7" \
                 : "+v"(a0), "+v"(a1), XC(x) : "a"(w0), "a"(w1), "v"(bias), "v"(xaddr), "n"(OFF), "n"(WAIT)); \
  else if constexpr (INIT == 1) \
    asm volatile("s_waitcnt lgkmcnt(%8)\n\t" MN "%0, %3, %2, %0\n\t" MN "%1, %4, %2, %5\n\tds_read_b128 %2, %6 offset:%7" \
                 : "+v"(a0), "+v"(a1), XC(x) : "a"(w0), "a"(w1), "v"(bias), "v"(xaddr), "n"(OFF), "n"(WAIT)); \
  else \
    asm volatile("s_waitcnt lgkmcnt(%7)\n\t" MN "%0, %3, %2, %0\n\t" MN "%1, %4, %2, %1\n\tds_read_b128 %2, %5 offset:%6" \
                 : "+v"(a0), "+v"(a1), XC(x) : "a"(w0), "a"(w1), "v"(xaddr), "n"(OFF), "n"(WAIT)); \
} \
template <int OFF, int WAIT> \
__device__ __forceinline__ void c3_slot3##SFX(c3_f32x16& a0, c3_f32x16& a1, c3_f32x16& a2, const c3_u32x4& w0, const c3_u32x4& w1, const c3_u32x4& w2, \
                                         c3_u32x4& x, uint32_t xaddr) { \
  asm volatile("s_waitcnt lgkmcnt(%9)\n\t" MN "%0, %4, %3, %0\n\t" MN "%1, %5, %3, %1\n\t" MN "%2, %6, %3, %2\n\tds_read_b128 %3, %7 offset:%8" \
               : "+v"(a0), "+v"(a1), "+v"(a2), XC(x) : "a"(w0), "a"(w1), "a"(w2), "v"(xaddr), "n"(OFF), "n"(WAIT)); \
}
C3_DEFINE_SLOTS(_bf16, "v_mfma_f32_32x32x16_bf16 ", "+v")
C3_DEFINE_SLOTS(_f16, "v_mfma_f32_32x32x16_f16 ", "+v")
C3_DEFINE_SLOTS(_bf16a, "v_mfma_f32_32x32x16_bf16 ", "+a")
C3_DEFINE_SLOTS(_f16a, "v_mfma_f32_32x32x16_f16 ", "+a")
#undef C3_DEFINE_SLOTS
template <bool F16, bool XA, int OFF, int WAIT, int INIT>
__device__ __forceinline__ void c3_slot1(c3_f32x16& a0, const c3_u32x4& w0, c3_u32x4& x, uint32_t xaddr, const c3_f32x16& bias) {
  if constexpr (F16 && XA) c3_slot1_f16a<OFF, WAIT, INIT>(a0, w0, x, xaddr, bias);
  else if constexpr (F16) c3_slot1_f16<OFF, WAIT, INIT>(a0, w0, x, xaddr, bias);
  else if constexpr (XA) c3_slot1_bf16a<OFF, WAIT, INIT>(a0, w0, x, xaddr, bias);
  else c3_slot1_bf16<OFF, WAIT, INIT>(a0, w0, x, xaddr, bias);
}
template <bool F16, bool XA, int OFF, int WAIT, int INIT>
__device__ __forceinline__ void c3_slot2(c3_f32x16& a0, c3_f32x16& a1, const c3_u32x4& w0, const c3_u32x4& w1, c3_u32x4& x, uint32_t xaddr, const c3_f32x16& bias) {
  if constexpr (F16 && XA) c3_slot2_f16a<OFF, WAIT, INIT>(a0, a1, w0, w1, x, xaddr, bias);
  else if constexpr (F16) c3_slot2_f16<OFF, WAIT, INIT>(a0, a1, w0, w1, x, xaddr, bias);
  else if constexpr (XA) c3_slot2_bf16a<OFF, WAIT, INIT>(a0, a1, w0, w1, x, xaddr, bias);
  else c3_slot2_bf16<OFF, WAIT, INIT>(a0, a1, w0, w1, x, xaddr, bias);
}
template <bool F16, bool XA, int OFF, int WAIT>
__device__ __forceinline__ void c3_slot3(c3_f32x16& a0, c3_f32x16& a1, c3_f32x16& a2, const c3_u32x4& w0, const c3_u32x4& w1, const c3_u32x4& w2, c3_u32x4& x, uint32_t xaddr) {
  if constexpr (F16 && XA) c3_slot3_f16a<OFF, WAIT>(a0, a1, a2, w0, w1, w2, x, xaddr);
  else if constexpr (F16) c3_slot3_f16<OFF, WAIT>(a0, a1, a2, w0, w1, w2, x, xaddr);
  else if constexpr (XA) c3_slot3_bf16a<OFF, WAIT>(a0, a1, a2, w0, w1, w2, x, xaddr);
  else c3_slot3_bf16<OFF, WAIT>(a0, a1, a2, w0, w1, w2, x, xaddr);
}

// slot n of a tile -> fragment (input row r, kx, k-slice q): the input rows are paired (0,5), (1,4), (2,3) and interleaved, so that
// consecutive MFMAs never form a chain on ONE accumulator (rows 0 and 5 feed a single output row each)
__host__ __device__ constexpr int wr_slot_r(int n) { return ((n % 24) & 1) == 0 ? n / 24 : 5 - n / 24; }
__host__ __device__ constexpr int wr_slot_kx(int n) { return ((n % 24) / 2) / 4; }
__host__ __device__ constexpr int wr_slot_q(int n) { return ((n % 24) / 2) % 4; }
__host__ __device__ constexpr int wr_slot_off(int n) { return (wr_slot_r(n) * WR_PW + wr_slot_kx(n)) * WR_PITCH + wr_slot_q(n) * 32; }
// is (slot n, ky) the first MFMA of the tile on accumulator j = r - ky? (it takes the bias as its C operand)
__host__ __device__ constexpr bool wr_first_touch(int n, int ky) {
  const int j = wr_slot_r(n) - ky;
  for (int m = 0; m <= n; ++m)
    for (int k2 = 0; k2 < 3; ++k2) {
      const int j2 = wr_slot_r(m) - k2;
      if (j2 != j) continue;
      return m == n && k2 == ky;      // MFMAs of one slot run in ascending ky
    }
  return false;
}

// ---------------------------------------------------------------------------------------------
// FUSE: conv1_1 (reference VGGnet_test.py:20-22, the layer in front of conv1_2) computed INSIDE conv1_2's window stage. The 12 LDS-DMA
// pieces that fetched window k + 2 from conv1_1's stored output (69 MB per 600 x 900 image, written once and read back once) are replaced by
//   * ONE 1-KiB LDS-DMA per wave and tile: the 12 x 36-pixel patch of the q-image (common.h: 8-byte pixels (q_B, q_G, q_R, P), 4.7 MB per
//     image) under tile k + 3, into one of two 4-KiB planes;
//   * a producer for window k + 2 threaded through the tile's slots: the window's 340 pixels are 4 waves x 85, each wave's 85 as three
//     32-pixel MFMA groups (the third overlaps the second by 11 pixels: identical values written twice); per group three ds_read2_b64
//     (tap rows ky = 0..2; lanes 0..31 read q-pixels (x - 1, x), lanes 32..63 (x, x + 1): K-slot order in layers.hip, pack_conv1_frags),
//     2 x 3 MFMAs (two 32-channel halves, ky = 0, 1, 2 from a zero accumulator) and 2 x 4 epilogue pieces (two packed converts, the
//     ReLU as a packed integer max, one ds_write_b64 into the window buffer at the 144-byte pixel pitch -- two-way bank-conflicted; the
//     conflict-free ds_write_b128 form behind v_permlane32_swap measured 0.5 % slower);
//   * window pixels outside the image (conv1_2's SAME padding; the overhang of ragged tiles) read their operands from a zero region
//     instead: zero operands, zero sums (the bias rides on the centre pixel's P), zero after the ReLU -- no masking of results.
// Every producer LDS operation and MFMA is its own asm statement in a fixed slot (fq_* below), at most ONE LDS operation per slot, so the
// ring's counted waits stay exact: slot n waits with lgkmcnt(7 + the producer's operations of the eight slots before it). A fragment read
// is consumed >= 9 slots after its issue (the ring wait of that slot covers it), an accumulator is read by the epilogue's VALU >= 2
// slots after its last MFMA. 18 MFMAs per wave on top of the tile's 144.
// conv_first_p_kernel (layers.hip) runs the same MFMA sequence on the same operands from global memory: what keep_acts stores.
// ---------------------------------------------------------------------------------------------
constexpr int FQ_PW = 36, FQ_ROWB = FQ_PW * 8, FQ_PLANE = 4096;
constexpr int FQ_PLANE_OFF = WR_NBUF * WR_WIN + 16, FQ_ZERO_OFF = FQ_PLANE_OFF + 2 * FQ_PLANE, FQ_LDS = FQ_ZERO_OFF + 1024;
static_assert(12 * FQ_ROWB <= FQ_PLANE && 3 * FQ_ROWB <= 1024 && FQ_LDS <= 160 * 1024, "q planes / zero region");
__host__ __device__ constexpr int fq_goff(int gi) { return gi == 0 ? 0 : (gi == 1 ? 32 : 53); }      // first window pixel of a wave's group gi, relative to 85 * wave
// pieces by slot (-1: none). setup: group; read: gi * 3 + ky; mma: (gi * 2 + i) * 3 + ky; epi: (gi * 2 + i) * 4 + h
__host__ __device__ constexpr int fq_setup(int n) { return n == 9 ? 0 : (n == 13 ? 1 : (n == 30 ? 2 : -1)); }
__host__ __device__ constexpr int fq_read(int n) { return (n >= 10 && n <= 12) ? n - 10 : ((n >= 14 && n <= 16) ? 3 + n - 14 : ((n >= 36 && n <= 38) ? 6 + n - 36 : -1)); }
__host__ __device__ constexpr int fq_mma(int n) {
  return (n >= 24 && n <= 29) ? n - 24 : ((n >= 32 && n <= 34) ? 6 + n - 32 : ((n >= 36 && n <= 38) ? 9 + n - 36 : ((n >= 48 && n <= 53) ? 12 + n - 48 : -1)));
}
__host__ __device__ constexpr int fq_epi(int n) { return (n >= 28 && n <= 35) ? n - 28 : ((n >= 39 && n <= 46) ? 8 + n - 39 : ((n >= 54 && n <= 61) ? 16 + n - 54 : -1)); }
constexpr int FQ_INCOMING = 17, FQ_DMA = 18;
__host__ __device__ constexpr int fq_lds_ops(int n) { return (fq_read(n) >= 0 ? 1 : 0) + (fq_epi(n) >= 0 ? 1 : 0); }
// lgkmcnt of slot n: the ring read it waits for was issued in slot n - 8; behind it: 7 ring reads, the producer's operations of slots
// n - 8 .. n - 1 (a slot's pieces follow its ring read) and the queue word read in front of slot 8's ring read (every wave issues it)
// (left at 7 the waits are merely stricter: measured 0.2 % slower, profiles/r04_ab_conv1_fuse.txt)
__host__ __device__ constexpr int fq_wait(int n) {
  int w = WR_PD - 1 + ((n >= 8 && n <= 15) ? 1 : 0);
  for (int m = n - 8; m < n; ++m) w += fq_lds_ops((m + 72) % 72);
  return w > 15 ? 15 : w;
}

// the hand schedule's invariants, checked at compile time (edit the tables above and this tells what broke)
__host__ __device__ constexpr int fq_slot_of(int kind, int id) {      // kind: 0 setup, 1 read, 2 mma, 3 epi
  for (int n = 0; n < 72; ++n)
    if ((kind == 0 ? fq_setup(n) : kind == 1 ? fq_read(n) : kind == 2 ? fq_mma(n) : fq_epi(n)) == id) return n;
  return -1;
}
__host__ __device__ constexpr bool fq_schedule_ok() {
  for (int n = 0; n < 72; ++n) {
    if (fq_lds_ops(n) > 1) return false;                                                   // the lgkmcnt table assumes at most one per slot
    // behind the barrier; the last LDS write early enough that slot 7's wait of the NEXT tile (in front of its barrier) has retired it:
    // a write in slot s has (71 - s) + 7 ring reads behind it there, the wait leaves 7 (+ 1 if slot 71 holds an operation) in flight
    if ((fq_setup(n) >= 0 || fq_lds_ops(n) || fq_mma(n) >= 0) && (n <= WR_PD || n > 69)) return false;
  }
  if (!(FQ_INCOMING > WR_PD + 8 && FQ_DMA > FQ_INCOMING && FQ_DMA < WR_PD + 1 + 12)) return false;      // queue word complete; base a slot ahead; loads in front of the stores
  for (int gi = 0; gi < 3; ++gi) {
    if (fq_slot_of(0, gi) < 0 || fq_slot_of(0, gi) >= fq_slot_of(1, gi * 3)) return false;               // address before the reads
    for (int ky = 0; ky < 3; ++ky) {
      const int r = fq_slot_of(1, gi * 3 + ky);
      if (r < 0) return false;
      for (int i = 0; i < 2; ++i) {
        const int m = fq_slot_of(2, (gi * 2 + i) * 3 + ky);
        if (m < r + 9) return false;                                                        // the ring wait of slot r + 9 covers the read
        if (ky > 0 && m <= fq_slot_of(2, (gi * 2 + i) * 3 + ky - 1)) return false;            // the chain in order, one MFMA per slot
      }
      // operand set (gi & 1) and its address register are rewritten for group gi + 2 only after group gi's last use (in-slot order: mma, then read)
      if (gi + 2 < 3 && (fq_slot_of(1, (gi + 2) * 3 + ky) < fq_slot_of(2, (gi * 2 + 1) * 3 + ky) || fq_slot_of(0, gi + 2) <= fq_slot_of(1, gi * 3 + 2))) return false;
    }
    for (int i = 0; i < 2; ++i)
      for (int h = 0; h < 4; ++h) {
        const int e = fq_slot_of(3, (gi * 2 + i) * 4 + h);
        if (e < fq_slot_of(2, (gi * 2 + i) * 3 + 2) + 2) return false;                      // MFMA result -> VALU read
        if (gi + 1 < 3 && fq_slot_of(2, ((gi + 1) * 2 + i) * 3) < e) return false;           // accumulator i is overwritten after its epilogue (in-slot order: epi, then mma)
      }
  }
  return true;
}
static_assert(fq_schedule_ok(), "producer schedule violates one of its invariants");

typedef uint32_t c3_u32x2 __attribute__((ext_vector_type(2)));
// the producer's instructions, one asm statement each (operands in the accumulation file: "a")
template <int O0, int O1>
__device__ __forceinline__ void c3_fq_read2(c3_u32x4& dst, uint32_t addr) {
  asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "+a"(dst) : "v"(addr), "n"(O0), "n"(O1));
}
template <bool F16, bool FIRST>
__device__ __forceinline__ void c3_fq_mfma(c3_f32x16& acc, const c3_u32x4& w, const c3_u32x4& x) {
  if constexpr (FIRST) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "+v"(acc) : "a"(w), "a"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+v"(acc) : "a"(w), "a"(x));
  } else {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "a"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "a"(x));
  }
}
template <int OFF>
__device__ __forceinline__ void c3_fq_write(uint32_t addr, const c3_u32x2& d) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(d), "n"(OFF) : "memory");
}
// ABL (measurement only, wrong results): 1 = no window DMA after the prologue, 2 = no epilogue
template <typename HF, bool POOL, bool FULL, int ABL = 0, bool FUSE = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wr_kernel(Conv3WR g) {
  constexpr bool F16 = std::is_same<HF, h_f16>::value;
  static_assert(POOL || FULL, "nothing to store");
  static_assert(!FUSE || (POOL && !FULL && ABL == 0), "the fused form is conv1_2's production launch");
  constexpr int PD = WR_PD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave & 1, chh = wave >> 1;
  const int l31 = lane & 31, fhalf = lane >> 5;
  const int H = g.H, W = g.W, Co = g.Co, Wp = W + 2, Hp = H + 2;
  const int tiles_n = g.tiles_n, tiles_x = g.tiles_x;
  const unsigned per_img = (unsigned)(g.tiles_x * g.tiles_y);
  const unsigned magic_img = g.magic_img, magic_row = g.magic_row;
  // XCD-local tile ranges: every XCD (own L2) walks ONE contiguous band of tiles -- and, for Co = 128, walks it with BOTH channel slices.
  // With tiles dealt w, w + nworkers, ... over the whole grid, x- and y-neighbours (which share a third of their 10 x 34 window) and the
  // two slices of a tile (which read the SAME window) sat on different XCDs, so every L2 fetched its own copy: conv2_1 read 2.5 x its
  // input from HBM, conv1_2 1.3 x. `ptiles` below is the END of this workgroup's range; dynamic claims come from the group's own counter.
  const unsigned grp = blockIdx.x % g.groups, kq = blockIdx.x / g.groups;
  const int tn = (int)(kq % (unsigned)tiles_n);
  const int n0 = tn * 64 + chh * 32;                        // this wave's first output channel
  const unsigned worker = kq / (unsigned)tiles_n, nworkers = gridDim.x / (g.groups * (unsigned)tiles_n);
  const unsigned range_lo = grp * g.per_group;
  const unsigned ptiles = range_lo + g.per_group < g.ptiles ? range_lo + g.per_group : g.ptiles;     // end of the range (may be <= range_lo: empty)
  const char* const in_base = (const char*)g.in;

  // ---- weights and bias of this wave's 32 channels: registers, once ----
  c3_u32x4 wf[9][4];
  {
    const char* wp = (const char*)g.wt + (size_t)(n0 + l31) * (9 * 64 * 2) + fhalf * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) wf[t][q] = *(const c3_u32x4*)(wp + t * 128 + q * 32);
  }
  // the bias enters as the accumulators' initial value (C operand of each tile's first MFMA per pixel row): no add in the epilogue
  c3_f32x16 bias16;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const c3_f32x4 b4 = *(const c3_f32x4*)(g.bias + n0 + 8 * g4 + 4 * fhalf);
#pragma unroll
    for (int e = 0; e < 4; ++e) bias16[4 * g4 + e] = b4[e];
  }

  // ---- window pieces of this wave: piece P = wave + 4 i covers LDS bytes [1024 P, 1024 P + 1024) of a window buffer ----
  uint32_t voff[FUSE ? 1 : 12];
  if constexpr (!FUSE) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int o = (wave + 4 * i) * 1024 + lane * 16;
      int row = o / WR_PITCH;
      int slot = (o - row * WR_PITCH) >> 4;
      if (row >= WR_ROWS || slot == 8) { row = 0; slot = 0; }   // pad slots / tail of the last piece: any valid 16 bytes
      const int i2 = row / WR_PW, j2 = row - i2 * WR_PW;
      voff[i] = (uint32_t)((i2 * Wp + j2) * 128 + slot * 16);
    }
  }
  // ---- FUSE: the producer's per-lane constants ----
  // fq_rd[gi]: LDS address (plane 0) of this lane's 16 operand bytes of tap row 0, group gi; fq_rc[gi]: the group pixel's window row | column << 8;
  // fq_wr: this lane's write address in window buffer 0 for group offset 0; voff[0]: source offset of its 16 bytes of the q patch
  uint32_t fq_rd[3] = {0u, 0u, 0u}, fq_rc[3] = {0u, 0u, 0u}, fq_wr = 0u;
  c3_u32x4 wq[6];
  if constexpr (FUSE) {
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
      const int p = 85 * wave + fq_goff(gi) + l31;                 // < 340
      const int r = p / WR_PW, c = p - r * WR_PW;
      fq_rd[gi] = lds0 + (uint32_t)(FQ_PLANE_OFF + (r * FQ_PW + c + fhalf) * 8);
      fq_rc[gi] = (uint32_t)(r | (c << 8));
    }
    fq_wr = lds0 + (uint32_t)((85 * wave + l31) * WR_PITCH + 8 * fhalf);
    int j = wave * 64 + lane;                                      // 16-byte chunk of the 12 x 288-byte patch; the plane's tail: any valid bytes
    if (j >= 12 * (FQ_ROWB / 16)) j = 0;
    const int row = j / (FQ_ROWB / 16), cc = j - row * (FQ_ROWB / 16);
    voff[0] = (uint32_t)(row * g.Wq * 8 + cc * 16);
    const char* wp = (const char*)g.wfq + lane * 16;
#pragma unroll
    for (int k = 0; k < 6; ++k) wq[k] = *(const c3_u32x4*)(wp + k * 1024);
  }
  // tile bookkeeping is wave-uniform: kept on the scalar unit (readfirstlane pins the values to SGPRs; the divisions are
  // multiplications by host-computed reciprocals)
  auto sgpr = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  auto mulhi = [](unsigned a, unsigned b) -> unsigned { return (unsigned)(((unsigned long long)a * (unsigned long long)b) >> 32); };
  auto tile_coords = [&](unsigned pt, int& img, int& y0, int& x0) {
    pt = sgpr(pt);
    // (a divisor of 1 has no 32-bit reciprocal of this form: floor(2^32 / 1) + 1 wraps -- one tile per image / one tile column)
    const unsigned im = per_img == 1u ? pt : mulhi(pt, magic_img);
    const unsigned rem = pt - im * per_img;
    const unsigned ty = tiles_x == 1 ? rem : mulhi(rem, magic_row);
    img = (int)sgpr(im); y0 = (int)sgpr(ty * 8u); x0 = (int)sgpr((rem - ty * (unsigned)tiles_x) * 32u);
  };
  auto window_base = [&](unsigned pt) -> const char* {
    int img, y0, x0;
    tile_coords(pt, img, y0, x0);
    const unsigned pix = sgpr((unsigned)((img * Hp + y0) * Wp + x0));           // < 2^31 (checked by the launcher)
    const unsigned long long a = (unsigned long long)(uintptr_t)in_base + ((unsigned long long)pix << 7);
    const unsigned lo = sgpr((unsigned)a), hi = sgpr((unsigned)(a >> 32));
    return (const char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  };
  auto issue_piece = [&](auto ic, const char* sbase, uint32_t buf_lds) {
    constexpr int i = decltype(ic)::value;
    c3_glds16_saddr(sbase, voff[i], __builtin_amdgcn_readfirstlane(buf_lds + (wave + 4 * i) * 1024));
  };
  // FUSE: the q patch under tile pt starts at q pixel (y0, x0) of its image (image pixel (y0 - 2, x0 - 2)); one 1-KiB piece per wave
  auto q_base = [&](unsigned pt) -> const char* {
    int img, y0, x0;
    tile_coords(pt, img, y0, x0);
    const unsigned pix = sgpr((unsigned)((img * g.Hq + y0) * g.Wq + x0));       // < 2^31 (checked by the launcher)
    const unsigned long long a = (unsigned long long)(uintptr_t)g.q + ((unsigned long long)pix << 3);
    const unsigned lo = sgpr((unsigned)a), hi = sgpr((unsigned)(a >> 32));
    return (const char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  };
  auto issue_plane = [&](const char* sbase, int plane) {
    c3_glds16_saddr(sbase, voff[0], __builtin_amdgcn_readfirstlane(lds0 + FQ_PLANE_OFF + plane * FQ_PLANE + wave * 1024));
  };
  // ---- FUSE: the producer's pieces (kernel comment). Operand sets xq[gi & 1][ky] and the 24 weight fragment registers live in the
  // accumulation file; the two accumulators (one per 32-channel half) in VGPRs, tied in place like everything else asm writes ----
  c3_u32x4 xq[2][3];
  c3_f32x16 pacc[2];
  uint32_t fq_a[2] = {0u, 0u};                // operand address of the group whose reads are in flight, per operand set
  if constexpr (FUSE) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int k = 0; k < 3; ++k) xq[a][k] = c3_u32x4{0u, 0u, 0u, 0u};
      pacc[a] = c3_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
  }
  // window pixel (r, c) of the tile at (y2, x2) is image pixel (y2 + r - 1, x2 + c - 1); outside the image its operands come from the zero region
  auto fq_setup_piece = [&](auto gic, int y2, int x2, uint32_t plane_off) {
    constexpr int gi = decltype(gic)::value;
    const uint32_t r = fq_rc[gi] & 0xffu, c = fq_rc[gi] >> 8;
    const bool in = (r + (uint32_t)(y2 - 1)) < (uint32_t)H && (c + (uint32_t)(x2 - 1)) < (uint32_t)W;
    fq_a[gi & 1] = in ? fq_rd[gi] + plane_off : lds0 + (uint32_t)FQ_ZERO_OFF;
  };
  auto fq_read_piece = [&](auto gic, auto kyc) {
    constexpr int gi = decltype(gic)::value, ky = decltype(kyc)::value;
    c3_fq_read2<ky * FQ_PW, ky * FQ_PW + 1>(xq[gi & 1][ky], fq_a[gi & 1]);
  };
  auto fq_mma_piece = [&](auto gic, auto ic, auto kyc) {
    constexpr int gi = decltype(gic)::value, i = decltype(ic)::value, ky = decltype(kyc)::value;
    c3_fq_mfma<F16, ky == 0>(pacc[i], wq[i * 3 + ky], xq[gi & 1][ky]);
  };
  // a lane owns channels 32 i + 8 h + 4 fhalf + e (accumulator element 4 h + e) of window pixel 85 wave + goff + l31: 8 bytes per piece
  auto fq_epi_piece = [&](auto gic, auto ic, auto hc, uint32_t wbuf) {
    constexpr int gi = decltype(gic)::value, i = decltype(ic)::value, h = decltype(hc)::value;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    auto rp = [](uint32_t u) -> uint32_t { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), s16x2{0, 0})); };
    c3_u32x2 d;
    d[0] = rp(c3_cvt_pk<HF>(pacc[i][4 * h], pacc[i][4 * h + 1]));
    d[1] = rp(c3_cvt_pk<HF>(pacc[i][4 * h + 2], pacc[i][4 * h + 3]));
    c3_fq_write<fq_goff(gi) * WR_PITCH + 64 * i + 16 * h>(wbuf, d);
  };
  // one whole window, back to back (prologue): plane -> window buffer
  auto fq_produce_sync = [&](unsigned pt, int plane, uint32_t wbuf) {
    int img, y2, x2;
    tile_coords(pt, img, y2, x2);
    c3_static_for<3>([&](auto gic) {
      fq_setup_piece(gic, y2, x2, (uint32_t)(plane * FQ_PLANE));
      c3_static_for<3>([&](auto kyc) { fq_read_piece(gic, kyc); });
      c3_wait_lgkm<0>();
      c3_static_for<2>([&](auto ic) {
        c3_static_for<3>([&](auto kyc) { fq_mma_piece(gic, ic, kyc); });
      });
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA result -> VALU read
      __builtin_amdgcn_sched_barrier(0);
      c3_static_for<2>([&](auto ic) {
        c3_static_for<4>([&](auto hc) { fq_epi_piece(gic, ic, hc, wbuf); });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- tile queue: t[k] (current), t[k+1], t[k+2] (its window is issued during tile k); t[k+3] arrives during tile k ----
  // static: t[i] = worker + i * nworkers for i < 5; dynamic: 5 * nworkers + (old value of the slice's counter). An index >= ptiles
  // means "no tile": the walk ends at the first one.
  unsigned* claim_ctr;
  {
    const unsigned long long a = (unsigned long long)(uintptr_t)(g.claim + 2 * (grp * (unsigned)tiles_n + (unsigned)tn));
    const unsigned lo = sgpr((unsigned)a), hi = sgpr((unsigned)(a >> 32));          // pinned to an SGPR pair (asm "s" operand below)
    claim_ctr = (unsigned*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  }
  const unsigned claim_base = sgpr(range_lo + 5u * nworkers);
  const uint32_t claim_lds = lds0 + WR_NBUF * WR_WIN;           // two words, alternating by tile parity
  // The fetched value and the word read back from LDS arrive ASYNCHRONOUSLY into their destination registers; hipcc, which takes an
  // asm's outputs as ready when the asm ends, must never touch them before the covering wait (a first version returned into a
  // VGPR that hipcc, short of VGPRs, copied to an AGPR in the very next instruction -- i.e. before the atomic had returned:
  // every workgroup then claimed the same tile for ever). Both therefore live in AGPRs (plenty are free, nothing spills them),
  // tied in place ("+a"), and are only read by asm that runs behind the wait.
  auto claim_issue = [&](uint32_t& ret) {                        // wave 0, lane 0: fetch-and-add; the result is read a tile later
    if (wave == 0 && lane == 0) {
      const uint32_t zero = 0u, one = 1u;
      asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "+a"(ret) : "v"(zero), "a"(one), "s"(claim_ctr) : "memory");   // one ACC bit covers vdst and vdata
    }
  };
  auto claim_publish = [&](uint32_t& ret, unsigned word) {       // wave 0, lane 0: the value fetched during the previous tile -> LDS
    if (wave == 0 && lane == 0) {
      const uint32_t a = claim_lds + 4 * word;
      uint32_t tmp;
      asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_add_u32 %0, %0, %3\n\tds_write_b32 %2, %0" : "=&v"(tmp) : "a"(ret), "v"(a), "s"(claim_base) : "memory");
    }
  };
  auto claim_read = [&](uint32_t& dst, unsigned word) {          // every lane of every wave (same address: broadcast)
    const uint32_t a = claim_lds + 4 * word;
    asm volatile("ds_read_b32 %0, %1" : "+a"(dst) : "v"(a));
  };
  auto claim_value = [&](uint32_t& dst) -> unsigned {            // behind the ring waits that cover claim_read (slot PD + 8 and later)
    uint32_t v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(dst));
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  };
  unsigned q0 = range_lo + worker, q1 = q0 + nworkers, q2 = q0 + 2 * nworkers;
  if (q0 >= ptiles) {                                            // nothing to do (never with the launcher's grid); still counts as finished
    if (tid == 0) {
      const unsigned done = __hip_atomic_fetch_add(claim_ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done == nworkers - 1) { __hip_atomic_store(claim_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(claim_ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    return;
  }
  // ---- prologue: windows of the first two tiles ----
  if constexpr (!FUSE) {
    const char* b0 = window_base(q0);
    const char* b1 = window_base(q1 < ptiles ? q1 : q0);
    c3_static_for<12>([&](auto ic) { issue_piece(ic, b0, lds0); });
    c3_static_for<12>([&](auto ic) { issue_piece(ic, b1, lds0 + WR_WIN); });
  } else {
    // zero region; patches of the first two tiles; their windows, produced back to back; then the third tile's patch into plane 0
    {
      const uint32_t za = lds0 + (uint32_t)FQ_ZERO_OFF + 4u * (uint32_t)tid, zero = 0u;
      asm volatile("ds_write_b32 %0, %1" : : "v"(za), "v"(zero) : "memory");
    }
    // (the scalar bases are fresh from v_readfirstlane: VALU-written SGPR -> VMEM address needs five wait states, and hipcc pads nothing
    // for an asm statement -- all three bases first, then a nop, then the loads)
    const char* const pb0 = q_base(q0);
    const char* const pb1 = q_base(q1 < ptiles ? q1 : q0);
    const char* const pb2 = q_base(q2 < ptiles ? q2 : q0);
    asm volatile("s_nop 4" ::: "memory");
    issue_plane(pb0, 0);
    issue_plane(pb1, 1);
    c3_wait_vm<0>();
    c3_wait_lgkm<0>();
    c3_barrier();
    fq_produce_sync(q0, 0, fq_wr);
    fq_produce_sync(q1 < ptiles ? q1 : q0, 1, fq_wr + (uint32_t)WR_WIN);
    c3_wait_lgkm<0>();
    c3_barrier();                               // every wave is done with both planes
    issue_plane(pb2, 0);
  }
  const uint32_t xbase = lds0 + (uint32_t)((4 * ph * WR_PW + l31) * WR_PITCH + fhalf * 16);
  c3_wait_vm<0>();
  if constexpr (FUSE) c3_wait_lgkm<0>();
  c3_barrier();

  c3_u32x4 xr[PD];
  c3_f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < PD; ++i) xr[i] = c3_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[a][j] = bias16;
  // fragment read of slot n = (input row r, kx, k-slice q), see wr_slot_*
  auto read_frag = [&](auto nc, uint32_t xaddr) {
    constexpr int n = decltype(nc)::value;
    c3_ds_read_b128_off<wr_slot_off(n), FUSE>(xr[n % PD], xaddr);
  };
  c3_static_for<PD>([&](auto nc) { read_frag(nc, xbase); });   // tile 0, buffer 0

  int p_img = 0, p_y0 = 0, p_x0 = 0;     // previous tile (its epilogue runs inside the current one)
  bool p_valid = false;

  // ---- epilogue of the tile at (img, y0, x0) on accumulator set `es`, in small pieces (a few VALU each, so that they hide in
  // the gaps between the next tile's MFMAs) ----
  // The bias entered through the accumulators' initial value (bias16 below), ReLU is a packed integer max on the bf16 pairs
  // (sign bit set <=> negative). A lane owns channels 8 g4 + 4 fhalf + e of pixel column l31 of each of its 4 pixel rows;
  // v_permlane32_swap pairs the two half-waves so that every lane stores 8 consecutive channels (16 bytes).
  typedef short c3_s16x2 __attribute__((ext_vector_type(2)));
  auto relu_pk = [](uint32_t p) -> uint32_t {
    const c3_s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(c3_s16x2, p), z));
  };
  uint32_t pk[8];                         // packed bf16 pairs of the piece group in flight: pk[2 g4 + h] = channels 8 g4 + 4 fhalf + 2 h, + 1
  float pm[2];                            // pool: the two values of the pair being built
  // Store addressing: scalar 64-bit row base (SALU) + per-lane 32-bit byte offset computed once per kernel -> the store is
  // `global_store_dwordx4 voff, data, s[base] offset:imm`. (Per-lane 64-bit pixel arithmetic cost two v_mad_u64_u32 + two
  // v_mul_lo_u32 -- quarter-rate -- per 16-byte store: the full-resolution epilogue took 40 % of conv2_1's time.)
  typedef __attribute__((address_space(1))) char* c3_gptr;     // global address space: a pointer rebuilt from integers would otherwise be
                                                               // generic, i.e. a flat_store, which also counts in lgkmcnt
  const c3_gptr dump_lane = (c3_gptr)(uintptr_t)(g.dump + (size_t)blockIdx.x * 4096 + tid * 16);
  // Every store is ALWAYS issued (lanes outside the image write to the dump page): the number of stores per tile is a compile-time
  // constant, which is what lets the tile barrier wait with a counted vmcnt for "everything but my newest stores"
  auto sbase64 = [&](const void* base, unsigned long long byte_off) -> c3_gptr {     // uniform pointer pinned to an SGPR pair
    const unsigned long long a = (unsigned long long)(uintptr_t)base + byte_off;
    const unsigned lo = sgpr((unsigned)a), hi = sgpr((unsigned)(a >> 32));
    return (c3_gptr)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  };
  const int co_shift = Co == 64 ? 7 : 8;                                                                 // bytes per pixel = Co * 2 (Co is 64 or 128)
  // pool: piece i (0..15) = accumulator element idx i: vertical max over the wave's own rows (2 jp, 2 jp + 1), horizontal max
  // with lane ^ 1 (DPP quad_perm [1,0,3,2]); lanes 2k / 2k+1 then hold the same two pooled pixels: the even lane keeps pooled
  // row 0 of the wave, the odd lane pooled row 1. max commutes with the bias, the ReLU and the bf16 rounding.
  auto pool_elem = [&](auto esc, auto ic) {
    constexpr int es = decltype(esc)::value, idx = decltype(ic)::value;
    const bool odd = (lane & 1) != 0;
    const float v0 = __builtin_fmaxf(acc[es][0][idx], acc[es][1][idx]), v1 = __builtin_fmaxf(acc[es][2][idx], acc[es][3][idx]);
    // the lane keeps `own` and sends the other pooled row to its partner: one cross-lane move per element, pinned by an empty asm
    // (cross-lane results that only feed a later store are otherwise fair game for hipcc's sinking, see conv3x3_p_kernel's hpool)
    const float own = odd ? v1 : v0, send = odd ? v0 : v1;
    float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
    asm volatile("" : "+v"(recv));
    pm[idx & 1] = __builtin_fmaxf(own, recv);
    if constexpr (idx & 1) pk[idx >> 1] = relu_pk(c3_cvt_pk<HF>(pm[0], pm[1]));
  };
  // both 16-byte pieces of a pooled pixel are stored by the second call, after the 16-lane exchange described at full_piece: store A
  // carries pooled columns 0..7 (both pooled rows of the wave), store B columns 8..15, four consecutive lanes per pixel
  const uint32_t pool_pair_off = (uint32_t)(((lane & 1) * ((W >> 1) + 2) + ((l31 & 15) >> 1)) * Co * 2 + (2 * ((lane >> 4) & 1) + fhalf) * 16);
  auto pool_store = [&](auto qc, int img, int y0, int x0, bool valid) {
    constexpr int q2 = decltype(qc)::value;
    if constexpr (q2 == 1) {
      const bool odd = (lane & 1) != 0;
      const int Ho = H >> 1, Wo = W >> 1;
      const int Ys = (y0 >> 1) + 2 * ph, Xs = x0 >> 1;                     // wave-uniform: first pooled row / column of this wave
      const unsigned pix = sgpr((unsigned)((img * (Ho + 2) + Ys + 1) * (Wo + 2) + Xs + 1));
      const c3_gptr rb = sbase64(g.pool_out, ((unsigned long long)pix << co_shift) + (unsigned)(n0 * 2));
      c3_u32x4 v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[4 * q + 0], pk[4 * q + 2], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[4 * q + 1], pk[4 * q + 3], false, false);
        v[q] = c3_u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
      c3_u32x4 va, vb;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const auto r = __builtin_amdgcn_permlane16_swap(v[0][c], v[1][c], false, false);
        va[c] = r[0]; vb[c] = r[1];
      }
      const bool rowok = valid && Ys + (odd ? 1 : 0) < Ho;
      const int xa = (l31 & 15) >> 1;
      const c3_gptr da = rowok && Xs + xa < Wo ? rb + (size_t)pool_pair_off : dump_lane;
      const c3_gptr db = rowok && Xs + 8 + xa < Wo ? rb + (size_t)pool_pair_off + (size_t)(8 * Co * 2) : dump_lane;
      *(__attribute__((address_space(1))) c3_u32x4*)(da) = va;                 // always issued
      *(__attribute__((address_space(1))) c3_u32x4*)(db) = vb;
    }
  };
  // full resolution: piece (j, q2): 8 values of pixel row j -> 4 packed pairs + one 16-byte store
  // The two pieces of a pixel row are stored TOGETHER by the second one: v_permlane16_swap exchanges piece 1 of lanes r with piece 0
  // of lanes r + 16, so that one store carries pixels 0..15 of the row and the other pixels 16..31 with the wave's 64 bytes of a pixel
  // on four consecutive lanes -- 16 distinct lines per store instruction instead of 32 (what a store costs the texture path).
  const uint32_t full_pair_off = (uint32_t)((l31 & 15) * Co * 2 + (2 * ((lane >> 4) & 1) + fhalf) * 16);
  auto full_piece = [&](auto esc, auto jc, auto qc, int img, int y0, int x0, bool valid) {
    constexpr int es = decltype(esc)::value, j = decltype(jc)::value, q2 = decltype(qc)::value;
#pragma unroll
    for (int h = 0; h < 4; ++h) pk[4 * q2 + h] = relu_pk(c3_cvt_pk<HF>(acc[es][j][8 * q2 + 2 * h], acc[es][j][8 * q2 + 2 * h + 1]));
    if constexpr (q2 == 1) {
      const int y = y0 + 4 * ph + j;                                         // wave-uniform
      const unsigned pix = sgpr((unsigned)((img * Hp + y + 1) * Wp + x0 + 1));
      const c3_gptr rb = sbase64(g.out, ((unsigned long long)pix << co_shift) + (unsigned)(n0 * 2));
      c3_u32x4 v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[4 * q + 0], pk[4 * q + 2], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[4 * q + 1], pk[4 * q + 3], false, false);
        v[q] = c3_u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
      c3_u32x4 va, vb;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const auto r = __builtin_amdgcn_permlane16_swap(v[0][c], v[1][c], false, false);
        va[c] = r[0]; vb[c] = r[1];
      }
      const bool rowok = valid && y < H;
      // two stores per pixel row, always issued: the same count per tile as with one store per piece
      const c3_gptr da = rowok && x0 + (l31 & 15) < W ? rb + (size_t)full_pair_off : dump_lane;
      const c3_gptr db = rowok && x0 + 16 + (l31 & 15) < W ? rb + (size_t)full_pair_off + (size_t)(16 * Co * 2) : dump_lane;
      *(__attribute__((address_space(1))) c3_u32x4*)(da) = va;
      *(__attribute__((address_space(1))) c3_u32x4*)(db) = vb;
    }
  };
  // piece list of a tile: pool: 16 element pieces, a store after the 8th and the 16th; then full: 8 pieces
  // Slots of a tile: 0 .. PD: nothing but the K loop; PD: barrier; PD + 1 .. PD + 12: the window pieces of tile k + 2 (right behind
  // the barrier: they get a whole tile to land); from E_FIRST on: the epilogue pieces of tile k - 1. Program order per tile is
  // therefore [12 loads][NS stores], and the next barrier waits with vmcnt(NS): all loads have landed, the stores (whose
  // acknowledgements take microseconds under load) stay in flight for another tile.
  constexpr int NE = (POOL ? 18 : 0) + (FULL ? 8 : 0);
  constexpr int NS = (POOL ? 2 : 0) + (FULL ? 8 : 0);         // 16-byte stores per wave and tile
  constexpr int D_FIRST = PD + 1, E_FIRST = D_FIRST + 12, E_STRIDE = (64 - E_FIRST) / NE;
  static_assert(E_STRIDE >= 1 && E_FIRST + (NE - 1) * E_STRIDE <= 63, "epilogue pieces must fit the tile's slots");
  auto epi_piece = [&](auto esc, auto ec, int img, int y0, int x0, bool valid) {
    constexpr int e = decltype(ec)::value;
    if constexpr (POOL && e < 18) {
      if constexpr (e == 8) pool_store(std::integral_constant<int, 0>{}, img, y0, x0, valid);
      else if constexpr (e == 17) pool_store(std::integral_constant<int, 1>{}, img, y0, x0, valid);
      else pool_elem(esc, std::integral_constant<int, (e < 8 ? e : e - 1)>{});
    } else {
      constexpr int f = e - (POOL ? 18 : 0);
      full_piece(esc, std::integral_constant<int, f / 2>{}, std::integral_constant<int, f % 2>{}, img, y0, x0, valid);
    }
  };

  // ---- one tile on accumulator set `as`; k = its index in this workgroup's walk (q0 = its tile) ----
  uint32_t claim_ret = 0u, claim_val = 0u;      // wave 0 lane 0: counter value fetched during the previous tile; all: the LDS word read this tile
  unsigned bcur = 0, bnext = 1, bdma = 2;       // window buffers of t[k], t[k+1], t[k+2]
  auto tile = [&](auto asc, unsigned k) {
    constexpr int as = decltype(asc)::value;
    k = sgpr(k);
    int c_img, c_y0, c_x0;
    tile_coords(q0, c_img, c_y0, c_x0);
    const char* nbase = nullptr;
    int n_img = 0, n_y0 = 0, n_x0 = 0;                          // FUSE: the tile whose window (k + 2) is produced during this one
    if constexpr (FUSE) tile_coords(q2 < ptiles ? q2 : q0, n_img, n_y0, n_x0);   // past the end: the own window once more, never read
    else nbase = window_base(q2 < ptiles ? q2 : q0);            // past the end: a harmless re-fetch of the own window
    const uint32_t nbuf_lds = sgpr(lds0 + bdma * WR_WIN);
    const uint32_t fq_wbuf = fq_wr + sgpr(bdma * WR_WIN);
    const uint32_t xcur = xbase + sgpr(bcur * WR_WIN);
    const uint32_t xnext = xbase + sgpr((q1 < ptiles ? bnext : bcur) * WR_WIN);   // last tile: dummy reads of its own window
    unsigned incoming = 0u;                                     // t[k + 3]
    const char* dma_base = nullptr;                             // FUSE: its q patch
    c3_static_for<72>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int r = wr_slot_r(n), kx = wr_slot_kx(n), q = wr_slot_q(n);
      if constexpr (n == PD) {
        // every wave has drained its reads of window k - 1 (the ring waits) and, with vmcnt(NS), its pieces of window k + 1
        // (issued during tile k - 1, in front of that tile's NS stores) and wave 0's counter fetch of tile k - 1: after the
        // barrier buffer (k + 2) % 3 may be overwritten and window k + 1 may be read
        // (FUSE: window k + 1 was WRITTEN by the producer during tile k - 1; its last ds_write, slot 61, has 18 ring reads behind it and
        // slot 7's wait leaves at most 15 LDS operations in flight: retired. vmcnt covers the q patch of tile k + 2.)
        c3_wait_vm<(ABL & 2) ? 0 : NS>();
        c3_barrier();
        // tile queue: read the word wave 0 published during tile k - 1 (it sits behind TWO barriers: no wait on the write
        // itself is needed); publish the fetch of tile k - 1 into the other word; fetch the next one. The extra LDS
        // operations only make the ring's counted waits stricter; the word is complete once slot PD + 8 has waited.
        claim_read(claim_val, (k + 1) & 1);
        claim_publish(claim_ret, k & 1);
        claim_issue(claim_ret);
      }
      // the slot: wait for fragment n, its MFMAs (output rows j = r - ky, ascending ky), read of the fragment PD slots ahead
      constexpr int nn = (n + PD) % 72;
      constexpr int off = wr_slot_off(nn);
      constexpr int WT = FUSE ? fq_wait(n) : PD - 1;
      const uint32_t xa = (n + PD < 72) ? xcur : xnext;
      constexpr int j_lo = r - 2 < 0 ? 0 : r - 2, j_hi = r > 3 ? 3 : r;       // output rows fed: j_lo .. j_hi (ky = r - j)
      constexpr int nm = j_hi - j_lo + 1;
      // ascending ky = descending j
      if constexpr (nm == 1) {
        constexpr int ky = r - j_hi;
        c3_slot1<F16, FUSE, off, WT, wr_first_touch(n, ky) ? 0 : -1>(acc[as][j_hi], wf[ky * 3 + kx][q], xr[n % PD], xa, bias16);
      } else if constexpr (nm == 2) {
        constexpr int ky0 = r - j_hi, ky1 = ky0 + 1;
        constexpr int init = wr_first_touch(n, ky0) ? 0 : (wr_first_touch(n, ky1) ? 1 : -1);
        c3_slot2<F16, FUSE, off, WT, init>(acc[as][j_hi], acc[as][j_hi - 1], wf[ky0 * 3 + kx][q], wf[ky1 * 3 + kx][q], xr[n % PD], xa, bias16);
      } else {
        static_assert(!wr_first_touch(n, 0) && !wr_first_touch(n, 1) && !wr_first_touch(n, 2), "three-row slots never start a chain");
        constexpr int ky0 = r - j_hi;
        c3_slot3<F16, FUSE, off, WT>(acc[as][j_hi], acc[as][j_hi - 1], acc[as][j_hi - 2], wf[ky0 * 3 + kx][q], wf[(ky0 + 1) * 3 + kx][q], wf[(ky0 + 2) * 3 + kx][q],
                                     xr[n % PD], xa);
      }
      if constexpr (!FUSE) {
        if constexpr (n >= D_FIRST && n < D_FIRST + 12 && !(ABL & 1)) issue_piece(std::integral_constant<int, n - D_FIRST>{}, nbase, nbuf_lds);
      } else {
        // the producer of window k + 2 (operand plane k & 1 = `as`) and the q patch of tile k + 3 (into the other plane)
        if constexpr (fq_setup(n) >= 0) fq_setup_piece(std::integral_constant<int, fq_setup(n)>{}, n_y0, n_x0, (uint32_t)(as * FQ_PLANE));
        if constexpr (fq_epi(n) >= 0)
          fq_epi_piece(std::integral_constant<int, fq_epi(n) / 8>{}, std::integral_constant<int, (fq_epi(n) / 4) % 2>{}, std::integral_constant<int, fq_epi(n) % 4>{}, fq_wbuf);
        if constexpr (fq_mma(n) >= 0)
          fq_mma_piece(std::integral_constant<int, fq_mma(n) / 6>{}, std::integral_constant<int, (fq_mma(n) / 3) % 2>{}, std::integral_constant<int, fq_mma(n) % 3>{});
        if constexpr (fq_read(n) >= 0) fq_read_piece(std::integral_constant<int, fq_read(n) / 3>{}, std::integral_constant<int, fq_read(n) % 3>{});
        if constexpr (n == FQ_INCOMING) {      // (a slot ahead of the DMA: its scalar base is fresh from readfirstlane)
          incoming = sgpr(k < 2 ? range_lo + worker + (k + 3) * nworkers : claim_value(claim_val));
          dma_base = q_base(incoming < ptiles ? incoming : q0);
        }
        if constexpr (n == FQ_DMA) issue_plane(dma_base, as ^ 1);
      }
      if constexpr (n >= E_FIRST && (n - E_FIRST) % E_STRIDE == 0 && (n - E_FIRST) / E_STRIDE < NE && !(ABL & 2))
        epi_piece(std::integral_constant<int, as ^ 1>{}, std::integral_constant<int, (n - E_FIRST) / E_STRIDE>{}, p_img, p_y0, p_x0, p_valid);
      __builtin_amdgcn_sched_barrier(0);
    });
    p_img = c_img; p_y0 = c_y0; p_x0 = c_x0; p_valid = true;
    // t[k + 3]: static for the first two tiles, then what wave 0 fetched during tile k - 2 (the word read behind this tile's barrier)
    if constexpr (!FUSE) incoming = k < 2 ? range_lo + worker + (k + 3) * nworkers : claim_value(claim_val);
    q0 = q1; q1 = q2; q2 = sgpr(incoming);
    const unsigned b = bcur; bcur = bnext; bnext = bdma; bdma = b;
  };

  unsigned k = 0;
  bool last_set1 = false;
  for (;;) {
    tile(std::integral_constant<int, 0>{}, k);
    last_set1 = false;
    if (q0 >= ptiles) break;
    tile(std::integral_constant<int, 1>{}, k + 1);
    last_set1 = true;
    if (q0 >= ptiles) break;
    k += 2;
  }
  c3_wait_lgkm<0>();
  if (!last_set1) c3_static_for<NE>([&](auto ec) { epi_piece(std::integral_constant<int, 0>{}, ec, p_img, p_y0, p_x0, true); });
  else c3_static_for<NE>([&](auto ec) { epi_piece(std::integral_constant<int, 1>{}, ec, p_img, p_y0, p_x0, true); });
  c3_wait_vm<0>();
  // the last workgroup of the slice to finish re-arms the counters for the next launch (claims all precede a workgroup's exit)
  if (tid == 0) {
    const unsigned done = __hip_atomic_fetch_add(claim_ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == nworkers - 1) {
      __hip_atomic_store(claim_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(claim_ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// per-device resources of the weights-in-registers kernel (conv3x3.hip): the dump pages and the next of 64 tile-claim counter slots
int c3_wr_resources(int dev, hipStream_t s, char** dump, unsigned** claim);

template <typename H>
static int c3_launch_wr(const Conv3& c, bool pool, hipStream_t s) {
  Conv3WR g{};
  g.in = c.in; g.wt = c.wt; g.bias = c.bias; g.out = c.out; g.pool_out = c.pool_out;
  g.N = c.N; g.H = c.H; g.W = c.W; g.Co = c.Co;
  int he, we;
  {
    he = (pool && !c.out) ? (c.H & ~1) : c.H;
    we = (pool && !c.out) ? (c.W & ~1) : c.W;
    if (c.w_cover > 0 && c.w_cover < we) we = c.w_cover;
  }
  g.tiles_x = (we + 31) / 32;
  g.tiles_y = (he + 7) / 8;
  g.tiles_n = c.Co / 64;
  const long long ptiles = (long long)c.N * g.tiles_x * g.tiles_y;
  const long long per_img = (long long)g.tiles_x * g.tiles_y;
  if (ptiles <= 0 || ptiles * per_img >= (1LL << 32) || (long long)c.N * (c.H + 2) * (c.W + 2) * 128 >= (1LL << 40))
    return fail(CTPN_ERR_ARG, "conv3x3_wr: problem out of range");
  g.ptiles = (unsigned)ptiles;
  g.magic_img = (unsigned)((1ULL << 32) / (unsigned long long)per_img + 1ULL);
  g.magic_row = (unsigned)((1ULL << 32) / (unsigned long long)g.tiles_x + 1ULL);
  int dev = 0, ncu = 0, rc;
  if ((rc = c3_device(dev)) || (rc = c3_cu_count(dev, ncu))) return rc;
  g.groups = 8u;                                                     // one tile range per XCD (kernel comment)
  g.per_group = (unsigned)((ptiles + g.groups - 1) / g.groups);
  long long workers = (ncu / g.tiles_n) / (long long)g.groups;      // per group and channel slice
  if (workers < 1) workers = 1;
  if (workers > (long long)g.per_group) workers = g.per_group;
  workers *= g.groups;                                              // per channel slice
  if (workers * g.tiles_n > 1024) return fail(CTPN_ERR_ARG, "conv3x3_wr: more workgroups than dump pages");
  if (g.tiles_n > 4) return fail(CTPN_ERR_ARG, "conv3x3_wr: more channel slices than claim counters per slot");
  if ((rc = c3_wr_resources(dev, s, &g.dump, &g.claim))) return rc;
  const bool fuse = c.q1 != nullptr;
  if (fuse) {
    if (!pool || c.out || c.Co != 64 || !c.q1_frags) return fail(CTPN_ERR_ARG, "conv3x3_wr: the fused conv1_1 form is conv1_2's pooled production launch");
    g.q = c.q1; g.wfq = c.q1_frags; g.Hq = conv1_q_h(c.H); g.Wq = conv1_q_w(c.W);
    // the patch of the last tile must lie inside the q-image (rows y0 .. y0 + 11, columns x0 .. x0 + 35), pixel indices below 2^31
    if (8 * g.tiles_y + 4 > g.Hq || 32 * g.tiles_x + 4 > g.Wq || (long long)c.N * g.Hq * g.Wq >= (1LL << 31))
      return fail(CTPN_ERR_ARG, "conv3x3_wr: q-image geometry out of range");
  }
  const int lds = fuse ? FQ_LDS : WR_NBUF * WR_WIN + 16;
  const dim3 grid((unsigned)(workers * g.tiles_n)), block(256);
  static bool attr[10][C3_MAX_DEV] = {{false}};
  auto launch = [&](auto kern, bool (&done)[C3_MAX_DEV]) -> int {
    const int r = c3_raise_lds((const void*)kern, done, dev);
    if (r) return r;
    hipLaunchKernelGGL(kern, grid, block, lds, s, g);
    return CTPN_OK;
  };
#ifdef CTPN_ABLATION
  // CTPN_C3_WR_VAR (measurement builds only, WRONG results): 1 = no window DMA after the prologue, 2 = no epilogue, 3 = neither
  static const int var = [] { const char* e = std::getenv("CTPN_C3_WR_VAR"); return e ? std::atoi(e) : 0; }();
#else
  constexpr int var = 0;
#endif
  if (fuse) rc = launch(conv3x3_wr_kernel<H, true, false, 0, true>, attr[9]);
  else if (pool && c.out) rc = launch(conv3x3_wr_kernel<H, true, true>, attr[0]);
  else if (pool) {
    switch (var) {
#ifdef CTPN_ABLATION
      case 1: rc = launch(conv3x3_wr_kernel<H, true, false, 1>, attr[1]); break;
      case 2: rc = launch(conv3x3_wr_kernel<H, true, false, 2>, attr[2]); break;
      case 3: rc = launch(conv3x3_wr_kernel<H, true, false, 3>, attr[3]); break;
#endif
      default: rc = launch(conv3x3_wr_kernel<H, true, false>, attr[4]);
    }
  } else {
    switch (var) {
#ifdef CTPN_ABLATION
      case 1: rc = launch(conv3x3_wr_kernel<H, false, true, 1>, attr[5]); break;
      case 2: rc = launch(conv3x3_wr_kernel<H, false, true, 2>, attr[6]); break;
      case 3: rc = launch(conv3x3_wr_kernel<H, false, true, 3>, attr[7]); break;
#endif
      default: rc = launch(conv3x3_wr_kernel<H, false, true>, attr[8]);
    }
  }
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv3x3_wr launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// pixels the 2D tiling has to cover: a fused 2x2 VALID pool that does not keep the full-resolution map never reads an
// odd last row / column (150 x 225 -> 75 x 112 uses 150 x 224), which for W = 225 = 7 * 32 + 1 removes a whole tile column
static inline void c3_extent(const Conv3& g, bool pool, int& he, int& we) {
  he = (pool && !g.out) ? (g.H & ~1) : g.H;
  we = (pool && !g.out) ? (g.W & ~1) : g.W;
  if (g.w_cover > 0 && g.w_cover < we) we = g.w_cover;
}
static inline long long c3_tiles2d(const Conv3& g, bool pool, int tw) {
  int he, we;
  c3_extent(g, pool, he, we);
  const int th = C3_BM / tw;
  return (long long)((we + tw - 1) / tw) * ((he + th - 1) / th);
}

template <typename T, typename OutT, int BN, int WGM, int WGN, bool FLAT, bool POOL, int ABUF, int NBUF, int TW = 32, bool SPLIT = false>
static int c3_launch(Conv3 g, hipStream_t s) {
  constexpr int C3_TW = TW, C3_TH = C3_BM / TW, C3_PW2D = C3_TW + 2;
  constexpr int NTHR = WGM * WGN * 64;
  constexpr int EP = BN * (int)sizeof(OutT) + 16;
  const int Wp = g.W + 2;
  const int rows = FLAT ? (C3_BM + 2 * Wp + 2) : (C3_TH + 2) * C3_PW2D;
  g.a_rows = (rows + 7) & ~7;
  constexpr int NWL = WGM * WGN;
  constexpr int AG_MAX = FLAT ? (61 + NWL - 1) / NWL : (43 + NWL - 1) / NWL;
  if ((g.a_rows >> 3) > AG_MAX * NWL) return fail(CTPN_ERR_ARG, "conv3x3: input window does not fit the flat-mode staging plan");
  g.tiles_n = (g.Co + BN - 1) / BN;
  long long ptiles;
  if (FLAT) {
    g.m_total = (long long)g.N * (g.H + 2) * Wp;
    ptiles = (g.m_total + C3_BM - 1) / C3_BM;
  } else {
    int he, we;
    c3_extent(g, POOL, he, we);
    g.tiles_x = (we + C3_TW - 1) / C3_TW;
    g.tiles_y = (he + C3_TH - 1) / C3_TH;
    ptiles = (long long)g.N * g.tiles_x * g.tiles_y;
  }
  const long long nblk = ptiles * g.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "conv3x3: grid out of range");
  const int main_lds = ABUF * g.a_rows * 128 + NBUF * BN * 128;
  const int epi_lds = C3_BM * EP;
  const int lds = main_lds > epi_lds ? main_lds : epi_lds;
  if (lds > 160 * 1024) return fail(CTPN_ERR_ARG, "conv3x3: LDS budget exceeded");
  auto k = conv3x3_kernel<T, OutT, BN, WGM, WGN, FLAT, POOL, ABUF, NBUF, TW, SPLIT>;
  static bool attr[C3_MAX_DEV] = {false};      // per instantiation and device
  int dev = 0, rc;
  if ((rc = c3_device(dev)) || (rc = c3_raise_lds((const void*)k, attr, dev))) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(NTHR), lds, s, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv3x3 launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}


template <typename T, bool FLAT, bool POOL, int TW, bool SPLIT = false, int BN_T = 128, int BM_T = 256, bool HT32 = false>
static int c3_launch_p(Conv3 g, hipStream_t s) {
  constexpr int BN = BN_T;
  constexpr int BMF = BM_T;                    // pixels per flat-window tile (256; 64: the one-image form)
  const int Wp = g.W + 2;
  const int rows = FLAT ? (BMF + 2 * Wp + 2) : (C3_BM / TW + 2) * (TW + 2);
  g.a_rows = (rows + 7) & ~7;
  constexpr int AG_MAX = FLAT ? ((BMF == 64 ? 37 : 61) + 7) / 8 : (43 + 7) / 8;
  if ((g.a_rows >> 3) > AG_MAX * 8) return fail(CTPN_ERR_ARG, "conv3x3: input window does not fit the staging plan");
  g.tiles_n = (g.Co + BN - 1) / BN;
  long long ptiles;
  if (FLAT) {
    g.m_total = (long long)g.N * (g.H + 2) * Wp;
    ptiles = (g.m_total + BMF - 1) / BMF;
  } else {
    int he, we;
    c3_extent(g, POOL, he, we);
    g.tiles_x = (we + TW - 1) / TW;
    g.tiles_y = (he + C3_BM / TW - 1) / (C3_BM / TW);
    ptiles = (long long)g.N * g.tiles_x * g.tiles_y;
    // Stacked tile rows (no fused pool): the bordered NHWC batch is ONE tall image of N (H + 2) rows -- image i's bottom border row is
    // followed by image i + 1's top border row, which is exactly the zero halo both need -- so tile rows can run over the batch's
    // N (H + 2) - 2 rows instead of restarting per image: 75-row maps in 16-row patches pay 80 rows per image, stacked 2462 rows pay 2464
    // (conv4_1 / conv4_2: 1078 instead of 1120 tiles). Border rows that fall inside a tile are computed and not stored (the output's
    // borders must stay zero).
    g.stacked = 0;
    if constexpr (!POOL) {
      const int th = C3_BM / TW;
      const long long ty_st = ((long long)g.N * (g.H + 2) - 2 + th - 1) / th;
      if (g.H >= 16 && g.N > 1 && ty_st < (long long)g.N * g.tiles_y && ty_st * g.tiles_x < 0x7fffffffLL) {
        g.stacked = 1;
        g.tiles_y = (int)ty_st;
        ptiles = (long long)g.tiles_x * g.tiles_y;
      }
    }
  }
  g.ptiles_total = ptiles * g.tiles_n;
#ifdef CTPN_ABLATION
  { static const int abl = [] { const char* e = std::getenv("CTPN_C3_P_ABL"); return e ? std::atoi(e) : 0; }(); g.abl = abl; }
#endif
  if (g.ptiles_total <= 0 || (long long)g.N * (g.H + 2) * Wp > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "conv3x3: problem out of range");
  const int lds = 2 * g.a_rows * 128 + 3 * BN * 128 + g.tiles_n * BN * 4;
  if (lds > 160 * 1024) return fail(CTPN_ERR_ARG, "conv3x3: LDS budget exceeded");
  int dev = 0, ncu = 0, rc;
  if ((rc = c3_device(dev)) || (rc = c3_cu_count(dev, ncu))) return rc;
  long long workers = g.ptiles_total < ncu ? g.ptiles_total : ncu;
  // half-tile tail (kernel comment): the r tiles of a last partial round as 2 r halves; a launch with at most half as many tiles as CUs
  // (small batches: conv5_x of ONE 600 x 900 image is 36 tiles) is split into halves altogether -- a half runs one wave per SIMD and takes
  // 0.59 of a tile's time. Same K order per output either way: results do not depend on the split.
  g.ht_full = g.ptiles_total; g.ht_r = 0;
  if constexpr (((FLAT || TW == 16) && BMF == 256) || HT32) {
    if (2 * g.ptiles_total <= ncu) { g.ht_full = 0; g.ht_r = (int)g.ptiles_total; workers = 2 * g.ptiles_total; }
    else {
      const long long r = g.ptiles_total % workers;
      if (r > 0 && 2 * r <= workers) { g.ht_r = (int)r; g.ht_full = g.ptiles_total - r; }
    }
  }
  // AHEAD (fragment reads one k-slice group ahead of the MFMAs through a second register set, no LDS read in flight across a barrier): EVERY
  // form since round 6. Round 3 had kept the plain form for 16 x 16 patches and flat windows (+-0 / +1 % there) -- but the plain form was only
  // as fast as it was because hipcc let the last k-slice group's reads straddle the barrier, which is a write-after-read race against the
  // LDS-DMA that recycles the strip (see the INVARIANT in the kernel; profiles/r06_barrier_war.txt). With the wait the plain form needs to be
  // correct it is 0.7 % (bf16) / 0.8 % (split) slower per batch of 32 than before; AHEAD everywhere is 0.3 % / 1.2 % FASTER than before
  // (3660 against 3645 and 1165 against 1151 images/s on one box), at 207 .. 246 VGPRs and no scratch.
  constexpr bool AH = true;
  static bool attr[C3_MAX_DEV] = {false};      // per instantiation and device
  auto k = conv3x3_p_kernel<T, T, FLAT, POOL, TW, AH, SPLIT, BN_T, BM_T, HT32>;
  if ((rc = c3_raise_lds((const void*)k, attr, dev))) return rc;
  hipLaunchKernelGGL(k, dim3((unsigned)workers), dim3(512), lds, s, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv3x3_p launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// flat windows need 256 + 2(W+2) + 2 rows per buffer; they must fit LDS twice next to nb weight strips
static inline bool c3_flat_ok(const Conv3& g, bool pool, int nb = 3) {
  const int flat_rows = (C3_BM + 2 * (g.W + 2) + 2 + 7) & ~7;
  const int bias_bytes = ((g.Co + 127) / 128) * 128 * 4;      // the persistent kernel keeps the bias vector in LDS as well
  return !pool && (g.W + 2) <= 114 && (2 * flat_rows * 128 + nb * 128 * 128 + bias_bytes) <= 160 * 1024;
}

// Kernel family for one layer. Co <= 64 (conv1_2 outside the 16-bit modes' weights-in-registers kernel): the non-persistent kernel on
// 256 x 64 tiles; Co % 128 == 0 with ReLU (every other layer of the network): persistent workgroups, flat windows where the map is small
// enough, else 8 x 32 or 16 x 16 patches -- whichever covers the map with fewer tiles; anything else (debug entry point): non-persistent
// 256 x 128 tiles.
template <typename T, bool SPLIT = false>
static int c3_dispatch(const Conv3& g, bool pool, hipStream_t s) {
  using OT = typename std::conditional<SPLIT, float, T>::type;      // staging type of the non-persistent kernel's LDS epilogue
  const bool flat = c3_flat_ok(g, pool);
  // conv1_2 in split precision (option conv_p64, default 1): the persistent kernel's 64-channel form, 3.56 ms against the
  // non-persistent kernel's 4.53 at batch 32 (same box: 1145 against 1115 images/s, profiles/r06_ab_split_conv1.txt). NOT fp32: exact-fp32
  // MFMAs are 16 x slower per flop, the per-tile fixed costs the persistent form removes are 1 % there (measured: 352.3 against 353.4
  // images/s at batch 8)
  if constexpr (SPLIT) {
    if (g.Co == 64 && g.relu && g.opt_p64 != 0)
      return pool ? c3_launch_p<T, false, true, 32, SPLIT, 64>(g, s) : c3_launch_p<T, false, false, 32, SPLIT, 64>(g, s);
  }
  if (g.Co <= 64)
    return pool ? c3_launch<T, OT, 64, 4, 1, false, true, 2, 3, 32, SPLIT>(g, s) : c3_launch<T, OT, 64, 4, 1, false, false, 2, 3, 32, SPLIT>(g, s);
  const bool persist = g.Co % 128 == 0 && g.relu;      // the persistent kernel's epilogue has the ReLU built in
  // 16 x 16 patches where they cover the map with fewer tiles. (Round 6 also tried choosing by ROUNDS of the persistent walk for one-image
  // problems -- conv3_x of one 600 x 900 image is 266 8 x 32-patch tiles on 256 CUs, two rounds for ten tiles, against 280 16 x 16-patch tiles =
  // one round + a half-tile tail: measured 45.0 / 28.2 / 45.4 / 46.7 us for conv2_2 .. conv3_3 against 39.0 / 29.3 / 48.2 / 45.5 with this
  // rule -- the 16 x 16 kernel's tile is slower than the 8 x 32 kernel's by what the tail saves; profiles/r06_timeline_sync_1image_tiling_by_rounds.txt.)
  const bool tw16 = !flat && c3_tiles2d(g, pool, 16) < c3_tiles2d(g, pool, 32);
  if (persist) {
    if (flat) {
      // one image per call (conv5_x / rpn_conv of a 600 x 900 image: 36 tiles of 256 x 128): 64-pixel x 128-channel items, one round on the machine
      int dev = 0, ncu = 0;
      const long long m_total = (long long)g.N * (g.H + 2) * (g.W + 2), tn = (g.Co + 127) / 128;
      const long long t256 = (m_total + 255) / 256 * tn, t64 = (m_total + 63) / 64 * tn;
      if (g.opt_small != 0 && 2 * (g.W + 2) + 66 <= 37 * 8 && c3_device(dev) == CTPN_OK && c3_cu_count(dev, ncu) == CTPN_OK && 2 * t256 <= ncu && t64 <= ncu)
        return c3_launch_p<T, true, false, 32, SPLIT, 128, 64>(g, s);
      return c3_launch_p<T, true, false, 32, SPLIT>(g, s);
    }
    if (tw16) return pool ? c3_launch_p<T, false, true, 16, SPLIT>(g, s) : c3_launch_p<T, false, false, 16, SPLIT>(g, s);
    {
      // one or two images per call: 8 x 32 patches with a half-tile tail where the walk's last round is at most half full (conv3_x of one
      // 600 x 900 image: 266 tiles on 256 CUs -- the ten tiles of the second round as twenty halves: 1.59 rounds instead of 2).
      // (At batch 32 -- conv3_x: 33 rounds + 64 tiles -- the same form measured -0.5 % images/s in bf16, +0.2 % in split precision, round 6 on
      // the final tree: the tail it shortens is where the forked edge kernels run. Not used there.)
      int dev = 0, ncu = 0;
      if (g.opt_small != 0 && g.N <= 2 && c3_device(dev) == CTPN_OK && c3_cu_count(dev, ncu) == CTPN_OK && ncu > 0) {
        const long long t = c3_tiles2d(g, pool, 32) * g.N * ((g.Co + 127) / 128);
        const long long G = t < ncu ? t : ncu, full = t / G, r = t % G;
        if (2 * t <= ncu || (r > 0 && 2 * r <= G && full <= 3))
          return pool ? c3_launch_p<T, false, true, 32, SPLIT, 128, 256, true>(g, s) : c3_launch_p<T, false, false, 32, SPLIT, 128, 256, true>(g, s);
      }
    }
    return pool ? c3_launch_p<T, false, true, 32, SPLIT>(g, s) : c3_launch_p<T, false, false, 32, SPLIT>(g, s);
  }
  if constexpr (SPLIT) {
    return fail(CTPN_ERR_ARG, "conv3x3 (split precision): Co must be <= 64 or a multiple of 128, with ReLU");
  } else {
    if (flat) return c3_launch<T, T, 128, 4, 2, true, false, 2, 3>(g, s);
    if (tw16) return pool ? c3_launch<T, T, 128, 4, 2, false, true, 2, 3, 16>(g, s) : c3_launch<T, T, 128, 4, 2, false, false, 2, 3, 16>(g, s);
    return pool ? c3_launch<T, T, 128, 4, 2, false, true, 2, 3>(g, s) : c3_launch<T, T, 128, 4, 2, false, false, 2, 3>(g, s);
  }
}

// ---------------------------------------------------------------------------------------------
// Ragged right edge of a 2D-tiled layer (bf16): the one or two pixel columns left of W after the largest multiple of the
// tile width (W = 225 = 7 * 32 + 1, 450 = 14 * 32 + 2, 113 = 7 * 16 + 1). A padded tile column for them would cost the layer
// 1/8 (W = 113, 225) of its tile work; the im2col GEMM (igemm.hip, 64 KB of LDS) cannot share a CU with the persistent
// workgroups (135 - 147 KB of LDS, 432 of a SIMD's 512 registers), so it only got CUs when the layer was over and
// finished 20 - 40 us after it. This kernel is made to fit in what the persistent kernels leave free: ONE wave per
// workgroup, NO LDS, <= 80 registers; the MFMA operands come straight from global memory (the [co][tap][ci] weights and the
// bordered NHWC input both have the K index contiguous, which is the 32x32x16 operand layout: lane = row, 16 bytes = 8 k).
// One wave = 32 edge pixels x 64 channels; latency-bound by design, it has the whole duration of the main launch.
// ---------------------------------------------------------------------------------------------
struct ConvEdge {
  const void* in; const void* wt; const float* bias; void* out;
  int H, W, Ci, Co, rx0, rw, relu;
  int in_pitch, out_pitch, dup_hi;   // 16-bit elements per input / output pixel (Ci / Co; split precision: 2 Ci / 2 Co or 3 Co with dup_hi)
  long long M;                 // plain: N * H * rw edge pixels, m = (n * H + y) * rw + xs
                               // pooled: N * (H / 2) * (rw / 2) POOLED edge pixels, m = (n * Ho + Y) * (rw / 2) + X; out = the pooled map
};

// POOL: the layer's fused 2x2 / 2 VALID max-pool on the edge columns (conv1_2: W = 900 = 28 * 32 + 4, conv2_2: 450 = 28 * 16 + 2 --
// an even number of edge columns starting at an even x, so the pooled pixels lie entirely inside the edge). A wave then holds
// 8 pooled pixels x their four conv pixels (lane quad = one pooled pixel: dy = bit 1, dx = bit 0 of the lane); the pool is a max
// over the quad with two DPP moves per value, and max commutes with the bias (in the sums), the ReLU and the bf16 rounding.
// DEEP (one or two images: the main launch leaves most of the machine empty and the edge kernel runs IN the layer's stream, behind the main
// launch, instead of on a forked stream -- a fork / join pair costs 12 - 19 us of cross-queue signalling per layer, 85 us of a lone image's
// millisecond): the K steps in rounds of four with three rounds in flight (36 sixteen-byte loads per lane) instead of one round of two; the
// kernel is a chain of load round trips (0.6 - 1 us each on an idle part), and a round trip now feeds 24 MFMAs instead of 4. The MFMAs are
// issued in the SAME order on the same operands: the two forms agree bit for bit, which is what lets the batch size choose between them.
// SPLIT (round 6): pixels are [hi | lo] bf16 planes, weight rows [hi | hi | lo] per tap (pack_split_kernel): three K blocks per tap --
// x_hi w_hi, x_lo w_hi, x_hi w_lo -- through the same loop (block b reads input plane b & 1 and weight block b); ReLU in fp32, then the (hi, lo)
// pair of every output (plus the hi plane once more for the layer that feeds the LSTM projection). Until round 6 split precision computed a
// padded tile column instead: an eighth of conv4_1 / conv4_2, a fifteenth of conv3_1 / conv3_2.
// KS > 1 (split precision, always): KS waves per workgroup share one 32-pixel x 64-channel tile, wave w sums K steps [w S / KS, (w + 1) S / KS)
// (deep form), waves 1 .. KS - 1 hand their partial sums to wave 0 through LDS, which adds them in wave order -- a fixed order, chosen by the
// layer's Ci alone (c3_launch_edge), so a batch and its images run alone still agree bit for bit. The split form runs in its layer's stream,
// behind the main launch. For ONE image it is a handful of workgroups, each a chain of load round trips: 432 K steps = 36 round trips for
// conv3_2, 6 with the K split -- the lone-image call in split precision 1.84 -> 1.7 ms. At batch 32 the kernel is bound by the request rate of
// its fragment loads (64 cache lines per KB of operands) and the split changes nothing (119 / 58 / 122 / 68 / 131 us per layer against
// 137 / 55 / 111 / 61 / 122).
template <typename H, bool POOL, bool DEEP = false, bool SPLIT = false, int KS = 1>
__global__ __launch_bounds__(64 * KS, DEEP ? 2 : 6) void conv3x3_edge_kernel(ConvEdge g) {
  static_assert(KS == 1 || (DEEP && SPLIT), "the K-split form is the split-precision deep form");
  constexpr int NB = SPLIT ? 3 : 1;
  const int lane = threadIdx.x & 63, l31 = lane & 31, fhalf = lane >> 5;
  const int kw = KS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  const int ntn = g.Co >> 6;
  const int tn = blockIdx.x % ntn;
  const unsigned tm = blockIdx.x / ntn;
  const int Wp = g.W + 2, Hp = g.H + 2, Ci = g.Ci;
  long long pix, opix;                                       // bordered input position of tap (0, 0); bordered output pixel
  bool mok;
  if constexpr (POOL) {
    const int Ho = g.H >> 1, Wo = g.W >> 1, rw2 = g.rw >> 1;
    unsigned m = tm * 8u + (unsigned)(l31 >> 2);             // pooled pixel of this lane's quad
    mok = m < (unsigned)g.M;
    if (!mok) m = (unsigned)g.M - 1u;
    const int X = (int)(m % (unsigned)rw2);
    const unsigned t = m / (unsigned)rw2;
    const int Y = (int)(t % (unsigned)Ho), n = (int)(t / (unsigned)Ho);
    const int y = 2 * Y + ((l31 >> 1) & 1), x = g.rx0 + 2 * X + (l31 & 1);
    pix = ((long long)n * Hp + y) * Wp + x;
    opix = ((long long)n * (Ho + 2) + Y + 1) * (Wo + 2) + (g.rx0 >> 1) + X + 1;
    mok = mok && (l31 & 3) == 0;                             // one lane of the quad stores
  } else {
    unsigned m = tm * 32u + (unsigned)l31;                   // M < 2^31 (launcher)
    mok = m < (unsigned)g.M;
    if (!mok) m = (unsigned)g.M - 1u;
    const int xs = (int)(m % (unsigned)g.rw);
    const unsigned t = m / (unsigned)g.rw;
    const int y = (int)(t % (unsigned)g.H), n = (int)(t / (unsigned)g.H);
    pix = ((long long)n * Hp + y) * Wp + g.rx0 + xs;
    opix = pix + Wp + 1;
  }
  const long long ipb = (long long)g.in_pitch * 2;          // bytes per input pixel
  const char* ip = (const char*)g.in + pix * ipb + fhalf * 16;
  const int co0 = tn * 64;
  const long long wrow = (long long)9 * NB * Ci * 2;
  const char* wp0 = (const char*)g.wt + (co0 + l31) * wrow + fhalf * 16;
  const char* wp1 = wp0 + 32 * wrow;
  c3_f32x16 acc0, acc1;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {                          // the bias is the accumulators' initial value (K-split: wave 0's)
    float4 b0 = *(const float4*)(g.bias + co0 + 8 * g4 + 4 * fhalf), b1 = *(const float4*)(g.bias + co0 + 32 + 8 * g4 + 4 * fhalf);
    if (KS > 1 && kw != 0) { b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = b0; }
    acc0[4 * g4] = b0.x; acc0[4 * g4 + 1] = b0.y; acc0[4 * g4 + 2] = b0.z; acc0[4 * g4 + 3] = b0.w;
    acc1[4 * g4] = b1.x; acc1[4 * g4 + 1] = b1.y; acc1[4 * g4 + 2] = b1.z; acc1[4 * g4 + 3] = b1.w;
  }
  const int kc_n = Ci >> 4;                                 // 16-element K steps per tap (Ci is a multiple of 64)
  if constexpr (DEEP) {
    // The 9 kc_n K steps (tap-major, the order of the loop below) in rounds of four, THREE rounds in flight: round r + 2 is requested
    // before round r's eight MFMAs are issued, so the chain is one load round trip per three rounds instead of one per round.
    constexpr int R = 4;
    const int S = 9 * NB * kc_n;                             // a multiple of 4
    c3_u32x4 xs0[R], fa0[R], fb0[R], xs1[R], fa1[R], fb1[R], xs2[R], fa2[R], fb2[R];
    auto issue = [&](c3_u32x4 (&xs)[R], c3_u32x4 (&fa)[R], c3_u32x4 (&fb)[R], int j0) {
      const int tb = j0 / kc_n, kc = j0 - tb * kc_n;        // a round never straddles two taps / K blocks: kc_n is a multiple of R
      const int tap = tb / NB, b = tb - tap * NB;
      const int ky = tap / 3, kx = tap - 3 * ky;
      const char* a = ip + (long long)(ky * Wp + kx) * ipb + (b & 1) * Ci * 2 + kc * 32;
      const char* w0 = wp0 + (long long)tb * Ci * 2 + kc * 32;
      const char* w1 = wp1 + (long long)tb * Ci * 2 + kc * 32;
#pragma unroll
      for (int q = 0; q < R; ++q) { xs[q] = *(const c3_u32x4*)(a + q * 32); fa[q] = *(const c3_u32x4*)(w0 + q * 32); fb[q] = *(const c3_u32x4*)(w1 + q * 32); }
    };
    auto mma = [&](const c3_u32x4 (&xs)[R], const c3_u32x4 (&fa)[R], const c3_u32x4 (&fb)[R]) {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        acc0 = HalfOps<H>::mfma_32x32x16(__builtin_bit_cast(uint4, fa[q]), __builtin_bit_cast(uint4, xs[q]), acc0);
        acc1 = HalfOps<H>::mfma_32x32x16(__builtin_bit_cast(uint4, fb[q]), __builtin_bit_cast(uint4, xs[q]), acc1);
      }
    };
    const int jb = kw * (S / KS), je = jb + S / KS;          // this wave's K steps: S / KS is a multiple of 4 (launcher), >= 36
    issue(xs0, fa0, fb0, jb);
    issue(xs1, fa1, fb1, jb + R);
#pragma unroll 1
    for (int j = jb; j < je; j += 3 * R) {
      if (j + 2 * R < je) issue(xs2, fa2, fb2, j + 2 * R);
      mma(xs0, fa0, fb0);
      if (j + 3 * R < je) issue(xs0, fa0, fb0, j + 3 * R);
      if (j + R < je) mma(xs1, fa1, fb1);
      if (j + 4 * R < je) issue(xs1, fa1, fb1, j + 4 * R);
      if (j + 2 * R < je) mma(xs2, fa2, fb2);
    }
    if constexpr (KS > 1) {
      __shared__ float red[KS - 1][32][64];                 // [wave][accumulator register][lane]: conflict-free 256-byte rows
      if (kw != 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { red[kw - 1][e][lane] = acc0[e]; red[kw - 1][16 + e][lane] = acc1[e]; }
      }
      __syncthreads();
      if (kw != 0) return;
#pragma unroll 1
      for (int w = 0; w < KS - 1; ++w) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] += red[w][e][lane]; acc1[e] += red[w][16 + e][lane]; }
      }
    }
  } else {
#pragma unroll 1
  for (int tb = 0; tb < 9 * NB; ++tb) {
    const int tap = tb / NB, b = tb - tap * NB;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const char* a = ip + (long long)(ky * Wp + kx) * ipb + (b & 1) * Ci * 2;
    const char* w0 = wp0 + (long long)tb * Ci * 2;
    const char* w1 = wp1 + (long long)tb * Ci * 2;
#pragma unroll 1
    for (int kc = 0; kc < kc_n; kc += 2) {                  // two K steps per round: six 16-byte loads in flight per lane
      const c3_u32x4 x = *(const c3_u32x4*)(a + kc * 32), x2 = *(const c3_u32x4*)(a + kc * 32 + 32);
      const c3_u32x4 f0 = *(const c3_u32x4*)(w0 + kc * 32), f2 = *(const c3_u32x4*)(w0 + kc * 32 + 32);
      const c3_u32x4 f1 = *(const c3_u32x4*)(w1 + kc * 32), f3 = *(const c3_u32x4*)(w1 + kc * 32 + 32);
      acc0 = HalfOps<H>::mfma_32x32x16(__builtin_bit_cast(uint4, f0), __builtin_bit_cast(uint4, x), acc0);
      acc1 = HalfOps<H>::mfma_32x32x16(__builtin_bit_cast(uint4, f1), __builtin_bit_cast(uint4, x), acc1);
      acc0 = HalfOps<H>::mfma_32x32x16(__builtin_bit_cast(uint4, f2), __builtin_bit_cast(uint4, x2), acc0);
      acc1 = HalfOps<H>::mfma_32x32x16(__builtin_bit_cast(uint4, f3), __builtin_bit_cast(uint4, x2), acc1);
    }
  }
  }
  if constexpr (POOL) {
    // quad max BEFORE any lane leaves: lane ^ 1 (quad_perm [1,0,3,2] = 0xB1), then lane ^ 2 ([2,3,0,1] = 0x4E). The moved values are
    // pinned with an empty asm: cross-lane results that only feed an exec-masked store are otherwise sunk into the masked block
    auto qmax = [](float v) -> float {
      float a = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
      asm volatile("" : "+v"(a));
      v = __builtin_fmaxf(v, a);
      float b = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
      asm volatile("" : "+v"(b));
      return __builtin_fmaxf(v, b);
    };
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = qmax(acc0[e]); acc1[e] = qmax(acc1[e]); }
  }
  if (!mok) return;
  char* op = (char*)g.out + (opix * g.out_pitch + co0 + 4 * fhalf) * 2;
  if constexpr (SPLIT) {
    const long long plane = (long long)g.Co * 2;
#pragma unroll
    for (int half64 = 0; half64 < 2; ++half64) {
      const c3_f32x16& a = half64 ? acc1 : acc0;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = g.relu ? __builtin_fmaxf(a[4 * g4 + e], 0.f) : a[4 * g4 + e];
        uint32_t h0, l0, h1, l1;
        ctpn_split_pk_bf16(v[0], v[1], h0, l0);
        ctpn_split_pk_bf16(v[2], v[3], h1, l1);
        char* d = op + 64 * half64 + 16 * g4;
        *(uint2*)d = make_uint2(h0, h1);
        *(uint2*)(d + plane) = make_uint2(l0, l1);
        if (g.dup_hi) *(uint2*)(d + 2 * plane) = make_uint2(h0, h1);
      }
    }
    return;
  }
  auto pk = [&](float lo, float hi) -> uint32_t {
    const uint32_t p = c3_cvt_pk<H>(lo, hi);
    if (!g.relu) return p;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), s16x2{0, 0}));   // bf16 ReLU on the packed pair
  };
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    *(uint2*)(op + 16 * g4) = make_uint2(pk(acc0[4 * g4], acc0[4 * g4 + 1]), pk(acc0[4 * g4 + 2], acc0[4 * g4 + 3]));
    *(uint2*)(op + 64 + 16 * g4) = make_uint2(pk(acc1[4 * g4], acc1[4 * g4 + 1]), pk(acc1[4 * g4 + 2], acc1[4 * g4 + 3]));
  }
}

// r edge columns [w - r, w) of an h x w layer; pooled: r even, w - r even, `out` is the pooled map ((h / 2 + 2) x (w / 2 + 2) bordered)
template <typename H, bool SPLIT = false>
static int c3_launch_edge(const void* in, const void* wt, const float* bias, void* out, int n, int h, int w, int ci, int co, int relu, int r,
                          bool pooled, hipStream_t s, bool deep, int dup_hi = 0) {
  ConvEdge e{};
  e.in = in; e.wt = wt; e.bias = bias; e.out = out; e.H = h; e.W = w; e.Ci = ci; e.Co = co; e.rx0 = w - r; e.rw = r; e.relu = relu;
  e.in_pitch = SPLIT ? 2 * ci : ci; e.out_pitch = SPLIT ? (dup_hi ? 3 : 2) * co : co; e.dup_hi = SPLIT && dup_hi ? 1 : 0;
  if (pooled && ((r & 1) || ((w - r) & 1) || h < 2)) return fail(CTPN_ERR_ARG, "conv3x3 edge: pooled edge needs even columns");
  e.M = pooled ? (long long)n * (h / 2) * (r / 2) : (long long)n * h * r;
  const long long per_wave = pooled ? 8 : 32;
  const long long nblk = ((e.M + per_wave - 1) / per_wave) * (co / 64);
  if (nblk <= 0 || nblk > 0x7fffffffLL || e.M > 0x7fffffffLL || !bias) return fail(CTPN_ERR_ARG, "conv3x3 edge: problem out of range");
  if constexpr (SPLIT) {
    // K split over waves, by Ci alone: S = 27 Ci / 16 K steps = 108 / 216 / 432 for Ci = 64 / 128 / 256 -> 3 / 6 / 6 waves of 36 / 36 / 72 steps
    const int S = 27 * (ci / 16), ks = S % 24 == 0 && S / 6 >= 36 ? 6 : (S % 12 == 0 && S / 3 >= 36 ? 3 : 1);
    if (deep && ks > 1) {
      if (ks == 6) {
        if (pooled) hipLaunchKernelGGL((conv3x3_edge_kernel<H, true, true, true, 6>), dim3((unsigned)nblk), dim3(64 * 6), 0, s, e);
        else hipLaunchKernelGGL((conv3x3_edge_kernel<H, false, true, true, 6>), dim3((unsigned)nblk), dim3(64 * 6), 0, s, e);
      } else {
        if (pooled) hipLaunchKernelGGL((conv3x3_edge_kernel<H, true, true, true, 3>), dim3((unsigned)nblk), dim3(64 * 3), 0, s, e);
        else hipLaunchKernelGGL((conv3x3_edge_kernel<H, false, true, true, 3>), dim3((unsigned)nblk), dim3(64 * 3), 0, s, e);
      }
      hipError_t errk = hipGetLastError();
      if (errk != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv3x3 edge launch: ") + hipGetErrorString(errk));
      return CTPN_OK;
    }
  }
  if (deep) {
    if (pooled) hipLaunchKernelGGL((conv3x3_edge_kernel<H, true, true, SPLIT>), dim3((unsigned)nblk), dim3(64), 0, s, e);
    else hipLaunchKernelGGL((conv3x3_edge_kernel<H, false, true, SPLIT>), dim3((unsigned)nblk), dim3(64), 0, s, e);
  } else if (pooled) hipLaunchKernelGGL((conv3x3_edge_kernel<H, true, false, SPLIT>), dim3((unsigned)nblk), dim3(64), 0, s, e);
  else hipLaunchKernelGGL((conv3x3_edge_kernel<H, false, false, SPLIT>), dim3((unsigned)nblk), dim3(64), 0, s, e);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv3x3 edge launch: ") + hipGetErrorString(err));
  return CTPN_OK;
}

// ---- per-type entry points: one translation unit each (conv3x3_<type>.hip), called by launch_conv3x3 (conv3x3.hip) ----
int c3_run_f32(const Conv3& g, bool pool, hipStream_t s);                    // exact-fp32 MFMA kernels
int c3_run_bf16(const Conv3& g, bool pool, bool wr, hipStream_t s);          // wr: the weights-in-registers kernel (Ci = 64)
int c3_run_f16(const Conv3& g, bool pool, bool wr, hipStream_t s);
int c3_run_split(const Conv3& g, bool pool, hipStream_t s);                  // (hi, lo) bf16 planes, three MFMA terms
int c3_edge_bf16(const void* in, const void* wt, const float* bias, void* out, int n, int h, int w, int ci, int co, int relu, int r, bool pooled, hipStream_t s, bool deep);
int c3_edge_f16(const void* in, const void* wt, const float* bias, void* out, int n, int h, int w, int ci, int co, int relu, int r, bool pooled, hipStream_t s, bool deep);
int c3_edge_split(const void* in, const void* wt, const float* bias, void* out, int n, int h, int w, int ci, int co, int relu, int r, bool pooled, hipStream_t s, bool deep, int dup_hi);

}  // namespace ctpn
