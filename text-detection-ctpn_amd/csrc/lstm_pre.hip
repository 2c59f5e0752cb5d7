// LSTM input projection of the 16-bit modes: lstm_pre[m][1024] = rpn_conv[m][512] @ Wx^T + b for every feature-map cell m, fp16 out.
//
// Replaces the x_t @ kernel[:512] + bias half of tf.contrib.rnn.LSTMCell inside Network.Bilstm (reference lib/networks/network.py:97-100),
// hoisted out of the recurrence for all rows and time steps at once (both directions: 2 x 512 gate columns, permuted at weight-pack time
// into the order the recurrent kernels read, bilstm.hip).
//
// Why not igemm.hip (rounds 1-3: 128 x 128 tiles, 4144 short workgroups, 170 us = 16 % of the MFMA peak): with K = 512 a tile is eight K
// steps between an exposed prologue and a 64 KB epilogue, and every tile re-fetches its 128 KB slice of Wx. Here the WEIGHT SLICE IS
// RESIDENT: a workgroup owns 128 of the 1024 gate columns, keeps their [128][512] 16-bit weights in LDS (128 KB) for the whole launch and
// walks M tiles of 128 cells; only the activation tile streams (two 16 KB buffers, LDS-DMA, chunk c + 1 in flight under chunk c's MFMAs,
// across tile boundaries). The eight column slices of an M range run on ONE XCD (consecutive workers), so the activation tile is fetched
// from HBM once and hit seven times in that XCD's L2. Epilogue from registers: bias from the accumulators' initial value, packed converts,
// v_permlane32_swap -> 16-byte stores; no LDS round trip.
#include "common.h"

namespace ctpn {

typedef uint32_t lp_u32x4 __attribute__((ext_vector_type(4)));

struct LstmPre {
  const void* a;        // bordered NHWC 16-bit activations: n x (hf + 2) x (wf + 2) x 512
  const void* wt;       // [1024][512] 16-bit (gate rows permuted)
  const float* bias;    // [1024]
  void* out;            // [M][1024] fp16
  long long M;
  int hf, wf;
  int mtiles, mgroups;  // ceil(M / 128); M ranges (workers per column slice)
};

template <typename H>
__global__ __launch_bounds__(512) void lstm_pre_kernel(LstmPre g) {
  constexpr int BM = 128, BN = 128, KC = 8;                       // 8 K chunks of 64
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sB = smem;                                           // [8 chunks][128 rows][128 B], 16-byte slots XOR-swizzled by (row >> 1) & 7
  char* const sA = smem + KC * BN * 128;                           // 2 x [128 rows][128 B], same swizzle
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                         // 4 x 2 waves: 32 cells x 64 columns each
  // worker -> (column slice, M range): the hardware deals consecutive workgroup ids round-robin over the 8 XCDs; the 8 slices of an M
  // range share an XCD (and with it the L2 copy of every activation tile)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tn = j & 7;
  const int per_xcd = gridDim.x >> 6;                              // M ranges per XCD (grid = 64 * per_xcd)
  const int mg = xcd * per_xcd + (j >> 3);
  const int n0 = tn * BN;
  const int srow = lane >> 3, sslot = lane & 7;
  const int Wp = g.wf + 2, Hp = g.hf + 2;

  // ---- resident weight slice: 128 rows x 512 k, once ----
  {
    const char* wb = (const char*)g.wt + (size_t)n0 * 1024;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int grp = wave + i * 8, row = grp * 8 + srow;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + (size_t)row * 1024 + c * 128 + ((sslot ^ ((row >> 1) & 7)) << 4)),
                                         (__attribute__((address_space(3))) void*)(sB + c * (BN * 128) + grp * 1024), 16, 0, 0);
      }
  }
  // bias of this lane's 2 x 16 output columns (C layout: lane = cell column l31, rows = gate columns 8 g4 + 4 fhalf + e)
  const int l31 = lane & 31, fhalf = lane >> 5;
  ctpn_f32x16 bias16[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b4 = *(const float4*)(g.bias + n0 + wn * 64 + i * 32 + 8 * g4 + 4 * fhalf);
      bias16[i][4 * g4] = b4.x; bias16[i][4 * g4 + 1] = b4.y; bias16[i][4 * g4 + 2] = b4.z; bias16[i][4 * g4 + 3] = b4.w;
    }

  // activation rows of one tile: this lane's two 16-byte sources (rows grp * 8 + srow, grp = wave, wave + 8)
  auto a_src = [&](int mt, long long (&off)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave + i * 8) * 8 + srow;
      long long m = (long long)mt * BM + row;
      if (m > g.M - 1) m = g.M - 1;
      const long long hw = (long long)g.hf * g.wf;
      const long long n = m / hw;
      const int rem = (int)(m - n * hw);
      const int y = rem / g.wf, x = rem - y * g.wf;
      off[i] = (((n * Hp + y + 1) * Wp + x + 1) * 512) * 2 + ((sslot ^ ((row >> 1) & 7)) << 4);
    }
  };
  auto issue_a = [&](const long long (&off)[2], int c, int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)g.a + off[i] + c * 128),
                                       (__attribute__((address_space(3))) void*)(sA + buf * (BM * 128) + (wave + i * 8) * 1024), 16, 0, 0);
  };

  const int fsw = (l31 >> 1) & 7;
  int mt = mg;
  if (mt >= g.mtiles) { __syncthreads(); return; }
  long long cur_off[2], nxt_off[2];
  a_src(mt, cur_off);
  issue_a(cur_off, 0, 0);
  __syncthreads();                                                 // weights + first chunk landed (hipcc drains vmcnt at the barrier)
  int buf = 0;
  for (;;) {
    const int mt_next = mt + g.mgroups;
    const bool has_next = mt_next < g.mtiles;
    a_src(has_next ? mt_next : mt, nxt_off);
    ctpn_f32x16 acc[2] = {bias16[0], bias16[1]};
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (c + 1 < KC) issue_a(cur_off, c + 1, buf ^ 1);
      else issue_a(nxt_off, 0, buf ^ 1);                           // next tile's first chunk (a harmless re-fetch behind the last tile)
      const char* sa = sA + buf * (BM * 128) + (wm * 32 + l31) * 128;
      const char* sb = sB + c * (BN * 128) + (wn * 64 + l31) * 128;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int slot = ((2 * q + fhalf) ^ fsw) << 4;
        const uint4 xf = *(const uint4*)(sa + slot);
        const uint4 w0 = *(const uint4*)(sb + slot), w1 = *(const uint4*)(sb + 32 * 128 + slot);
        acc[0] = HalfOps<H>::mfma_32x32x16(w0, xf, acc[0]);
        acc[1] = HalfOps<H>::mfma_32x32x16(w1, xf, acc[1]);
      }
      __syncthreads();
      buf ^= 1;
    }
    // ---- epilogue: this lane's cell m, gate columns n0 + 64 wn + 32 i + 8 g4 + 4 fhalf + e -> fp16, 16-byte stores ----
    const long long m = (long long)mt * BM + wm * 32 + l31;
    if (m < g.M) {
      char* ob = (char*)g.out + (m * 1024 + n0 + wn * 64) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t e0 = ctpn_cvt_pk_f16(acc[i][8 * q + 0], acc[i][8 * q + 1]), e1 = ctpn_cvt_pk_f16(acc[i][8 * q + 2], acc[i][8 * q + 3]);
          const uint32_t o0 = ctpn_cvt_pk_f16(acc[i][8 * q + 4], acc[i][8 * q + 5]), o1 = ctpn_cvt_pk_f16(acc[i][8 * q + 6], acc[i][8 * q + 7]);
          const auto r0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);   // low lanes: channel group 2q complete, high lanes: 2q + 1
          const auto r1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
          *(lp_u32x4*)(ob + (i * 32 + 16 * q + 8 * fhalf) * 2) = lp_u32x4{r0[0], r1[0], r0[1], r1[1]};
        }
    }
    if (!has_next) break;
    mt = mt_next;
    cur_off[0] = nxt_off[0]; cur_off[1] = nxt_off[1];
  }
}

// a: bordered NHWC 16-bit map of n x hf x wf cells x 512 channels (dtype t: BF16 or F16); out: [n * hf * wf][1024] fp16
int launch_lstm_pre(const void* a, const void* wt, const float* bias, void* out, DType t, int n, int hf, int wf, hipStream_t s) {
  if (!dtype_is_half(t)) return fail(CTPN_ERR_ARG, "lstm_pre: 16-bit modes only");
  LstmPre g{};
  g.a = a; g.wt = wt; g.bias = bias; g.out = out; g.M = (long long)n * hf * wf; g.hf = hf; g.wf = wf;
  if (g.M <= 0 || g.M > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "lstm_pre: problem out of range");
  g.mtiles = (int)((g.M + 127) / 128);
  int dev = 0, ncu = 0, rc;
  if ((rc = current_device(dev)) || (rc = device_cu_count(dev, ncu))) return rc;
  int per_xcd = ncu / 64;                                          // M ranges per XCD: 8 column slices x per_xcd workgroups on each of 8 XCDs
  if (per_xcd < 1) per_xcd = 1;
  while (per_xcd > 1 && 8 * (per_xcd - 1) >= g.mtiles) --per_xcd;  // small problems: no idle M ranges
  g.mgroups = 8 * per_xcd;
  const int lds = 8 * 128 * 128 + 2 * 128 * 128;
  auto launch = [&](auto kern) -> int {
    static bool done[CTPN_MAX_DEV] = {false};
    const int r = raise_dynamic_lds((const void*)kern, lds, done, dev);
    if (r) return r;
    hipLaunchKernelGGL(kern, dim3(64 * per_xcd), dim3(512), lds, s, g);
    return CTPN_OK;
  };
  rc = t == DType::F16 ? launch(lstm_pre_kernel<h_f16>) : launch(lstm_pre_kernel<h_bf16>);
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("lstm_pre launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
