// 3x3 convolution + bias + ReLU (+ 2x2 max-pool) in fp16 through the 1-D Winograd transform F(2, 3) along x: the fast form of the
// correctness-first reference in winograd.hip, built on conv3x3_p_kernel's persistent pipeline (conv3x3_impl.h). CTPN_PREC_FP16W runs the
// four K >= 1152 layers that tile as 8 x 32 patches (conv2_2, conv3_1, conv3_2, conv3_3: 39 % of the network's multiplies) through it.
//
// Replaces tf.nn.conv2d + bias_add + relu (+ max_pool) of Network.conv (reference lib/networks/network.py:160-196). Arithmetic
// specification: oracle/winograd.py (conv3x3_relu_winograd_x), fp16 roundings:
//   U[ky][f]      = G[f][:] . w[ky][:]            from the fp32 weights, rounded ONCE to fp16 (wino_pack_kernel)
//   V[r][t][f]    = B^T[f][:] . d[r][2t .. 2t+3]   ONE v_pk_add_f16 per dword on fp16 activations (correctly rounded sums / differences)
//   m_f[y][t]     = sum_ky sum_ci V[y+ky][t][f][ci] U[ky][f][ci][co]            v_mfma_f32_32x32x16_f16, fp32 accumulate
//   out[y][2t]    = relu(m0 + m1 + m2 + b),  out[y][2t+1] = relu(m1 - m2 - m3 + b)     fp32 -> fp16
// 12 Ci multiplies per output pair instead of 18: 2 / 3 of the direct form's MFMAs for the same algorithmic work. Under this part's power
// limit (the conv stack runs at the package cap) fewer multiplies are the one lever that still converts into time.
//
// What changes against conv3x3_p_kernel<8 x 32 patches>:
//   * a 32-column MFMA tile is 32 output PAIRS -- two patch rows of 16 pairs -- so a wave (4 row pairs x 2 channel halves per workgroup) owns
//     64 pixels x 64 channels x 4 frequencies: 8 accumulator tiles = 128 VGPRs;
//   * the window is staged DE-INTERLEAVED: patch row r holds its 17 even columns, then its 17 odd ones (the LDS-DMA's per-lane source
//     offsets make any row permutation free). Pair t then reads d0, d2 from rows t, t + 1 of the even half and d1, d3 from rows t, t + 1
//     of the odd half -- unit stride across lanes; with the second row's lanes rotated by two pairs (34 = 2 mod 16, as for 16 x 16 patches)
//     every ds_read_b128 lane group lands on 16 distinct rows mod 16: conflict-free (tests/test_layouts.py);
//   * a K step is (ky, 16-channel slice): four window fragments -> four V fragments (16 packed adds) -> 8 MFMAs against the slice's
//     weights for all four frequencies. The weight strip of a step is [2 k-halves][4 f][128 co][16 B] = 16 KB, pre-arranged by the pack
//     kernel so that the LDS-DMA is a straight copy and the fragment reads are conflict-free without a swizzle; 12 steps per 64-channel
//     chunk (9 in the direct form), same three strip buffers, prefetch distance 2, counted vmcnt;
//   * the epilogue applies A^T in registers: the bias rides on m1 (it enters both outputs with +1), a lane stores TWO adjacent pixels;
//     the fused pool's horizontal partner is the lane's own second pixel, its vertical partner the other patch row (ds_bpermute).
#include "conv3x3_impl.h"

namespace ctpn {

typedef _Float16 wx_h2 __attribute__((ext_vector_type(2)));

struct ConvWx {
  const void* in;       // bordered NHWC fp16
  const void* u;        // wino_pack_kernel's output
  const float* bias;
  void* out;            // bordered NHWC fp16 (may be null with POOL)
  void* pool_out;
  int N, H, W, Ci, Co;
  int tiles_x, tiles_y, tiles_n;
  long long ptiles_total;
};

constexpr int WX_PW = 34, WX_ROWS = 10 * WX_PW, WX_AROWS = 344, WX_STRIP = 16384, WX_STEPS = 12;

// U: [tn = co / 128][chunk = ci / 64][ky][q = 16-channel slice][k-half][f][co % 128][8 ci] fp16
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w_hwio, uint16_t* __restrict__ u, int Ci, int Co) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)Co * 12 * Ci;
  if (idx >= total) return;
  const int j = (int)(idx & 7), co_l = (int)((idx >> 3) & 127), f = (int)((idx >> 10) & 3), half = (int)((idx >> 12) & 1), q = (int)((idx >> 13) & 3);
  long long rest = idx >> 15;                                     // (tn, chunk, ky)
  const int ky = (int)(rest % 3); rest /= 3;
  const int nchunks = Ci >> 6;
  const int chunk = (int)(rest % nchunks), tn = (int)(rest / nchunks);
  const int ci = chunk * 64 + q * 16 + half * 8 + j, co = tn * 128 + co_l;
  const double g0 = w_hwio[((size_t)(ky * 3 + 0) * Ci + ci) * Co + co], g1 = w_hwio[((size_t)(ky * 3 + 1) * Ci + ci) * Co + co],
               g2 = w_hwio[((size_t)(ky * 3 + 2) * Ci + ci) * Co + co];
  const double d = f == 0 ? g0 : f == 1 ? 0.5 * g0 + 0.5 * g1 + 0.5 * g2 : f == 2 ? 0.5 * g0 - 0.5 * g1 + 0.5 * g2 : g2;
  u[idx] = __builtin_bit_cast(unsigned short, (_Float16)(float)d);
}

__device__ __forceinline__ uint4 wx_sub(const uint4& a, const uint4& b) {
  uint4 r;
  r.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.x) - __builtin_bit_cast(wx_h2, b.x));
  r.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.y) - __builtin_bit_cast(wx_h2, b.y));
  r.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.z) - __builtin_bit_cast(wx_h2, b.z));
  r.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.w) - __builtin_bit_cast(wx_h2, b.w));
  return r;
}
__device__ __forceinline__ uint4 wx_add(const uint4& a, const uint4& b) {
  uint4 r;
  r.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.x) + __builtin_bit_cast(wx_h2, b.x));
  r.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.y) + __builtin_bit_cast(wx_h2, b.y));
  r.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.z) + __builtin_bit_cast(wx_h2, b.z));
  r.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wx_h2, a.w) + __builtin_bit_cast(wx_h2, b.w));
  return r;
}

template <bool POOL>
__global__ __launch_bounds__(512) void conv3x3_wx_kernel(ConvWx g) {
  constexpr int BN = 128, NW = 8, B_LOADS = 2, AG_MAX = 6;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x, bid = blockIdx.x;
  const int xq = G >> 3, xr = G & 7, xcd = bid & 7;
  const int w0 = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);   // XCD-mates walk neighbouring tiles
  const long long total = g.ptiles_total;
  if (w0 >= total) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int Wp = g.W + 2, Hp = g.H + 2;
  constexpr int a_bytes = WX_AROWS * 128;
  char* const sA = smem;                                   // 2 windows
  char* const sB = smem + 2 * a_bytes;                     // 3 weight strips
  float* const sbias = (float*)(smem + 2 * a_bytes + 3 * WX_STRIP);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int srow = lane >> 3, sslot = lane & 7;
  const int l31 = lane & 31, fhalf = lane >> 5;
  const int nchunks = g.Ci >> 6;
  const int pix_bytes = g.Ci * 2;

  for (int i = tid; i < g.tiles_n * BN; i += 512) sbias[i] = i < g.Co ? g.bias[i] : 0.f;

  // window staging: LDS row R = patch row R / 34, slot s = R % 34: column 2 s (s < 17) or 2 (s - 17) + 1 -- even columns first
  uint32_t aoff[AG_MAX];
#pragma unroll
  for (int i = 0; i < AG_MAX; ++i) {
    int grp = wave + i * NW;
    if (grp > WX_AROWS / 8 - 1) grp = WX_AROWS / 8 - 1;    // every wave issues every slot (duplicates of the last group)
    const int R = grp * 8 + srow;
    const int i2 = R / WX_PW, s = R - i2 * WX_PW;
    const int j2 = s < 17 ? 2 * s : 2 * (s - 17) + 1;
    aoff[i] = (uint32_t)((i2 * Wp + j2) * pix_bytes + ((sslot ^ ((R >> 1) & 7)) << 4));
  }
  const uint32_t boff0 = (uint32_t)(wave * 1024 + lane * 16), boff1 = boff0 + 8 * 1024;

  struct Tile { int n0, img, y0, x0; const char* ab; const char* bb; };
  auto usg = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  auto spin = [&](const char* base, long long byte_off) -> const char* {
    const unsigned long long a = (unsigned long long)(uintptr_t)base + (unsigned long long)byte_off;
    const unsigned lo = usg((unsigned)a), hi = usg((unsigned)(a >> 32));
    return (const char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
  };
  auto setup = [&](long long lid, Tile& t) {
    const int tn = (int)(lid % g.tiles_n);
    const long long pt = lid / g.tiles_n;
    const int per_img = g.tiles_x * g.tiles_y;
    t.n0 = tn * BN;
    t.img = (int)(pt / per_img);
    const int rem = (int)(pt - (long long)t.img * per_img);
    const int tyi = rem / g.tiles_x;
    t.y0 = tyi * 8;
    t.x0 = (rem - tyi * g.tiles_x) * 32;
    t.ab = spin((const char*)g.in, (((long long)t.img * Hp + t.y0) * Wp + t.x0) * pix_bytes);
    t.bb = spin((const char*)g.u, (long long)tn * nchunks * WX_STEPS * WX_STRIP);
  };
  auto issue_a_group = [&](int i, const char* ab, int chunk, int buf) {
    int grp = wave + i * NW;
    if (grp > WX_AROWS / 8 - 1) grp = WX_AROWS / 8 - 1;
    c3_glds16_saddr(ab + chunk * 128, aoff[i], __builtin_amdgcn_readfirstlane(lds0 + buf * a_bytes + grp * 1024));
  };
  auto issue_b = [&](const char* bb, int chunk, int t, int buf) {      // strip of step t = (ky = t / 4, q = t % 4) of `chunk`: a straight 16 KB copy
    const char* sb = bb + ((long long)chunk * WX_STEPS + t) * WX_STRIP;
    c3_glds16_saddr(sb, boff0, __builtin_amdgcn_readfirstlane(lds0 + 2 * a_bytes + buf * WX_STRIP + wave * 1024));
    c3_glds16_saddr(sb, boff1, __builtin_amdgcn_readfirstlane(lds0 + 2 * a_bytes + buf * WX_STRIP + (wave + 8) * 1024));
  };

  // this lane's pair: patch row 2 wm + (l31 >> 4), pair tp (the second row's lanes rotated by two pairs: conflict-free reads)
  const int rowsel = l31 >> 4, tp = c3_tw16_col(l31);
  const int rbase = (2 * wm + rowsel) * WX_PW + tp;       // LDS row of d0 at ky = 0
  c3_f32x16 acc[4][2];

  auto compute = [&](int abuf, int bbuf, auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int ky = t / 4, q = t % 4;
    const char* sa = sA + abuf * a_bytes;
    const char* sb = sB + bbuf * WX_STRIP + fhalf * 8192 + (wn * 64 + l31) * 16;
    const int slot = 2 * q + fhalf;
    const int R0 = rbase + ky * WX_PW;
    auto rd = [&](int R) -> uint4 { return *(const uint4*)(sa + R * 128 + ((slot ^ ((R >> 1) & 7)) << 4)); };
    const uint4 d0 = rd(R0), d2 = rd(R0 + 1), d1 = rd(R0 + 17), d3 = rd(R0 + 18);
    // V_f right in front of its two MFMAs (one transformed fragment live at a time, not four)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint4 v = f == 0 ? wx_sub(d0, d2) : f == 1 ? wx_add(d1, d2) : f == 2 ? wx_sub(d2, d1) : wx_sub(d1, d3);
      const uint4 u0 = *(const uint4*)(sb + f * 2048), u1 = *(const uint4*)(sb + f * 2048 + 512);
      acc[f][0] = HalfOps<h_f16>::mfma_32x32x16(u0, v, acc[f][0]);
      acc[f][1] = HalfOps<h_f16>::mfma_32x32x16(u1, v, acc[f][1]);
    }
  };

  Tile cur, nxt;
  long long lid = w0;
  setup(lid, cur);
#pragma unroll
  for (int i = 0; i < AG_MAX; ++i) issue_a_group(i, cur.ab, 0, 0);
  issue_b(cur.bb, 0, 0, 0);
  issue_b(cur.bb, 0, 1, 1);
  c3_wait_vm<B_LOADS>();
  __syncthreads();            // also publishes sbias
  int wpar = 0;

  auto step = [&](auto tc, auto lastc, int c) {
    constexpr int t = decltype(tc)::value;
    constexpr bool last = decltype(lastc)::value;
    constexpr int nA = t < AG_MAX ? 1 : 0;                 // one window slice of the NEXT chunk per step, in steps 0 .. 5
    if constexpr (t + 2 < WX_STEPS) issue_b(cur.bb, c, t + 2, (t + 2) % 3);
    else if constexpr (last) issue_b(nxt.bb, 0, t + 2 - WX_STEPS, (t + 2) % 3);
    else issue_b(cur.bb, c + 1, t + 2 - WX_STEPS, (t + 2) % 3);
    if constexpr (t < AG_MAX) {
      if constexpr (last) issue_a_group(t, nxt.ab, 0, wpar ^ 1);
      else issue_a_group(t, cur.ab, c + 1, wpar ^ 1);
    }
    compute(wpar, t % 3, tc);
    c3_wait_vm<B_LOADS + nA>();
    __builtin_amdgcn_s_barrier();
  };
  auto chunk = [&](auto lastc, int c) {
    c3_static_for<WX_STEPS>([&](auto tc) { step(tc, lastc, c); });
    wpar ^= 1;
  };

  for (;;) {
    const long long nlid = lid + G;
    const bool has_next = nlid < total;
    setup(has_next ? nlid : lid, nxt);
    // accumulators: the bias enters through m1 (A^T has +1 for it in both outputs), the other frequencies start at zero
    {
      const float* bl = sbias + cur.n0 + wn * 64 + 4 * fhalf;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const c3_f32x4 bv = *(const c3_f32x4*)(bl + i * 32 + 8 * g4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[0][i][4 * g4 + e] = 0.f; acc[1][i][4 * g4 + e] = bv[e]; acc[2][i][4 * g4 + e] = 0.f; acc[3][i][4 * g4 + e] = 0.f; }
        }
    }
    for (int c = 0; c + 1 < nchunks; ++c) chunk(std::false_type{}, c);
    chunk(std::true_type{}, nchunks - 1);

    // ---- epilogue: out0 = m0 + m1 + m2, out1 = m1 - m2 - m3 for pixels x0 + 2 tp, + 1 of row y0 + 2 wm + rowsel ----
    typedef short wx_s16x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) char* wx_gptr;
    auto relu_pk = [](uint32_t p) -> uint32_t {
      const wx_s16x2 z = {0, 0};
      return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(wx_s16x2, p), z));
    };
    // 16 fp32 values (channels 8 g4 + 4 fhalf + e of one pixel) -> two 16-byte stores of 8 channels each
    auto store16 = [&](wx_gptr dst, bool ok, const c3_f32x16& a) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint32_t e0 = relu_pk(ctpn_cvt_pk_f16(a[8 * q + 0], a[8 * q + 1])), e1 = relu_pk(ctpn_cvt_pk_f16(a[8 * q + 2], a[8 * q + 3]));
        const uint32_t o0 = relu_pk(ctpn_cvt_pk_f16(a[8 * q + 4], a[8 * q + 5])), o1 = relu_pk(ctpn_cvt_pk_f16(a[8 * q + 6], a[8 * q + 7]));
        const auto r0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);   // low lanes: even group complete, high lanes: odd group
        const auto r1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
        const c3_u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
        if (ok) *(__attribute__((address_space(1))) c3_u32x4*)(dst + (16 * q + 8 * fhalf) * 2) = v;
      }
    };
    int lq = l31;
    asm volatile("" : "+v"(lq));                          // (address terms recomputed per tile on purpose, see conv3x3_p_kernel)
    const int tq = c3_tw16_col(lq);
    const int ch0 = cur.n0 + wn * 64;
    const int yl = 2 * wm + (lq >> 4), xl = 2 * tq;
    if (g.out) {
      const bool rok = cur.y0 + yl < g.H;
      const long long pix = ((long long)cur.img * Hp + cur.y0 + yl + 1) * Wp + cur.x0 + xl + 1;
      const wx_gptr d0p = (wx_gptr)(uintptr_t)((char*)g.out + (pix * g.Co + ch0) * 2);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        c3_f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = acc[0][i][r] + acc[1][i][r] + acc[2][i][r]; o1[r] = acc[1][i][r] - acc[2][i][r] - acc[3][i][r]; }
        const bool cok = ch0 + i * 32 < g.Co;
        store16(d0p + i * 64, rok && cok && cur.x0 + xl < g.W, o0);
        store16(d0p + (size_t)g.Co * 2 + i * 64, rok && cok && cur.x0 + xl + 1 < g.W, o1);
      }
    }
    if constexpr (POOL) {
      // pooled pixel ((y0 >> 1) + wm, (x0 >> 1) + tq): horizontal max inside the lane, vertical partner = the same pair in the other patch
      // row (rotated lane order: row 0 lane c <-> row 1 lane 16 + ((c + 2) & 15)). max commutes with the bias (in m1), the ReLU and the rounding
      const int Ho = g.H >> 1, Wo = g.W >> 1;
      const int vpart = ((lane & 32) | ((lq & 16) ? ((lq - 2) & 15) : 16 + ((lq + 2) & 15))) << 2;
      const bool second = (lq & 16) != 0;                 // row-0 lanes keep channel tile 0, row-1 lanes tile 1
      // in place: element r of the pooled tile overwrites acc[0][0][r] (its inputs are dead by then); a scheduling fence every four
      // elements keeps hipcc from hoisting all 16 cross-lane moves and their operands at once (the kernel sits at the 256-register line)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float h0 = __builtin_fmaxf(acc[0][0][r] + acc[1][0][r] + acc[2][0][r], acc[1][0][r] - acc[2][0][r] - acc[3][0][r]);
        const float h1 = __builtin_fmaxf(acc[0][1][r] + acc[1][1][r] + acc[2][1][r], acc[1][1][r] - acc[2][1][r] - acc[3][1][r]);
        const float own = second ? h1 : h0, send = second ? h0 : h1;
        float recv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(vpart, __builtin_bit_cast(int, send)));
        asm volatile("" : "+v"(recv));                    // cross-lane results pinned outside the store's exec-masked block
        acc[0][0][r] = __builtin_fmaxf(own, recv);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      const c3_f32x16& mine = acc[0][0];
      const int Y = (cur.y0 >> 1) + wm, X = (cur.x0 >> 1) + tq;
      const long long ppix = ((long long)cur.img * (Ho + 2) + Y + 1) * (Wo + 2) + X + 1;
      const int co = ch0 + (second ? 32 : 0);
      store16((wx_gptr)(uintptr_t)((char*)g.pool_out + (ppix * g.Co + co) * 2), Y < Ho && X < Wo && co < g.Co, mine);
    }
    if (!has_next) break;
    lid = nlid;
    cur = nxt;
  }
  c3_wait_vm<0>();   // the dummy prefetch of the last tile
}

int launch_wino_pack(const float* w_hwio, void* u, int ci, int co, hipStream_t s) {
  if (ci % 64 || co % 128) return fail(CTPN_ERR_ARG, "winograd pack: Ci % 64 == 0 and Co % 128 == 0 required");
  const long long total = (long long)co * 12 * ci;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_hwio, (uint16_t*)u, ci, co);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("winograd pack launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// may this layer take the Winograd kernel? (fp16, 8 x 32 patches: not the flat-window maps, not where 16 x 16 patches of the direct
// kernel cover the map with more than 10 % fewer tiles -- the Winograd tile costs 0.8 - 0.9 of a direct one)
bool wino_layer_ok(int n, int h, int w, int ci, int co, bool pool, bool keep_full, int w_cover) {
  if (ci % 64 || co % 128 || ci < 128) return false;
  Conv3 g{};
  g.N = n; g.H = h; g.W = w; g.Ci = ci; g.Co = co; g.w_cover = w_cover;
  g.out = keep_full ? (void*)1 : nullptr;
  if (!pool) g.out = (void*)1;
  if (c3_flat_ok(g, pool)) return false;
  return 10 * c3_tiles2d(g, pool, 32) <= 11 * c3_tiles2d(g, pool, 16);
}

// in / out / pool_out: bordered NHWC fp16; u: launch_wino_pack's output; w_cover: columns [0, w_cover) are this launch's (0 = all)
int launch_conv3x3_wino(const void* in, const void* u, const float* bias, void* out, void* pool_out, int n, int h, int w, int ci, int co,
                        int w_cover, hipStream_t s) {
  if (!bias || (!out && !pool_out)) return fail(CTPN_ERR_ARG, "conv3x3 winograd: bias and an output required");
  const bool pool = pool_out != nullptr;
  ConvWx g{};
  g.in = in; g.u = u; g.bias = bias; g.out = out; g.pool_out = pool_out;
  g.N = n; g.H = h; g.W = w; g.Ci = ci; g.Co = co;
  int he = (pool && !out) ? (h & ~1) : h, we = (pool && !out) ? (w & ~1) : w;
  if (w_cover > 0 && w_cover < we) we = w_cover;
  g.tiles_x = (we + 31) / 32;
  g.tiles_y = (he + 7) / 8;
  g.tiles_n = co / 128;
  g.ptiles_total = (long long)n * g.tiles_x * g.tiles_y * g.tiles_n;
  if (g.ptiles_total <= 0 || (long long)n * (h + 2) * (w + 2) > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "conv3x3 winograd: problem out of range");
  const int lds = 2 * WX_AROWS * 128 + 3 * WX_STRIP + g.tiles_n * 128 * 4;
  int dev = 0, ncu = 0, rc;
  if ((rc = c3_device(dev)) || (rc = c3_cu_count(dev, ncu))) return rc;
  const long long workers = g.ptiles_total < ncu ? g.ptiles_total : ncu;
  static bool attr[2][C3_MAX_DEV] = {{false}};
  if (pool) {
    auto k = conv3x3_wx_kernel<true>;
    if ((rc = c3_raise_lds((const void*)k, attr[1], dev))) return rc;
    hipLaunchKernelGGL(k, dim3((unsigned)workers), dim3(512), lds, s, g);
  } else {
    auto k = conv3x3_wx_kernel<false>;
    if ((rc = c3_raise_lds((const void*)k, attr[0], dev))) return rc;
    hipLaunchKernelGGL(k, dim3((unsigned)workers), dim3(512), lds, s, g);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("conv3x3 winograd launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
