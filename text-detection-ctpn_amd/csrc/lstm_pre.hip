// LSTM input projection of the 16-bit modes: lstm_pre[m][1024] = rpn_conv[m][512] @ Wx^T + b for every feature-map cell m, fp16 out.
//
// Replaces the x_t @ kernel[:512] + bias half of tf.contrib.rnn.LSTMCell inside Network.Bilstm (reference lib/networks/network.py:97-100),
// hoisted out of the recurrence for all rows and time steps at once (both directions: 2 x 512 gate columns, permuted at weight-pack time
// into the order the recurrent kernels read, bilstm.hip).
//
// Why not igemm.hip (rounds 1-3: 128 x 128 tiles, 4144 short workgroups, 170 us = 16 % of the MFMA peak)? With K = 512 a tile is eight K
// steps between an exposed prologue and a 64 KB epilogue, and M / 128 x N / 128 tiles move every operand eight times through L2 -> LDS. A
// first rewrite (weight slice resident in LDS, activation tiles streamed, prefetch distance 1 -- all the LDS left) measured the same 170 us:
// 16 KB in flight per CU against ~1.5 us of load latency is 3.4 TB/s for the 543 MB of activation re-reads. So the ACTIVATIONS go to
// REGISTERS instead and are read exactly once: a wave owns 32 cells, its [32][512] operand is 32 MFMA fragments = 128 VGPRs per lane,
// loaded straight from the bordered NHWC map; a workgroup (8 waves = 256 cells) then walks ALL 1024 gate columns in 32 tiles of 32
// columns, whose [32][512] weight blocks (32 KB, pre-arranged fragment-major so that the LDS-DMA is a straight copy and every ds_read_b128
// lane group is conflict-free) stream through FOUR LDS buffers, three tiles ahead, shared by the eight waves. Weights are 1 MB in total
// and stay in L2; L2 -> LDS traffic is 259 MB instead of 543, HBM reads the activations once (68 MB). Per column tile a wave runs 32
// MFMAs on one accumulator pair of chains and stores its 32 x 32 fp16 results from registers (v_permlane32_swap -> 16-byte stores).
#include "common.h"

namespace ctpn {

typedef uint32_t lp_u32x4 __attribute__((ext_vector_type(4)));

struct LstmPre {
  const void* a;        // bordered NHWC 16-bit activations: n x (hf + 2) x (wf + 2) x 512
  const void* wt;       // fragment-major weights (lstm_pre_pack_kernel): [32 column tiles][32 k-slices][2 halves][32 columns][8 k] 16-bit
  const float* bias;    // [1024]
  void* out;            // [M][1024] fp16
  long long M;
  int hf, wf;
  int nbuf;             // LDS weight-tile buffers: 4 (prefetch distance 3, one workgroup per CU) or 2 (distance 1, two per CU: small problems)
  int cgroups;          // workgroups per cell range: each walks 32 / cgroups of the 32 column tiles (small problems: more, shorter workgroups)
};

constexpr int LP_TILE_B = 32 * 512 * 2;       // bytes of one column tile's weights
constexpr int LP_NBUF = 4;

// wt_x [1024 gate rows][512 k] (row-major, 16-bit) -> the fragment-major order above: element (tile T, slice q, half h, column c, j) =
// wt_x[32 T + c][16 q + 8 h + j]
__global__ __launch_bounds__(256) void lstm_pre_pack_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;            // one 16-byte unit each: 1024 * 512 / 8 = 65536 units
  if (idx >= 65536) return;
  const int c = idx & 31, h = (idx >> 5) & 1, q = (idx >> 6) & 31, T = idx >> 11;
  *(uint4*)(dst + (size_t)idx * 8) = *(const uint4*)(src + (size_t)(32 * T + c) * 512 + 16 * q + 8 * h);
}

// m0 is on the clobber list on purpose (conv3x3_impl.h, c3_glds16_saddr, says why and what guards it): the one -Winline-asm diagnostic
// of this file is silenced HERE and nowhere else, so that the build can treat every other warning as an error
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lp_glds16(const void* gsrc, uint32_t lds_dst) {      // one 1-KiB LDS-DMA piece (lds_dst wave-uniform; the hardware adds lane * 16)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
#pragma clang diagnostic pop
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate); above 16 it waits for 16: stricter, still correct
__device__ __forceinline__ void lp_wait_vm(int n) {
#define LP_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n > 16 ? 16 : n) {
    LP_VM(0) LP_VM(1) LP_VM(2) LP_VM(3) LP_VM(4) LP_VM(5) LP_VM(6) LP_VM(7) LP_VM(8) LP_VM(9) LP_VM(10) LP_VM(11) LP_VM(12) LP_VM(13) LP_VM(14) LP_VM(15)
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
#undef LP_VM
}

// blockDim = 64 W, W = 1 .. 12 waves of 32 cells each (the launcher picks W so that the grid is ONE round of workgroups where it can:
// 2072 wave-groups of a 32-image batch on 256 CUs are 231 workgroups of 9 waves, not 259 of 8)
template <typename H>
__global__ __launch_bounds__(768) void lstm_pre_kernel(LstmPre g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // LP_NBUF column tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = (int)(blockDim.x >> 6);
  const int l31 = lane & 31, fhalf = lane >> 5;
  const int Wp = g.wf + 2, Hp = g.hf + 2;

  // weight tiles: a straight 32 KB copy per tile, its 32 one-KB LDS-DMA pieces dealt round-robin over the waves. Issued from inline asm:
  // hipcc then neither counts them nor drains them (with the builtin, every __syncthreads() of the tile loop carried a vmcnt(0): the three
  // tiles of prefetch distance collapsed into "wait for the tile requested a microsecond ago" at every barrier -- waitcnt + barrier were
  // 66 % of the wave cycles, profiles/r04_pmc.json). The waits are the kernel's own, counted (lp_wait_vm below).
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto issue_w = [&](int T, int buf) {
    const char* src = (const char*)g.wt + (size_t)T * LP_TILE_B + lane * 16;
    for (int p = wave; p < 32; p += nwaves) lp_glds16(src + p * 1024, __builtin_amdgcn_readfirstlane(lds0 + buf * LP_TILE_B + p * 1024));
  };
  const int pieces_min = 32 / nwaves;                              // every wave issues at least this many pieces per tile
  // this workgroup's cell range and column tiles [T0, T0 + NT)
  const int cg = (int)(blockIdx.x % (unsigned)g.cgroups), cellblk = (int)(blockIdx.x / (unsigned)g.cgroups);
  const int NT = 32 / g.cgroups, T0 = cg * NT;
  const int nbuf = g.nbuf, dist = nbuf - 1;
  for (int i = 0; i < dist && i < NT; ++i) issue_w(T0 + i, i);
  float* const sbias = (float*)(smem + g.nbuf * LP_TILE_B);       // the whole bias vector: a global load at the top of every column tile
  for (int i = tid; i < 1024; i += (int)blockDim.x) sbias[i] = g.bias[i];      // would put ~1 us of latency in front of its first MFMA

  // this lane's cell and its 32 activation fragments: k = 16 q + 8 fhalf .. + 7 of cell m (MFMA B operand: column = cell)
  long long m = ((long long)cellblk * nwaves + wave) * 32 + l31;
  if (m > g.M - 1) m = g.M - 1;
  const long long hw = (long long)g.hf * g.wf;
  const long long n = m / hw;
  const int rem = (int)(m - n * hw);
  const int y = rem / g.wf, x = rem - y * g.wf;
  const char* ap = (const char*)g.a + (((n * Hp + y + 1) * Wp + x + 1) * 512) * 2 + fhalf * 16;
  uint4 xf[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) xf[q] = *(const uint4*)(ap + q * 32);
  // output rows of the two cells this lane STORES for (see the epilogue): cells (lane & 15) and 16 + (lane & 15) of the wave
  const long long mA = ((long long)cellblk * nwaves + wave) * 32 + (lane & 15), mB = mA + 16;
  const bool okA = mA < g.M, okB = mB < g.M;
  char* const obA = (char*)g.out + mA * 2048;
  char* const obB = (char*)g.out + mB * 2048;
  const int woff = fhalf * 512 + l31 * 16;                         // this lane's 16 bytes inside a (tile, k-slice) block of 1 KB

  lp_wait_vm(0);                                                   // tiles 0 .. dist - 1 have landed (and the fragments above)
  // ... and this wave's sbias stores have reached LDS before another wave reads them behind the barrier: gfx950 has back-off barriers, the
  // compiler no longer puts a wait in front of s_barrier by itself, and the builtin is no memory operation to it (the loop's barrier
  // carries the same explicit wait)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");                                   // no LDS read of the loop is hoisted above the barrier
#pragma unroll 1
  for (int Tl = 0; Tl < NT; ++Tl) {
    const int T = T0 + Tl;
    if (Tl + dist < NT) issue_w(T + dist, (Tl + dist) & (nbuf - 1)); // `dist` tiles ahead: its buffer was last read in tile T - 1, behind that tile's barrier
    // bias as the accumulator's initial value; ONE chain: the three waves of a SIMD interleave, which covers the dependent-MFMA latency,
    // and a second chain's 16 registers were exactly what spilled
    ctpn_f32x16 acc0;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b4 = *(const float4*)(sbias + 32 * T + 8 * g4 + 4 * fhalf);
      acc0[4 * g4] = b4.x; acc0[4 * g4 + 1] = b4.y; acc0[4 * g4 + 2] = b4.z; acc0[4 * g4 + 3] = b4.w;
    }
    const char* sw = smem + (Tl & (nbuf - 1)) * LP_TILE_B + woff;
#pragma unroll
    for (int q = 0; q < 32; ++q) acc0 = HalfOps<H>::mfma_32x32x16(*(const uint4*)(sw + q * 1024), xf[q], acc0);
    // gate columns 32 T + 8 g4 + 4 fhalf + e of this lane's cell -> fp16. v_permlane32_swap completes 16-byte pieces (8 columns),
    // v_permlane16_swap then trades piece 1 of lanes r with piece 0 of lanes r + 16: store A carries cells 0..15 of the wave, store B cells
    // 16..31, FOUR lanes = 64 contiguous bytes per cell -- a store instruction touches 16 lines instead of 32 (conv3x3's store_pair)
    {
      lp_u32x4 v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint32_t e0 = ctpn_cvt_pk_f16(acc0[8 * q + 0], acc0[8 * q + 1]), e1 = ctpn_cvt_pk_f16(acc0[8 * q + 2], acc0[8 * q + 3]);
        const uint32_t o0 = ctpn_cvt_pk_f16(acc0[8 * q + 4], acc0[8 * q + 5]), o1 = ctpn_cvt_pk_f16(acc0[8 * q + 6], acc0[8 * q + 7]);
        const auto r0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);   // low lanes: channel group 2q complete, high lanes: 2q + 1
        const auto r1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
        v[q] = lp_u32x4{r0[0], r1[0], r0[1], r1[1]};                               // piece 2 q + fhalf of cell l31
      }
      lp_u32x4 va, vb;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const auto r = __builtin_amdgcn_permlane16_swap(v[0][c], v[1][c], false, false);
        va[c] = r[0]; vb[c] = r[1];
      }
      // lane L now holds piece 2 * ((L >> 4) & 1) + (L >> 5) of cell (L & 15) [va] and of cell 16 + (L & 15) [vb]
      const int piece = 2 * ((lane >> 4) & 1) + fhalf;
      if (okA) *(lp_u32x4*)(obA + (32 * T + 8 * piece) * 2) = va;
      if (okB) *(lp_u32x4*)(obB + (32 * T + 8 * piece) * 2) = vb;
    }
    // Tile T + 1 has to have landed; tiles T + 2 .. stay in flight across the barrier. Vector-memory operations retire in order per kind
    // (the DMA pieces among themselves, the stores among themselves), so "at most as many outstanding as DMA pieces this wave issued
    // AFTER tile T + 1's" retires that tile whatever the stores do: pieces_min per tile still to come (a wave with one piece more waits
    // for it as well: stricter, never looser).
    {
      int later = NT - 2 - Tl;                                     // tiles behind T + 1 that have been requested
      later = later < 0 ? 0 : (later > dist - 1 ? dist - 1 : later);
      lp_wait_vm(later * pieces_min);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // every wave is done with tile T's buffer and has seen its share of tile T + 1 land
  }
}

// Few cells (one or two 600 x 900 images: 65 / 130 wave-groups): the resident-activation structure above would be a handful of workgroups
// walking 32 column tiles each behind a barrier per tile (65 us for one image even when the tiles are split over workgroups). Here ONE
// WAVE owns (32 cells) x (LPS_NT = 4 column tiles): the 32 activation fragments in registers as above, the weight fragments straight from
// L2 into registers (the fragment-major layout is one coalesced 1-KB read per (tile, k-slice)), four independent accumulator chains, no
// LDS, no barrier: 65 x 8 = 520 one-wave workgroups for one image, all resident at once. Every output is the same chain as in
// lstm_pre_kernel -- bias as the initial value, k-slices 0 .. 31 in order -- so a batch and its images run alone agree bit for bit.
constexpr int LPS_NT = 4;
template <typename H>
__global__ __launch_bounds__(64) void lstm_pre_small_kernel(LstmPre g) {
  const int lane = threadIdx.x, l31 = lane & 31, fhalf = lane >> 5;
  const int Wp = g.wf + 2, Hp = g.hf + 2;
  const int tq = (int)(blockIdx.x % (unsigned)(32 / LPS_NT)), grp = (int)(blockIdx.x / (unsigned)(32 / LPS_NT));
  const int T0 = tq * LPS_NT;
  long long m = (long long)grp * 32 + l31;
  if (m > g.M - 1) m = g.M - 1;
  const long long hw = (long long)g.hf * g.wf;
  const long long n = m / hw;
  const int rem = (int)(m - n * hw);
  const int y = rem / g.wf, x = rem - y * g.wf;
  const char* ap = (const char*)g.a + (((n * Hp + y + 1) * Wp + x + 1) * 512) * 2 + fhalf * 16;
  uint4 xf[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) xf[q] = *(const uint4*)(ap + q * 32);
  ctpn_f32x16 acc[LPS_NT];
#pragma unroll
  for (int t = 0; t < LPS_NT; ++t)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b4 = *(const float4*)(g.bias + 32 * (T0 + t) + 8 * g4 + 4 * fhalf);
      acc[t][4 * g4] = b4.x; acc[t][4 * g4 + 1] = b4.y; acc[t][4 * g4 + 2] = b4.z; acc[t][4 * g4 + 3] = b4.w;
    }
  const char* wp = (const char*)g.wt + (size_t)T0 * LP_TILE_B + fhalf * 512 + l31 * 16;
#pragma unroll
  for (int q = 0; q < 32; ++q)
#pragma unroll
    for (int t = 0; t < LPS_NT; ++t) acc[t] = HalfOps<H>::mfma_32x32x16(*(const uint4*)(wp + (size_t)t * LP_TILE_B + q * 1024), xf[q], acc[t]);
  const long long mA = (long long)grp * 32 + (lane & 15), mB = mA + 16;
  const bool okA = mA < g.M, okB = mB < g.M;
  char* const obA = (char*)g.out + mA * 2048;
  char* const obB = (char*)g.out + mB * 2048;
#pragma unroll
  for (int t = 0; t < LPS_NT; ++t) {      // the store form of lstm_pre_kernel
    lp_u32x4 v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t e0 = ctpn_cvt_pk_f16(acc[t][8 * q + 0], acc[t][8 * q + 1]), e1 = ctpn_cvt_pk_f16(acc[t][8 * q + 2], acc[t][8 * q + 3]);
      const uint32_t o0 = ctpn_cvt_pk_f16(acc[t][8 * q + 4], acc[t][8 * q + 5]), o1 = ctpn_cvt_pk_f16(acc[t][8 * q + 6], acc[t][8 * q + 7]);
      const auto r0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
      v[q] = lp_u32x4{r0[0], r1[0], r0[1], r1[1]};
    }
    lp_u32x4 va, vb;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const auto r = __builtin_amdgcn_permlane16_swap(v[0][c], v[1][c], false, false);
      va[c] = r[0]; vb[c] = r[1];
    }
    const int piece = 2 * ((lane >> 4) & 1) + fhalf;
    if (okA) *(lp_u32x4*)(obA + (32 * (T0 + t) + 8 * piece) * 2) = va;
    if (okB) *(lp_u32x4*)(obB + (32 * (T0 + t) + 8 * piece) * 2) = vb;
  }
}

// dst: device buffer of 1 MB (16-bit); src: wt_x [1024][512] 16-bit (gate rows already permuted)
int launch_lstm_pre_pack(const void* src, void* dst, hipStream_t s) {
  hipLaunchKernelGGL(lstm_pre_pack_kernel, dim3(256), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("lstm_pre pack launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// a: bordered NHWC 16-bit map of n x hf x wf cells x 512 channels (dtype t: BF16 or F16); wt: launch_lstm_pre_pack's output;
// out: [n * hf * wf][1024] fp16
int launch_lstm_pre(const void* a, const void* wt, const float* bias, void* out, DType t, int n, int hf, int wf, hipStream_t s) {
  if (!dtype_is_half(t)) return fail(CTPN_ERR_ARG, "lstm_pre: 16-bit modes only");
  LstmPre g{};
  g.a = a; g.wt = wt; g.bias = bias; g.out = out; g.M = (long long)n * hf * wf; g.hf = hf; g.wf = wf;
  if (g.M <= 0 || g.M > 0x7fffffffLL) return fail(CTPN_ERR_ARG, "lstm_pre: problem out of range");

  int dev = 0, ncu = 0, rc;
  if ((rc = current_device(dev)) || (rc = device_cu_count(dev, ncu))) return rc;
  const long long groups = (g.M + 31) / 32;                        // wave-groups of 32 cells
  long long W = (groups + ncu - 1) / ncu;                          // waves per workgroup: one round of workgroups if 12 waves suffice
  W = W < 1 ? 1 : (W > 12 ? 12 : W);
  // few cells (one image: 65 wave-groups): split the 32 column tiles of a cell range over 2 .. 8 workgroups so that the launch still fills the chip
  const long long cellblks = (groups + W - 1) / W;
  g.cgroups = 1; g.nbuf = LP_NBUF;
  if (cellblks * 2 <= ncu) {                                       // less than half a round of workgroups: one wave per (32 cells, four column tiles)
    if (t == DType::F16) hipLaunchKernelGGL((lstm_pre_small_kernel<h_f16>), dim3((unsigned)(groups * (32 / LPS_NT))), dim3(64), 0, s, g);
    else hipLaunchKernelGGL((lstm_pre_small_kernel<h_bf16>), dim3((unsigned)(groups * (32 / LPS_NT))), dim3(64), 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("lstm_pre (small) launch: ") + hipGetErrorString(e));
    return CTPN_OK;
  }
  const int lds = g.nbuf * LP_TILE_B + 1024 * 4;
  auto launch = [&](auto kern) -> int {
    static bool done[CTPN_MAX_DEV] = {false};
    const int r = raise_dynamic_lds((const void*)kern, LP_NBUF * LP_TILE_B + 1024 * 4, done, dev);
    if (r) return r;
    hipLaunchKernelGGL(kern, dim3((unsigned)(cellblks * g.cgroups)), dim3(64 * W), lds, s, g);
    return CTPN_OK;
  };
  rc = t == DType::F16 ? launch(lstm_pre_kernel<h_f16>) : launch(lstm_pre_kernel<h_bf16>);
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("lstm_pre launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
