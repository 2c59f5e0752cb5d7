// Persistent BiLSTM recurrence over feature-map rows.
//
// Replaces the tf.nn.bidirectional_dynamic_rnn of two tf.contrib.rnn.LSTMCell(128) in Network.Bilstm
// (reference lib/networks/network.py:88-101; TF-1.3 LSTMCell semantics):
//     z = [x_t, h_{t-1}] @ kernel + bias ;  i, j, f, o = split(z, 4)
//     c_t = sigmoid(f + 1.0) * c_{t-1} + sigmoid(i) * tanh(j) ;  h_t = sigmoid(o) * tanh(c_t)
// The x_t @ kernel[:512] + bias part is hoisted into one MFMA GEMM over all rows and steps (igemm.hip,
// "lstm_pre"); this kernel does the sequential part.
//
// One workgroup = 16 feature-map rows (independent sequences) of one direction, resident for all T steps:
//   * 8 waves; wave w owns hidden units [16w, 16w+16) and holds the matching 4 x (128 x 16) slices of
//     Wh = kernel[512:640] in 128 VGPRs per lane for the whole kernel (256 KB per direction across the
//     workgroup's register file -- never re-read from HBM or LDS);
//   * per step, per gate: D[unit][row] = xp[row][t][gate,unit] + sum_k Wh[k][gate,unit] * h[row][k] as 32
//     v_mfma_f32_16x16x4_f32 (exact fp32) with the pre-activation as the C input; the four gates are four
//     independent accumulator chains, so the 40-cycle dependent latency is hidden;
//   * the 16x16 C/D layout leaves each lane with i, j, f, o of the SAME 4 units of one row: the cell update is
//     entirely in registers, c never leaves the lane;
//   * h_t goes to a double-buffered LDS tile (pitch 544 B: conflict-free ds_read_b128 fragment reads) and to HBM
//     as 16-byte stores; one barrier per step;
//   * the next step's pre-activations (4 x 16 B per lane, row strips of lstm_pre) are requested before the MFMAs. The gate columns
//     of lstm_pre are PERMUTED at weight-pack time (lstm_gate_col below) so that a lane's 4 gates x 4 units are one 64-byte run
//     and a wave reads 256 contiguous bytes per row: in TF's i | j | f | o order the four strips were separate 64-byte runs,
//     fetched as 128-byte lines -- 1.69 x the algorithmic HBM traffic (344 MB instead of 204 MB per 32-image batch).
#include "common.h"

namespace ctpn {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f / (expf(2.f * x) + 1.f); }
// v_exp_f32 / v_rcp_f32 forms (1 ulp each): the gates sit on the step's critical path, expf()'s range reduction and the IEEE
// division do not pay there (bf16 throughput mode; the fp32 gate keeps the library forms)
// (fp contraction OFF in the gate math of the fast forms: bilstm_split_kernel and bilstm_split_few_kernel must give the same bits for the same
// row, and whether `a * b + c` becomes one v_fma or a v_mul and a v_add is otherwise the compiler's choice per call site)
__device__ __forceinline__ float fast_sigmoid(float x) {
#pragma clang fp contract(off)
  return __frcp_rn(1.f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
#pragma clang fp contract(off)
  const float r = __frcp_rn(__expf(2.f * x) + 1.f);
  const float t = 2.f * r;
  return 1.f - t;
}
// one LSTMCell update from the four gate pre-activations (TF order i, j, f, o; forget bias 1.0) in the v_exp / v_rcp forms: c is updated, h returned
__device__ __forceinline__ float lstm_cell_fast(float zi, float zj, float zf, float zo, float& c) {
#pragma clang fp contract(off)
  const float ig = fast_sigmoid(zi);
  const float jg = fast_tanh(zj);
  const float fg = fast_sigmoid(zf + 1.0f);
  const float og = fast_sigmoid(zo);
  const float a = fg * c;
  const float b = ig * jg;
  const float cn = a + b;
  c = cn;
  return og * fast_tanh(cn);
}

constexpr int LSTM_ROWS = 16;
constexpr int LSTM_HPITCH = 136;  // floats per h row in LDS (128 + 8 pad = 544 B)

// A lane's pre-activations of one position: 4 gates x 4 units, contiguous in the permuted lstm_pre layout -- 64 bytes of fp32, or 32
// bytes of fp16 (PRE16: the 16-bit throughput modes store lstm_pre as fp16, which halves the 272 MB the projection GEMM writes and
// this kernel reads back per batch; the rounding, 2^-11 relative, is the class of every 16-bit activation in front of it)
// The loaded bits stay RAW in registers until the step that consumes them: converting right behind the load would put the load's latency
// (it is issued one step ahead precisely to hide it) in front of the next instruction.
template <bool PRE16> struct LstmPreRaw { f32x4 v[4]; };
template <> struct LstmPreRaw<true> { uint4 v[2]; };
template <bool PRE16>
__device__ __forceinline__ void lstm_load_pre(const void* lane_base, size_t pos, LstmPreRaw<PRE16>& raw) {
  if constexpr (PRE16) {
    const uint4* p = (const uint4*)((const _Float16*)lane_base + pos * 1024);
    raw.v[0] = p[0]; raw.v[1] = p[1];
  } else {
    const f32x4* p = (const f32x4*)((const float*)lane_base + pos * 1024);
#pragma unroll
    for (int g = 0; g < 4; ++g) raw.v[g] = p[g];
  }
}
template <bool PRE16>
__device__ __forceinline__ void lstm_pre_values(const LstmPreRaw<PRE16>& raw, f32x4 (&pre)[4]) {
  if constexpr (PRE16) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const f16x8 a = __builtin_bit_cast(f16x8, raw.v[0]), b = __builtin_bit_cast(f16x8, raw.v[1]);
#pragma unroll
    for (int e = 0; e < 4; ++e) { pre[0][e] = (float)a[e]; pre[1][e] = (float)a[4 + e]; pre[2][e] = (float)b[e]; pre[3][e] = (float)b[4 + e]; }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] = raw.v[g];
  }
}
template <bool FAST, bool PRE16>
__global__ __launch_bounds__(512) void bilstm_kernel(const void* __restrict__ xp, const float* __restrict__ wh,
                                                     float* __restrict__ out, int rows, int T) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][LSTM_ROWS][LSTM_HPITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.y;
  const int r = lane & 15, q4 = lane >> 4;
  const int ucol = 16 * wave + r;           // A operand: this lane's unit (MFMA row index i = lane&15)
  const float* whd = wh + (size_t)dir * 128 * 512;

  // Wh slices -> registers. areg[g][qq*4+e] = Wh[k = 16qq + 4q4 + e][g*128 + ucol]
  float areg[4][32];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int qq = 0; qq < 8; ++qq)
#pragma unroll
      for (int e = 0; e < 4; ++e) areg[g][qq * 4 + e] = whd[(size_t)(16 * qq + 4 * q4 + e) * 512 + g * 128 + ucol];

  for (int i = tid; i < 2 * LSTM_ROWS * LSTM_HPITCH; i += 512) (&hbuf[0][0][0])[i] = 0.f;

  // this lane's output slot: row (lane&15) of the block, units 16*wave + 4*q4 + {0..3}
  const int row_l = r;
  const int row_g = blockIdx.x * LSTM_ROWS + row_l;
  const bool row_ok = row_g < rows;
  const int row_c = row_ok ? row_g : rows - 1;
  const int u0 = 16 * wave + 4 * q4;
  const size_t xoff = (size_t)row_c * T * 1024 + dir * 512 + 64 * wave + 16 * q4;           // permuted gate columns: [wave][q4][gate][4 units]
  const void* xrow = PRE16 ? (const void*)((const _Float16*)xp + xoff) : (const void*)((const float*)xp + xoff);
  float* orow = out + (size_t)row_c * T * 256 + dir * 128 + u0;

  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  LstmPreRaw<PRE16> praw;
  {
    const int t0 = dir ? T - 1 : 0;
    lstm_load_pre<PRE16>(xrow, (size_t)t0, praw);
  }
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    f32x4 acc[4];
    lstm_pre_values<PRE16>(praw, acc);
    if (s + 1 < T) {
      const int tn = dir ? t - 1 : t + 1;
      lstm_load_pre<PRE16>(xrow, (size_t)tn, praw);
    }
    // h_{t-1} fragments: lane reads h[row = lane&15][k = 16qq + 4q4 .. +3]
    f32x4 hf[8];
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) hf[qq] = *(const f32x4*)(&hbuf[cur][r][16 * qq + 4 * q4]);
#pragma unroll
    for (int qq = 0; qq < 8; ++qq)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[g][qq * 4 + e], hf[qq][e], acc[g], 0, 0, 0);
    // cell update: acc[g][e] is gate g of unit u0+e for row (lane&15)
    f32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ig = FAST ? fast_sigmoid(acc[0][e]) : sigmoidf_(acc[0][e]);
      const float jg = FAST ? fast_tanh(acc[1][e]) : tanhf_(acc[1][e]);
      const float fg = FAST ? fast_sigmoid(acc[2][e] + 1.0f) : sigmoidf_(acc[2][e] + 1.0f);
      const float og = FAST ? fast_sigmoid(acc[3][e]) : sigmoidf_(acc[3][e]);
      const float cn = fg * c[e] + ig * jg;
      c[e] = cn;
      h[e] = og * (FAST ? fast_tanh(cn) : tanhf_(cn));
    }
    *(f32x4*)(&hbuf[cur ^ 1][row_l][u0]) = h;
    if (row_ok) *(f32x4*)(orow + (size_t)t * 256) = h;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 form of the same recurrence for the bf16 throughput mode: the fp32 MFMA (157 TF) bounds the kernel above at
// 8192 MFMA cycles per SIMD per step (4.3 of the measured 6.2 us). Writing every fp32 operand as hi + lo bf16 and taking
//     W h  ~=  Wlo hhi + Whi hlo + Whi hhi      (fp32 accumulate; the dropped lo*lo term is ~2^-16 of a product)
// turns 128 v_mfma_f32_16x16x4_f32 (32 cycles each) per wave and step into 48 v_mfma_f32_16x16x32_bf16 (16 cycles each):
// 5.3x fewer MFMA cycles at fp32-class accuracy (tests: |lstm_out - fp32 kernel| < 2e-5). State, gates, pre-activations
// and the accumulation stay fp32; Wh hi/lo fragments take the same 128 VGPRs the fp32 slices did; h_t goes to LDS as two
// bf16 planes (pitch 288 B: every ds_read_b128 lane group lands on 16 distinct bank quads).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 lstm_bf16x8;
constexpr int LSTM_BPITCH = 144;   // bf16 per h row in LDS (128 + 16 pad = 288 B)

__device__ __forceinline__ void lstm_split(float v, uint32_t& hi, uint32_t& lo) {   // bf16 bits of v = hi + lo
  hi = ctpn_cvt_pk_bf16(v, 0.f) & 0xffffu;
  lo = ctpn_cvt_pk_bf16(v - __builtin_bit_cast(float, hi << 16), 0.f) & 0xffffu;
}

template <bool PRE16>
__global__ __launch_bounds__(512) void bilstm_split_kernel(const void* __restrict__ xp, const float* __restrict__ wh,
                                                           float* __restrict__ out, int rows, int T) {
  __shared__ __attribute__((aligned(16))) uint16_t hb[2][2][LSTM_ROWS][LSTM_BPITCH];   // [buffer][hi|lo][row][k]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.y;
  const int r = lane & 15, q4 = lane >> 4;
  const int ucol = 16 * wave + r;           // A operand row: this lane's unit
  const float* whd = wh + (size_t)dir * 128 * 512;

  // Wh fragments: wa[g][kk][part] = 8 bf16 of Wh[k = 32 kk + 8 q4 + j][g * 128 + ucol], j = 0..7
  uint4 wa[4][4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint32_t hi[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) lstm_split(whd[(size_t)(32 * kk + 8 * q4 + j) * 512 + g * 128 + ucol], hi[j], lo[j]);
      wa[g][kk][0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
      wa[g][kk][1] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    }
  for (int i = tid; i < 2 * 2 * LSTM_ROWS * LSTM_BPITCH / 2; i += 512) ((uint32_t*)&hb[0][0][0][0])[i] = 0u;

  const int row_l = r;
  const int row_g = blockIdx.x * LSTM_ROWS + row_l;
  const bool row_ok = row_g < rows;
  const int row_c = row_ok ? row_g : rows - 1;
  const int u0 = 16 * wave + 4 * q4;        // D rows: units u0 .. u0 + 3 of batch row (lane & 15)
  const size_t xoff = (size_t)row_c * T * 1024 + dir * 512 + 64 * wave + 16 * q4;           // permuted gate columns: [wave][q4][gate][4 units]
  const void* xrow = PRE16 ? (const void*)((const _Float16*)xp + xoff) : (const void*)((const float*)xp + xoff);
  float* orow = out + (size_t)row_c * T * 256 + dir * 128 + u0;

  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  LstmPreRaw<PRE16> praw;
  {
    const int t0 = dir ? T - 1 : 0;
    lstm_load_pre<PRE16>(xrow, (size_t)t0, praw);
  }
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    f32x4 acc[4];
    lstm_pre_values<PRE16>(praw, acc);
    if (s + 1 < T) {
      const int tn = dir ? t - 1 : t + 1;
      lstm_load_pre<PRE16>(xrow, (size_t)tn, praw);
    }
    // h_{t-1} fragments (B operand): lane reads h[row = lane & 15][k = 32 kk + 8 q4 .. + 7], hi and lo planes
    uint4 hh[4], hl[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      hh[kk] = *(const uint4*)(&hb[cur][0][r][32 * kk + 8 * q4]);
      hl[kk] = *(const uint4*)(&hb[cur][1][r][32 * kk + 8 * q4]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int g = 0; g < 4; ++g)      // four independent accumulator chains
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lstm_bf16x8, wa[g][kk][term == 0 ? 1 : 0]),
                                                           __builtin_bit_cast(lstm_bf16x8, term == 1 ? hl[kk] : hh[kk]), acc[g], 0, 0, 0);
    f32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // v_exp_f32 / v_rcp_f32 forms (1 ulp each): the gates sit on the step's critical path, expf()'s range reduction does not pay
      float ce = c[e];
      h[e] = lstm_cell_fast(acc[0][e], acc[1][e], acc[2][e], acc[3][e], ce);
      c[e] = ce;
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) lstm_split(h[e], hi[e], lo[e]);
    *(uint2*)(&hb[cur ^ 1][0][row_l][u0]) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    *(uint2*)(&hb[cur ^ 1][1][row_l][u0]) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
    if (row_ok) *(f32x4*)(orow + (size_t)t * 256) = h;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The same recurrence for a FEW rows (round 6: one or two images per call -- the reference's own calling convention, ctpn/demo.py:55-68).
// A step of bilstm_split_kernel is 96 MFMAs per SIMD (~1540 clk) FOLLOWED by the gate math of 16 rows x 128 units -- 40 transcendental
// instructions per lane, ~2500 clk for the two waves of a SIMD: the barrier keeps the waves in step, so the two phases do not overlap and the
// gates are the larger one (2.1 us per step, 120 us for the 57 steps of a 600 x 900 image on 6 of 256 CUs). The MFMA phase does not shrink
// with fewer rows (the 16 x 16 tile's columns are the rows), the gate phase does: here a workgroup takes FOUR rows, so only lanes
// (lane & 15) < 4 hold valid sums after the MFMAs -- 4 units x 4 gates each -- and one DPP row shift per gate hands units 1 .. 3 to the
// twelve idle lanes of the 16-lane row: every lane then runs the cell update of ONE (row, unit): 10 transcendentals instead of 40. Same
// MFMA sequence, same gate formulas, same split of h: bit-identical to bilstm_split_kernel (tests/test_gpu_round6.py). 37 rows are 10
// workgroups per direction instead of 3.
// ---------------------------------------------------------------------------------------------
constexpr int LSTM_FEW = 4;
template <int SHR>
__device__ __forceinline__ float lstm_row_shr(float v) {      // lane i of a 16-lane row <- lane i - SHR (v_mov_b32_dpp row_shr)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x110 + SHR, 0xF, 0xF, true));
}
template <bool PRE16>
__global__ __launch_bounds__(512) void bilstm_split_few_kernel(const void* __restrict__ xp, const float* __restrict__ wh,
                                                               float* __restrict__ out, int rows, int T) {
  __shared__ __attribute__((aligned(16))) uint16_t hb[2][2][LSTM_ROWS][LSTM_BPITCH];   // [buffer][hi|lo][row][k]; rows >= LSTM_FEW stay zero
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.y;
  const int r = lane & 15, q4 = lane >> 4;
  const int ucol = 16 * wave + r;
  const float* whd = wh + (size_t)dir * 128 * 512;
  uint4 wa[4][4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint32_t hi[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) lstm_split(whd[(size_t)(32 * kk + 8 * q4 + j) * 512 + g * 128 + ucol], hi[j], lo[j]);
      wa[g][kk][0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
      wa[g][kk][1] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    }
  for (int i = tid; i < 2 * 2 * LSTM_ROWS * LSTM_BPITCH / 2; i += 512) ((uint32_t*)&hb[0][0][0][0])[i] = 0u;

  // MFMA column r < LSTM_FEW: the lane that LOADS the pre-activations of (row r, units u0 .. u0 + 3); after the redistribution lane r'
  // owns (row r' & 3, unit u0 + (r' >> 2))
  const bool ld = r < LSTM_FEW;
  const int row_ld = blockIdx.x * LSTM_FEW + (r & (LSTM_FEW - 1));
  const int row_c = row_ld < rows ? row_ld : rows - 1;
  const int u0 = 16 * wave + 4 * q4;
  const size_t xoff = (size_t)row_c * T * 1024 + dir * 512 + 64 * wave + 16 * q4;
  const void* xrow = PRE16 ? (const void*)((const _Float16*)xp + xoff) : (const void*)((const float*)xp + xoff);
  const int esel = r >> 2;                               // which of the source lane's four units this lane takes
  const int row_l = r & 3, unit = u0 + esel;
  const bool row_ok = blockIdx.x * LSTM_FEW + row_l < rows;
  float* orow = out + (size_t)(row_ok ? blockIdx.x * LSTM_FEW + row_l : rows - 1) * T * 256 + dir * 128 + unit;

  float c = 0.f;
  LstmPreRaw<PRE16> praw;
#pragma unroll
  for (int k = 0; k < (PRE16 ? 2 : 4); ++k) praw.v[k] = {};
  if (ld) lstm_load_pre<PRE16>(xrow, (size_t)(dir ? T - 1 : 0), praw);
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    f32x4 acc[4];
    lstm_pre_values<PRE16>(praw, acc);
    if (s + 1 < T && ld) lstm_load_pre<PRE16>(xrow, (size_t)(dir ? t - 1 : t + 1), praw);
    uint4 hh[4], hl[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      hh[kk] = *(const uint4*)(&hb[cur][0][r][32 * kk + 8 * q4]);
      hl[kk] = *(const uint4*)(&hb[cur][1][r][32 * kk + 8 * q4]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lstm_bf16x8, wa[g][kk][term == 0 ? 1 : 0]),
                                                           __builtin_bit_cast(lstm_bf16x8, term == 1 ? hl[kk] : hh[kk]), acc[g], 0, 0, 0);
    // gate g of (row r & 3, unit u0 + esel): element esel of lane (r & 3)'s accumulator, i.e. of the lane 4 esel to the left in this row
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float a1 = lstm_row_shr<4>(acc[g][1]), a2 = lstm_row_shr<8>(acc[g][2]), a3 = lstm_row_shr<12>(acc[g][3]);
      z[g] = esel == 0 ? acc[g][0] : (esel == 1 ? a1 : (esel == 2 ? a2 : a3));
    }
    const float h = lstm_cell_fast(z[0], z[1], z[2], z[3], c);
    uint32_t hi, lo;
    lstm_split(h, hi, lo);
    hb[cur ^ 1][0][row_l][unit] = (uint16_t)hi;
    hb[cur ^ 1][1][row_l][unit] = (uint16_t)lo;
    if (row_ok) orow[(size_t)t * 256] = h;
    __syncthreads();
  }
}

// TF gate column c = g * 128 + u of one direction (LSTMCell kernel columns: i | j | f | o) -> column of the permuted lstm_pre layout
int lstm_gate_col(int c) {
  const int g = c >> 7, u = c & 127;
  return (u >> 4) * 64 + ((u >> 2) & 3) * 16 + g * 4 + (u & 3);
}

// dst row / element lstm_gate_col(c) of every 512-block <- src row / element c (weight rows [1024][row_bytes], bias [1024] floats)
__global__ __launch_bounds__(256) void lstm_permute_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, int row_bytes) {
  const int c = blockIdx.x, blk = c >> 9, cc = c & 511;
  const int g = cc >> 7, u = cc & 127;
  const int p = (blk << 9) + (u >> 4) * 64 + ((u >> 2) & 3) * 16 + g * 4 + (u & 3);
  for (int i = threadIdx.x * 4; i < row_bytes; i += 256 * 4) *(uint32_t*)(dst + (size_t)p * row_bytes + i) = *(const uint32_t*)(src + (size_t)c * row_bytes + i);
}
int launch_lstm_permute_rows(const void* src, void* dst, int row_bytes, hipStream_t s) {
  if (row_bytes <= 0 || row_bytes % 4) return fail(CTPN_ERR_ARG, "lstm_permute_rows: row_bytes must be a positive multiple of 4");
  hipLaunchKernelGGL(lstm_permute_rows_kernel, dim3(1024), dim3(256), 0, s, (const char*)src, (char*)dst, row_bytes);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("lstm_permute_rows launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

int launch_bilstm(const void* xp, int xp_is_f16, const float* wh, float* out, int rows, int T, hipStream_t s, int split_bf16, int fast_gates) {
  if (rows <= 0 || T <= 0) return fail(CTPN_ERR_ARG, "bilstm: empty problem");
  dim3 grid((rows + LSTM_ROWS - 1) / LSTM_ROWS, 2);
  // a few rows (one or two 600 x 900 images: 37 / 74): four rows per workgroup, the gate math spread over all lanes; identical bits
  if (split_bf16 && rows <= 128) {
    dim3 gf((rows + LSTM_FEW - 1) / LSTM_FEW, 2);
    if (xp_is_f16) hipLaunchKernelGGL(bilstm_split_few_kernel<true>, gf, dim3(512), 0, s, xp, wh, out, rows, T);
    else hipLaunchKernelGGL(bilstm_split_few_kernel<false>, gf, dim3(512), 0, s, xp, wh, out, rows, T);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("bilstm launch: ") + hipGetErrorString(e));
    return CTPN_OK;
  }
  if (xp_is_f16) {
    if (split_bf16) hipLaunchKernelGGL(bilstm_split_kernel<true>, grid, dim3(512), 0, s, xp, wh, out, rows, T);
    else if (fast_gates) hipLaunchKernelGGL((bilstm_kernel<true, true>), grid, dim3(512), 0, s, xp, wh, out, rows, T);
    else hipLaunchKernelGGL((bilstm_kernel<false, true>), grid, dim3(512), 0, s, xp, wh, out, rows, T);
  } else {
    if (split_bf16) hipLaunchKernelGGL(bilstm_split_kernel<false>, grid, dim3(512), 0, s, xp, wh, out, rows, T);
    else if (fast_gates) hipLaunchKernelGGL((bilstm_kernel<true, false>), grid, dim3(512), 0, s, xp, wh, out, rows, T);
    else hipLaunchKernelGGL((bilstm_kernel<false, false>), grid, dim3(512), 0, s, xp, wh, out, rows, T);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("bilstm launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
