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
//   * the next step's pre-activations (4 x 16 B per lane, row strips of lstm_pre) are requested before the MFMAs.
#include "common.h"

namespace ctpn {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f / (expf(2.f * x) + 1.f); }

constexpr int LSTM_ROWS = 16;
constexpr int LSTM_HPITCH = 136;  // floats per h row in LDS (128 + 8 pad = 544 B)

__global__ __launch_bounds__(512) void bilstm_kernel(const float* __restrict__ xp, const float* __restrict__ wh,
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
  const float* xrow = xp + (size_t)row_c * T * 1024 + dir * 512 + u0;
  float* orow = out + (size_t)row_c * T * 256 + dir * 128 + u0;

  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  f32x4 pre[4];
  {
    const int t0 = dir ? T - 1 : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] = *(const f32x4*)(xrow + (size_t)t0 * 1024 + g * 128);
  }
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = pre[g];
    if (s + 1 < T) {
      const int tn = dir ? t - 1 : t + 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[g] = *(const f32x4*)(xrow + (size_t)tn * 1024 + g * 128);
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
      const float ig = sigmoidf_(acc[0][e]);
      const float jg = tanhf_(acc[1][e]);
      const float fg = sigmoidf_(acc[2][e] + 1.0f);
      const float og = sigmoidf_(acc[3][e]);
      const float cn = fg * c[e] + ig * jg;
      c[e] = cn;
      h[e] = og * tanhf_(cn);
    }
    *(f32x4*)(&hbuf[cur ^ 1][row_l][u0]) = h;
    if (row_ok) *(f32x4*)(orow + (size_t)t * 256) = h;
    __syncthreads();
  }
}

int launch_bilstm(const float* xp, const float* wh, float* out, int rows, int T, hipStream_t s) {
  if (rows <= 0 || T <= 0) return fail(CTPN_ERR_ARG, "bilstm: empty problem");
  dim3 grid((rows + LSTM_ROWS - 1) / LSTM_ROWS, 2);
  hipLaunchKernelGGL(bilstm_kernel, grid, dim3(512), 0, s, xp, wh, out, rows, T);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("bilstm launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
