// conv3x3 kernels instantiated for bf16 MFMAs (CTPN_PREC_BF16: BASELINE.json's throughput dtype), see conv3x3_impl.h
#include "conv3x3_impl.h"
namespace ctpn {
int c3_run_bf16(const Conv3& g, bool pool, bool wr, hipStream_t s) { return wr ? c3_launch_wr<h_bf16>(g, pool, s) : c3_dispatch<h_bf16>(g, pool, s); }
int c3_edge_bf16(const void* in, const void* wt, const float* bias, void* out, int n, int h, int w, int ci, int co, int relu, int r, bool pooled, hipStream_t s, bool deep) {
  return c3_launch_edge<h_bf16>(in, wt, bias, out, n, h, w, ci, co, relu, r, pooled, s, deep);
}
}  // namespace ctpn
