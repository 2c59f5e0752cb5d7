// conv3x3 kernels instantiated for IEEE fp16 MFMAs (CTPN_PREC_FP16: same rate as bf16, three more mantissa bits), see conv3x3_impl.h
#include "conv3x3_impl.h"
namespace ctpn {
int c3_run_f16(const Conv3& g, bool pool, bool wr, hipStream_t s) { return wr ? c3_launch_wr<h_f16>(g, pool, s) : c3_dispatch<h_f16>(g, pool, s); }
int c3_edge_f16(const void* in, const void* wt, const float* bias, void* out, int n, int h, int w, int ci, int co, int relu, int r, bool pooled, hipStream_t s, bool deep) {
  return c3_launch_edge<h_f16>(in, wt, bias, out, n, h, w, ci, co, relu, r, pooled, s, deep);
}
}  // namespace ctpn
