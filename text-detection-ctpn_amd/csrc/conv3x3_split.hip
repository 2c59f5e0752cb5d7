// conv3x3 kernels instantiated for CTPN_PREC_SPLIT: (hi, lo) bf16 planes, three bf16 MFMA terms per product (fp32-class results at the
// matrix cores' 16-bit rate / 3), see conv3x3_impl.h
#include "conv3x3_impl.h"
namespace ctpn {
int c3_run_split(const Conv3& g, bool pool, hipStream_t s) { return c3_dispatch<h_bf16, true>(g, pool, s); }
int c3_edge_split(const void* in, const void* wt, const float* bias, void* out, int n, int h, int w, int ci, int co, int relu, int r, bool pooled, hipStream_t s, bool deep, int dup_hi) {
  return c3_launch_edge<h_bf16, true>(in, wt, bias, out, n, h, w, ci, co, relu, r, pooled, s, deep, dup_hi);
}
}  // namespace ctpn
