// conv3x3 kernels instantiated for exact-fp32 MFMAs (CTPN_PREC_FP32: the correctness-gate path), see conv3x3_impl.h
#include "conv3x3_impl.h"
namespace ctpn {
int c3_run_f32(const Conv3& g, bool pool, hipStream_t s) { return c3_dispatch<float>(g, pool, s); }
}  // namespace ctpn
