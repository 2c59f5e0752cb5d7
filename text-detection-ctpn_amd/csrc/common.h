// Internal declarations shared by the translation units of libctpn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/ctpn_hip.h"
#include "jpeg_pixel.h"

namespace ctpn {

// ---------------------------------------------------------------------------------------------
// 16-bit MFMA operand types. The throughput modes compute in bf16 (CTPN_PREC_BF16, BASELINE.json's dtype) or IEEE fp16
// (CTPN_PREC_FP16: same MFMA rate -- v_mfma_f32_32x32x16_f16 --, three more mantissa bits); the parity-grade
// mode CTPN_PREC_SPLIT stores every activation and weight as a (hi, lo) PAIR of bf16 and spends three bf16 MFMAs per product.
// Everything that only moves 16-bit payloads is type-blind; what differs is collected here: the packed convert, the widening, the
// MFMA opcode (builtin, and the mnemonic for the weights-in-registers kernel's inline asm).
// ---------------------------------------------------------------------------------------------
struct h_bf16 { uint16_t v; };
struct h_f16 { uint16_t v; };

#if defined(__HIPCC__)
typedef float ctpn_f32x2 __attribute__((ext_vector_type(2)));
typedef float ctpn_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ctpn_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ctpn_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ctpn_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ctpn_f16x8 __attribute__((ext_vector_type(8)));
// two fp32 -> packed bf16 (lo in bits 15:0), round-to-nearest-even: one v_cvt_pk_bf16_f32 instead of ~12 VALU ops.
// Bit-identical to the integer RNE formula for finite inputs (tests/test_gpu_parity.py::test_bf16_convert_matches_rne).
// Written as a vector fptrunc (hipcc selects v_cvt_pk_bf16_f32 for it on gfx950), NOT as inline asm: the hazard
// recognizer does not look into asm operands, so an asm convert that is the first reader of an MFMA result runs before
// the accumulator is written back (seen as NaNs in conv_first_mfma_kernel).
__device__ __forceinline__ unsigned int ctpn_cvt_pk_bf16(float lo, float hi) {
  const ctpn_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, ctpn_bf16x2));
}
__device__ __forceinline__ unsigned int ctpn_cvt_pk_f16(float lo, float hi) {
  const ctpn_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, ctpn_f16x2));      // v_cvt_pk_f16_f32 (RNE)
}
__device__ __forceinline__ float ctpn_bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned int)h << 16); }
__device__ __forceinline__ float ctpn_f16_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
// fp32 -> one bf16 (software RNE, NaN-preserving: the reference formula the hardware convert is tested against)
__device__ __forceinline__ unsigned short ctpn_f32_to_bf16(float f) {
  unsigned int u = __builtin_bit_cast(unsigned int, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <typename H> struct HalfOps;
template <> struct HalfOps<h_bf16> {
  static __device__ __forceinline__ unsigned int cvt_pk(float lo, float hi) { return ctpn_cvt_pk_bf16(lo, hi); }
  static __device__ __forceinline__ float to_f32(unsigned short h) { return ctpn_bf16_to_f32(h); }
  static __device__ __forceinline__ unsigned short from_f32(float f) { return ctpn_f32_to_bf16(f); }
  static __device__ __forceinline__ ctpn_f32x16 mfma_32x32x16(const uint4& a, const uint4& b, const ctpn_f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ctpn_bf16x8, a), __builtin_bit_cast(ctpn_bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct HalfOps<h_f16> {
  static __device__ __forceinline__ unsigned int cvt_pk(float lo, float hi) { return ctpn_cvt_pk_f16(lo, hi); }
  static __device__ __forceinline__ float to_f32(unsigned short h) { return ctpn_f16_to_f32(h); }
  static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
  static __device__ __forceinline__ ctpn_f32x16 mfma_32x32x16(const uint4& a, const uint4& b, const ctpn_f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ctpn_f16x8, a), __builtin_bit_cast(ctpn_f16x8, b), c, 0, 0, 0);
  }
};
// (hi, lo) bf16 pair of an fp32 value: hi = RNE(x), lo = RNE(x - hi) -- x - hi is exact in fp32, so hi + lo carries 16 mantissa
// bits (|x - hi - lo| <= 2^-17 |x|). Two values at a time: the packed converts.
__device__ __forceinline__ void ctpn_split_pk_bf16(float a, float b, unsigned int& hi, unsigned int& lo) {
  hi = ctpn_cvt_pk_bf16(a, b);
  lo = ctpn_cvt_pk_bf16(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}
#endif

void set_error(const std::string& s);
int fail(int code, const std::string& s);

#define CTPN_HIP_TRY(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::ctpn::fail(CTPN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Launch state is PER DEVICE: the ABI promises one ctx per GPU, and one process may hold ctxs on several GPUs. The CU count and the
// "MaxDynamicSharedMemorySize already raised for this kernel" flags are indexed by the current device (a function attribute set on
// device 0 does not carry over to device 1), under one mutex. `done` is one flag array per kernel instantiation (a function-local
// static of the launcher template).
// ---------------------------------------------------------------------------------------------
constexpr int CTPN_MAX_DEV = 16;
inline std::mutex& launch_state_mutex() { static std::mutex mu; return mu; }
inline int current_device(int& dev) {
  CTPN_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= CTPN_MAX_DEV) return fail(CTPN_ERR_ARG, "device index out of range (0..15)");
  return CTPN_OK;
}
inline int device_cu_count(int dev, int& ncu) {
  static int n[CTPN_MAX_DEV] = {0};
  std::lock_guard<std::mutex> lk(launch_state_mutex());
  if (!n[dev]) {
    hipDeviceProp_t p;
    CTPN_HIP_TRY(hipGetDeviceProperties(&p, dev));
    n[dev] = p.multiProcessorCount;
  }
  ncu = n[dev];
  return CTPN_OK;
}
inline int raise_dynamic_lds(const void* kern, int bytes, bool (&done)[CTPN_MAX_DEV], int dev) {
  std::lock_guard<std::mutex> lk(launch_state_mutex());
  if (!done[dev]) {
    CTPN_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[dev] = true;
  }
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM descriptor (conv3x3 over a zero-bordered NHWC buffer, or a plain row-major GEMM).
//   out[m][co] = act( bias[co] + sum_{tap, ci} A(m, tap, ci) * Wt[co][tap*Ci + ci] )
// A rows: conv mode -> pixel m = (n, y, x) of an n x (H+2) x (W+2) x Ci bordered buffer, tap (ky,kx)
// reads bordered pixel (y+ky, x+kx) (so ky=kx=1 is the centre); plain mode -> A[m*lda + ci].
// ---------------------------------------------------------------------------------------------
struct IGemm {
  const void* a;        // activations (T)
  const void* wt;       // packed weights, T, [co_pad][ntaps*Ci] (co_pad multiple of the N tile, zero rows)
  const float* bias;    // fp32 [co] or nullptr
  void* out;            // OutT
  long long M;          // rows (pixels)
  int Ci;               // channels per tap (multiple of 128/sizeof(T))
  int ntaps;            // 9 (3x3) or 1
  int Co;               // valid output channels
  // A addressing
  int a_plain;          // 1: A is row-major [M][lda]
  long long lda;        // plain: elements per row
  int H, W;             // conv: un-bordered spatial size of the input == output
  int rx0, rw;          // conv: rw > 0 restricts the rows to columns [rx0, rx0 + rw) of every image row (M = N*H*rw)
  int tap_base_y, tap_base_x;  // conv, ntaps==1: which bordered tap (1,1 = centre)
  // output addressing
  int out_bordered;     // 1: write into n x (H+2) x (W+2) x ldc bordered buffer at (y+1, x+1)
  long long ldc;        // elements per output row/pixel
  int relu;
};

// SPLIT: activations / weights as (hi, lo) bf16 pairs -- per pixel [hi(C) | lo(C)], per weight row and tap [hi(Ci) | hi(Ci) | lo(Ci)]
enum class DType { F32 = 0, BF16 = 1, F16 = 2, SPLIT = 3 };
static inline int dtype_bytes(DType t) { return (t == DType::BF16 || t == DType::F16) ? 2 : 4; }      // bytes per activation element
static inline bool dtype_is_half(DType t) { return t == DType::BF16 || t == DType::F16; }

// launchers (each returns hipGetLastError()-style status through int)
int launch_igemm(const IGemm& g, DType in_t, DType out_t, hipStream_t s);
// tap-reuse 3x3 conv (conv3x3.hip); pool_out != nullptr fuses the following 2x2/2 max-pool, out may then be null
// t == SPLIT: ci / co are the layer's channels, pixels are [hi | lo] bf16 planes (output [hi | lo | hi] with dup_hi), weights from
// launch_pack_transpose_split
int launch_conv3x3(const void* in, const void* wt, const float* bias, void* out, void* pool_out, DType t, int n, int h, int w,
                   int ci, int co, int relu, hipStream_t s, int dup_hi = 0, const void* q1 = nullptr,
                   const void* q1_frags = nullptr, int split_opts = 3 /* split precision. Bit 0 (option conv_p64): Co = 64 through the persistent kernel's 64-channel form
                                                                      instead of the non-persistent kernel; bit 1 (option split_edge): ragged tile columns through the edge
                                                                      kernel instead of a padded tile column */);
bool conv1_fusable(DType t, int n, int h, int w, int ci, int co, bool pool, bool keep_full);
// mfma_frags != nullptr: conv1_1 on the matrix cores with split-bf16 operands (pack_conv1_frags; fp32-class sums; SPLIT stores
// [hi(64) | lo(64)] per pixel); nullptr: the VALU kernel. (The uint8 feed of the 16-bit modes goes through the q-image instead, below.)
int launch_conv_first(const void* img, int img_is_f32, const float* w27x64, const float* bias, void* out, DType out_t,
                      int n, int h, int w, hipStream_t s, const void* mfma_frags = nullptr);
constexpr int CF_FRAG_BYTES = 12 * 64 * 16;   // [co tile 2][ky 3][hi|lo][64 lanes] x 8 bf16
constexpr int CFP_FRAG_BYTES = 6 * 64 * 16;   // conv1_1 over the q-image (conv_first_p_kernel and the producer inside conv3x3_wr_kernel): [co tile 2][ky 3][64 lanes] x 8 halves, stored behind the split fragments: a bf16 set, then an fp16 set
constexpr int CF_FRAGS_TOTAL = CF_FRAG_BYTES + 2 * CFP_FRAG_BYTES;
static inline const void* conv1_p_frags(const void* frags, DType t) { return (const char*)frags + CF_FRAG_BYTES + (t == DType::F16 ? CFP_FRAG_BYTES : 0); }
int pack_conv1_frags(const float* w27x64_dev, const float* bias_dev, uint4* frags_dev);
// ---- the q-image: the uint8 feed of the 16-bit modes as 8-byte pixels (q_B, q_G, q_R, P) of the mode's 16-bit type, q_c = p_c - round(mean_c)
// (an integer, exact in bf16 and fp16), P = 1.0; image pixel (y, x) sits at q pixel (y + 2, x + 2) of an Hq x Wq map whose other pixels are
// all-zero -- TF's SAME padding of conv1_1 AND the inside-the-image indicator its mean correction needs (layers.hip). 4.4 MB per 600 x 900
// image instead of the 69 MB of conv1_1's output: what conv1_2 reads when conv1_1 is computed inside its window stage (conv3x3_impl.h).
static inline int conv1_q_h(int h) { return ((h + 7) / 8) * 8 + 4; }
static inline int conv1_q_w(int w) { return ((w + 63) / 64) * 64 + 8; }
static inline size_t conv1_q_bytes(int n, int h, int w) { return ((size_t)n * conv1_q_h(h) * conv1_q_w(w) + 1024) * 8; }
int launch_image_to_q(const uint8_t* img, void* q, DType t, int n, int h, int w, hipStream_t s);
// conv1_1 from the q-image into the bordered NHWC map `out`, columns [xb, xe) of every image row (keep_acts; the ragged columns conv1_2's
// edge kernel reads): the same MFMA sequence on the same operands as the fused producer, so the two agree bit for bit
int launch_conv_first_from_q(const void* q, const void* frags, void* out, DType t, int n, int h, int w, int xb, int xe, hipStream_t s);
// cv2.resize(INTER_LINEAR) restated (preprocess.hip); src / dst: n x h x w x 3 and n x dh x dw x 3, uint8 or float32, device pointers
int resize_out_dim(int src, double f);
int launch_resize_linear(const void* src, void* dst, int is_f32, int n, int h, int w, int dh, int dw, double fx, double fy, hipStream_t s);
int launch_cvt_bf16(const float* in, uint16_t* out, int n, int hw, hipStream_t s);
// the two LDS-DMA helper forms of conv3x3_impl.h on `tiles` 1-KiB tiles of src (conv3x3.hip; test hook behind ctpn_debug_lds_dma)
int launch_lds_dma_check(const void* src, void* out_clobber, void* out_keep, int tiles, hipStream_t s);
int launch_pack_transpose(const float* src, long long src_ld, void* dst, long long dst_ld, DType dst_t,
                          int rows, int cols, hipStream_t s);
// split precision: src [taps * ci][cols] fp32 (TF HWIO / [in][out]) -> dst [cols][taps][hi(ci) | hi(ci) | lo(ci)] bf16
int launch_pack_transpose_split(const float* src, long long src_ld, void* dst, int taps, int ci, int cols, hipStream_t s);
// BiLSTM recurrence: xp [rows][T][1024] fp32 (fw gates 0..511 | bw gates 512..1023, TF order i,j,f,o, bias
// already added), wh [2][128][512] fp32, out [rows][T][256] fp32
// split_bf16: the recurrent product through three bf16 MFMAs on hi/lo operand halves (fp32-class accuracy, bf16 mode only)
// xp's gate columns are in the PERMUTED order lstm_gate_col() (a lane's 4 gates x 4 units contiguous), see bilstm.hip
// fast_gates: v_exp / v_rcp gate math in the exact-fp32 kernel (bf16 throughput mode)
// xp_is_f16: the pre-activations are IEEE fp16 (16-bit throughput modes), else fp32
int launch_bilstm(const void* xp, int xp_is_f16, const float* wh, float* out, int rows, int T, hipStream_t s, int split_bf16 = 0, int fast_gates = 0);
// lstm_pre.hip: the LSTM input projection of the 16-bit modes (resident weight slice, fp16 out): a = bordered NHWC map of n x hf x wf cells x 512
int launch_lstm_pre(const void* a, const void* wt_frag, const float* bias, void* out, DType t, int n, int hf, int wf, hipStream_t s);
int launch_lstm_pre_pack(const void* wt_x, void* wt_frag, hipStream_t s);      // [1024][512] 16-bit rows -> the kernel's fragment-major order (1 MB)
int lstm_gate_col(int c);     // TF gate column (g * 128 + u) -> permuted column, per direction
int launch_lstm_permute_rows(const void* src, void* dst, int row_bytes, hipStream_t s);
// proposal pipeline
struct ProposalCfg {
  int n, hf, wf;
  int pre_nms_topn, post_nms_topn;
  float nms_thresh, min_size;
};
// heads: [n*hf*wf][head_ld] fp32, cols 0..39 bbox deltas (a*4+{dx,dy,dw,dh}), 40..59 cls scores (a*2+{bg,fg})
int launch_decode(const float* heads, int head_ld, int heads_are_probs, const float* cls_prob_in,
                  const float* bbox_in, const float* im_info_dev, float* cls_prob_out, float* bbox_out,
                  unsigned long long* keys, float* boxes4, const ProposalCfg& c, int npad, hipStream_t s,
                  bool skip_fill = false /* the keys behind the image's anchors are not read (launch_sort_keys' segmented form) */,
                  const float* im_info_host = nullptr /* n <= 4: the rows travel in the kernel arguments and decode_kernel writes im_info_dev itself */);
bool sort_is_segmented(int n_img, int per_img);      // will launch_sort_keys(..., in_tmp != nullptr) take the segmented form?
// stable radix sort of the first per_img keys of every npad-strided segment (tmp: same size as keys)
// in_tmp != nullptr allows the segmented form for small batches; *in_tmp says which buffer holds the sorted keys afterwards
int launch_sort_keys(unsigned long long* keys, unsigned long long* tmp, int n_img, int npad, int per_img, hipStream_t s, int* in_tmp = nullptr);
// sorted_anchor (nullable): [n_img][topn] anchor index (y, x, a row-major) of every sorted row
int launch_gather_sorted(const unsigned long long* keys, const float* boxes4, float* sorted_boxes,
                         float* sorted_scores, int* sorted_anchor, int* valid_counts, int n_img, int npad,
                         int n_anchors_total, int topn, hipStream_t s,
                         unsigned char* colid = nullptr /* [n_img][(topn + 15) & ~15]: column group of every sorted box (launch_nms_columns' multi-workgroup form) */,
                         int ncols = 0);
// sorted_boxes [n_img][stride][4]; counts_in [n_img] (boxes per image, <= stride); keep idx out [n_img][keep_stride]
int launch_nms(const float* sorted_boxes, const float* sorted_scores, const int* counts_in, int stride,
               float thresh, int max_keep, int* keep_idx, int keep_stride, int* keep_counts, float* rois_out,
               float* kept_spill /* [n_img][stride][4] scratch */, int n_img, hipStream_t s,
               const int* sorted_anchor = nullptr /* [n_img][stride] */, int* roi_anchor = nullptr /* [n_img][max_keep] */);

// column-decomposed form for the proposal layer's boxes (16 px anchors on a 16 px grid): same result, see proposal.hip.
// PRECONDITION (not checked by nms_columns_ok, which only looks at ncols / stride / thresh): every box lies on the 16-px anchor grid,
// x1 in [16 c, 16 c + 16) and x2 <= 16 c + 16 for its column c (/ im_scale for the connector variant) -- true for decode_kernel's output
// (bbox_transform_inv leaves x alone), NOT for arbitrary boxes: those go through launch_nms (the ctpn_nms seam always does).
// option nms_check = 1 (debug) re-runs the generic kernel after the proposal layer's column launch (ctpn_api.hip) and fails loudly on a mismatch.
int launch_nms_columns(const float* sorted_boxes, const float* sorted_scores, const int* counts_in, int stride, float thresh, int max_keep,
                       int* keep_idx, int keep_stride, int* keep_counts, float* rois_out, float* kept_spill, int n_img, int ncols, hipStream_t s,
                       const int* sorted_anchor = nullptr, int* roi_anchor = nullptr,
                       const float* col_scale = nullptr /* im_info rows: the connector's boxes / im_scale variant */,
                       void* mw_scratch = nullptr /* n_img x NMS_MW_SCRATCH_BYTES, zeroed: the multi-workgroup form for small batches (one column per wave) */,
                       const unsigned char* colid = nullptr /* launch_gather_sorted's column ids (needed above 1024 candidates) */,
                       int prefix = 0 /* > 0: try the first `prefix` ranks first (they usually hold max_keep survivors); same result either way */);
constexpr size_t NMS_MW_SCRATCH_BYTES = 2048;       // per image: survivor mask (one bit per rank) + ticket; zero between launches
// ... and one STICKY word the kernel sets when a column held more candidates than its list (keep lists are then wrong): zero unless a caller
// broke launch_nms_columns' precondition; read and cleared by the host (option nms_check). The block's zero state between launches is
// restored by the kernel's own epilogue; launches that share a block are serialised by stream order (proposal NMS, then connector NMS, on
// one stream per submit), and a host path that cannot vouch for an epilogue having run (an error between launches, an option change)
// marks the ctx (nms_mw_dirty) so that the next launch is preceded by a memset of the block
constexpr size_t NMS_MW_OVERFLOW_OFF = 2040;
constexpr size_t NMS_MW_FLAG_OFF = 2032;            // prefix pass: "the prefix launch did not answer" (written by every stage-1 launch before the stage-2 launch reads it)
constexpr int NMS_MW_MAX_BATCH = 4;                 // batches up to this size spread their columns over the machine; larger ones fill it with images
constexpr int NMS_MW_CAP_BATCH = 32;                // ... unless option nms_columns = 3 asks for the multi-workgroup form explicitly: buffers are sized for this many images
bool nms_columns_ok(int ncols, int stride, float thresh);
int launch_hog(unsigned* sink, int n_wg, int usec, int touch, hipStream_t s, const void* src = nullptr, size_t src_bytes = 0);
extern int g_debug_nms;      // diagnostic parts mask of nms_columns_kernel<16, ..> (proposal.hip)      // diagnostic: the one-workgroup NMS's footprint without its work (proposal.hip)
bool nms_columns_tl_ok(int ncols, int stride, float thresh, float max_scale);

// host text connector (text_connector.cpp)
int text_lines_host(const float* boxes, const float* scores, int r, int im_h, int im_w, int mode,
                    int device_id, std::vector<double>& recs);
int connect_lines(const float* kept_boxes, const float* kept_scores, int n, int im_h, int im_w, int mode,
                  std::vector<double>& recs);
// rois [n_img][post][5] (descending score) -> per image: boxes/scale of the score > min_score prefix + its length
constexpr int CONN_CAP = 512;   // text lines per image and mode the device connector can return (chains <= proposals / 2 = 500)
// text-line connector on the device: recs [n_img][2 modes][cap][9] float64, counts [n_img][3] = lines H, lines O, status;
// scratch [n_img][1024][20] float64
int launch_connect(const float* boxes, const float* scores, const int* keep, const int* keep_counts, int stride, const float* im_info,
                   double* recs, int* counts, double* scratch, int cap, int n_img, hipStream_t s);
int launch_lines_prep(const float* rois, const int* roi_counts, const float* im_info, int post, float min_score,
                      float* tl_boxes, float* tl_scores, int* tl_counts, int n_img, hipStream_t s);
// host greedy NMS used by the connector when device_id < 0 (same predicate as the device kernel)
void nms_host(const float* boxes, int n, int dim, float thresh, std::vector<int>& keep);

// jpeg.hip: JPEG files, entropy decoding on the host, pixels on the device (JpegGeom: jpeg_pixel.h)
int jpeg_entropy_decode(const uint8_t* data, size_t len, int16_t* coef, size_t coef_cap, uint16_t* qt3x64, JpegGeom* g);
int launch_jpeg_pixels(const int16_t* coef_dev, const uint16_t* qt_dev, uint8_t* planes_dev, uint8_t* out_dev, const JpegGeom& g, int n, hipStream_t s);
int jpeg_probe(const uint8_t* data, size_t len, int* h, int* w, int* ncomp, int* luma_sampling);
size_t jpeg_coef_capacity(int h, int w);

static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace ctpn
