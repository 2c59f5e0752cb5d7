// Proposal layer on device: softmax + anchor decode + clip + min-size filter -> key sort -> greedy NMS.
//
// Replaces, per image and batched over images (the reference is batch-1 Python on a TF CPU thread):
//   spatial_softmax                       reference lib/networks/network.py:332-337
//   generate_anchors                      lib/rpn_msr/generate_anchors.py:3-32   (py3 table)
//   proposal_layer steps 1-8              lib/rpn_msr/proposal_layer_tf.py:65-155
//   bbox_transform_inv / clip_boxes       lib/fast_rcnn/bbox_transform.py:36-80  (dx, dw ignored, :50,52)
//   _filter_boxes                         lib/rpn_msr/proposal_layer_tf.py:160-165
//   nms -> gpu_nms -> _nms / nms_kernel   lib/fast_rcnn/nms_wrapper.py:11-20, lib/utils/nms_kernel.cu:24-143
//
// All box arithmetic is fp32 in numpy's operation order with FMA contraction disabled, so decoded boxes
// match the reference except for the last ulp of exp(); the NMS predicate is evaluated exactly as the CUDA
// kernel does (IEEE fp32 divide, `IoU > thr`), so for identical sorted boxes the keep list is bit-identical.
// Tie order of the sort is fixed: descending score, equal scores by ascending anchor index (h, w, a).
#include <cstdlib>

#include "common.h"

#pragma clang fp contract(off)

namespace ctpn {

// y1, y2 of the 10 base anchors (x1 = 0, x2 = 15), python-3 division + int32 truncation
__constant__ int c_anchor_y1[10] = {2, 0, -4, -9, -16, -26, -41, -62, -91, -134};
__constant__ int c_anchor_y2[10] = {13, 15, 19, 24, 31, 41, 56, 77, 106, 149};

constexpr unsigned long long KEY_INVALID = 0xFFFFFFFFFFFFFFFFull;

// float -> uint32 whose unsigned order is the float order (negative values below positive ones); inverse below
__device__ __forceinline__ unsigned int score_order_bits(float f) {
  const unsigned int u = __builtin_bit_cast(unsigned int, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float score_from_order_bits(unsigned int o) {
  return __builtin_bit_cast(float, (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// column group of a box for the column-decomposed NMS kernels below (16-px anchor grid; cs = im_scale for the connector's boxes / scale)
constexpr int NC_MAXN = 12288, NC_MAXCOL = 256, NC_TL_MAXN = 1024;
__device__ __forceinline__ int nms_col_of(float x1, float cs, int ncols) {
  const int c = (int)(x1 * cs + 0.5f) >> 4;
  return c < 0 ? 0 : (c > ncols - 1 ? ncols - 1 : c);
}

struct ImInfoSmall { float v[12]; };      // im_info rows [h, w, scale] of up to four images, by value

// ---------------------------------------------------------------------------------------------
// decode: one thread per anchor (n, y, x, a)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ heads, int head_ld,
                                                     const float* __restrict__ cls_prob_in, const float* __restrict__ bbox_in,
                                                     const float* __restrict__ im_info, float* __restrict__ cls_prob_out,
                                                     float* __restrict__ bbox_out, unsigned long long* __restrict__ keys,
                                                     float* __restrict__ boxes4, int n_img, int hf, int wf, float min_size,
                                                     int npad, ImInfoSmall small, float* __restrict__ im_info_pub) {
  const int per_img = hf * wf * 10;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  // small batches: im_info arrives in the kernel arguments (no 12-byte host-to-device copy in front of this kernel: ~5 us of a lone image's
  // tail) and is published here for the kernels behind this one (lines_prep, the connector's NMS, connect)
  if (im_info_pub && gid < 3 * n_img) {
    const int k = (int)gid;
    float v = small.v[0];
#pragma unroll
    for (int q = 1; q < 12; ++q) v = k == q ? small.v[q] : v;
    im_info_pub[k] = v;
  }
  if (gid >= (long long)n_img * per_img) return;
  const int img = (int)(gid / per_img);
  const int idx = (int)(gid - (long long)img * per_img);  // (y, x, a) row-major
  const int a = idx % 10;
  const int cell = idx / 10;
  const int x = cell % wf, y = cell / wf;
  const long long m = (long long)img * hf * wf + cell;

  float dy, dh, score;
  if (heads) {
    const float* hrow = heads + m * head_ld;
    const float4 d = *(const float4*)(hrow + a * 4);
    const float s0 = hrow[40 + 2 * a], s1 = hrow[40 + 2 * a + 1];
    const float mx = fmaxf(s0, s1);
    const float e0 = expf(s0 - mx), e1 = expf(s1 - mx);
    const float sum = e0 + e1;
    const float p0 = e0 / sum, p1 = e1 / sum;
    dy = d.y; dh = d.w; score = p1;
    if (cls_prob_out) { cls_prob_out[m * 20 + 2 * a] = p0; cls_prob_out[m * 20 + 2 * a + 1] = p1; }
    if (bbox_out) *(float4*)(bbox_out + m * 40 + a * 4) = d;
  } else {
    const float4 d = *(const float4*)(bbox_in + m * 40 + a * 4);
    dy = d.y; dh = d.w; score = cls_prob_in[m * 20 + 2 * a + 1];
  }

  float imH, imW, imS;
  if (im_info_pub) {                                       // (selects: a run-time index into kernel arguments would go through scratch)
    imH = img == 0 ? small.v[0] : img == 1 ? small.v[3] : img == 2 ? small.v[6] : small.v[9];
    imW = img == 0 ? small.v[1] : img == 1 ? small.v[4] : img == 2 ? small.v[7] : small.v[10];
    imS = img == 0 ? small.v[2] : img == 1 ? small.v[5] : img == 2 ? small.v[8] : small.v[11];
  } else { imH = im_info[img * 3 + 0]; imW = im_info[img * 3 + 1]; imS = im_info[img * 3 + 2]; }
  // shifted anchor (int -> fp32), bbox_transform_inv in numpy's fp32 operation order
  const float ax1 = (float)(x * 16), ax2 = (float)(x * 16 + 15);
  const float ay1 = (float)(y * 16 + c_anchor_y1[a]), ay2 = (float)(y * 16 + c_anchor_y2[a]);
  const float widths = ax2 - ax1 + 1.0f;
  const float heights = ay2 - ay1 + 1.0f;
  const float ctr_x = ax1 + 0.5f * widths;
  const float ctr_y = ay1 + 0.5f * heights;
  const float pred_ctr_y = dy * heights + ctr_y;
  const float pred_h = expf(dh) * heights;
  float x1 = ctr_x - 0.5f * widths;
  float y1 = pred_ctr_y - 0.5f * pred_h;
  float x2 = ctr_x + 0.5f * widths;
  float y2 = pred_ctr_y + 0.5f * pred_h;
  // clip_boxes: max(min(v, lim - 1), 0)
  const float wl = imW - 1.0f, hl = imH - 1.0f;
  x1 = fmaxf(fminf(x1, wl), 0.0f);
  y1 = fmaxf(fminf(y1, hl), 0.0f);
  x2 = fmaxf(fminf(x2, wl), 0.0f);
  y2 = fmaxf(fminf(y2, hl), 0.0f);
  // _filter_boxes
  const float ms = min_size * imS;
  const float ws = x2 - x1 + 1.0f, hs = y2 - y1 + 1.0f;
  const bool keep = (ws >= ms) && (hs >= ms);

  *(float4*)(boxes4 + ((long long)img * per_img + idx) * 4) = make_float4(x1, y1, x2, y2);
  // High word = ~(order-preserving image of the score): ascending key = descending score for EVERY finite score (also 0.0
  // and negative values handed in through ctpn_proposals_from_host). The image of a finite float is never 0, so the high
  // word of a valid key is never 0xFFFFFFFF: KEY_INVALID sorts strictly after every valid key and the valid keys form a
  // prefix of the sorted segment (the radix sort only orders the high word). NaN never passes `keep`.
  const unsigned long long key = keep && (score == score)
                                     ? (((unsigned long long)(~score_order_bits(score))) << 32) | (unsigned int)idx
                                     : KEY_INVALID;
  keys[(long long)img * npad + idx] = key;
}

__global__ void fill_keys_kernel(unsigned long long* keys, int n_img, int npad, int per_img) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int tail = npad - per_img;
  if (gid >= (long long)n_img * tail) return;
  const int img = (int)(gid / tail);
  const int i = (int)(gid - (long long)img * tail);
  keys[(long long)img * npad + per_img + i] = KEY_INVALID;
}

int launch_decode(const float* heads, int head_ld, int heads_are_probs, const float* cls_prob_in, const float* bbox_in,
                  const float* im_info_dev, float* cls_prob_out, float* bbox_out, unsigned long long* keys, float* boxes4,
                  const ProposalCfg& c, int npad, hipStream_t s, bool skip_fill, const float* im_info_host) {
  const int per_img = c.hf * c.wf * 10;
  const long long total = (long long)c.n * per_img;
  if (npad > per_img && !skip_fill) {      // (the segmented sort of small batches never reads behind the image's keys and pads its merged buffer itself)
    const long long tail = (long long)c.n * (npad - per_img);
    hipLaunchKernelGGL(fill_keys_kernel, dim3((unsigned)((tail + 255) / 256)), dim3(256), 0, s, keys, c.n, npad, per_img);
  }
  ImInfoSmall small{};
  if (im_info_host && c.n <= 4) for (int i = 0; i < 3 * c.n; ++i) small.v[i] = im_info_host[i];
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     heads_are_probs ? nullptr : heads, head_ld, cls_prob_in, bbox_in, im_info_dev, cls_prob_out, bbox_out,
                     keys, boxes4, c.n, c.hf, c.wf, c.min_size, npad, small, im_info_host && c.n <= 4 ? (float*)im_info_dev : nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("decode launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// Stable LSD radix sort of the same keys, one workgroup (16 waves) per image: four 8-bit passes over the score half of the
// key (the low half is the anchor index and the keys start in index order, so stability IS the ascending-index tie order;
// keys are unique per image, so the order is total). Per pass: every wave counts the digits of
// its contiguous segment (64 keys at a time; lanes with the same digit find each other with eight ballots, the lowest one
// adds the group's size to the wave's private histogram row), a digit-major scan turns the 16 x 256 counts into segment
// bases, and the same walk scatters the keys (rank inside the 64 = popcount of the lower lanes in the group).
// Sort stage (sort + gather, sharing the GPU with the next batch's convolutions): 20 720 keys x 32 images 0.68 -> 0.29 ms;
// 96 000 keys x 8 images (1280 x 1920) 2.9 -> 1.4 ms; one image 0.48 -> 0.16 ms against round 1's bitonic network (removed in round 3).
// ---------------------------------------------------------------------------------------------
constexpr int RS_WAVES = 16, RS_TILE = 8;
constexpr int MG_SEGS = 8, MG_SP = 4, MG_MAXSEG = 4096;      // segmented form for small batches (below): <= 8 x 4096 keys per image

__device__ __forceinline__ unsigned long long rs_match(unsigned d, bool valid) {
  unsigned long long mask = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bal = __ballot(bit);
    mask &= bit ? bal : ~bal;
  }
  return mask;      // lanes (valid ones) that hold the same digit as this lane
}

// grid (segments, images): block (g, i) sorts keys [g * seg_len, min((g + 1) * seg_len, n_total)) of image i in place (one segment = the
// whole image unless the caller merges afterwards: merge_rank_kernel)
// IN_LDS (segments of at most MG_MAXSEG keys: 2 x 32 KB of dynamic LDS): the segment is fetched once, the four passes ping-pong between two
// LDS buffers, the sorted segment is stored once -- a pass through HBM costs two dependent round trips and n scattered 8-byte stores.
template <bool IN_LDS>
__global__ __launch_bounds__(RS_WAVES * 64) void radix_sort_kernel(unsigned long long* __restrict__ keys_all, unsigned long long* __restrict__ tmp_all,
                                                                   int npad, int n_total, int seg_len) {
  __shared__ unsigned hist[RS_WAVES][256];
  __shared__ unsigned colbase[256];
  extern __shared__ __attribute__((aligned(16))) unsigned long long rs_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seg0 = (int)blockIdx.x * seg_len;
  const int n = n_total - seg0 < seg_len ? (n_total - seg0 > 0 ? n_total - seg0 : 0) : seg_len;
  unsigned long long* const ga = keys_all + (size_t)blockIdx.y * npad + seg0;
  unsigned long long* a = IN_LDS ? rs_dyn : ga;
  unsigned long long* b = IN_LDS ? rs_dyn + MG_MAXSEG : tmp_all + (size_t)blockIdx.y * npad + seg0;
  if constexpr (IN_LDS) {
    unsigned long long st[MG_MAXSEG / (RS_WAVES * 64)];
#pragma unroll
    for (int j = 0; j < MG_MAXSEG / (RS_WAVES * 64); ++j) { const int i = tid + j * RS_WAVES * 64; st[j] = i < n ? ga[i] : 0ull; }
#pragma unroll
    for (int j = 0; j < MG_MAXSEG / (RS_WAVES * 64); ++j) { const int i = tid + j * RS_WAVES * 64; if (i < n) a[i] = st[j]; }
    __syncthreads();
  }
  const int seg = (((n + RS_WAVES - 1) / RS_WAVES) + 63) & ~63;
  const int lo = wave * seg < n ? wave * seg : n;
  const int hi = lo + seg < n ? lo + seg : n;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 32 + 8 * pass;
    for (int i = tid; i < RS_WAVES * 256; i += RS_WAVES * 64) (&hist[0][0])[i] = 0u;
    __syncthreads();
    // RS_TILE chunks of 64 keys per step, their loads in flight together: a wave's walk over its segment is a chain of dependent steps, and
    // one HBM / L2 latency per 64 keys was most of the kernel's time (one image: 180 us for 20 720 keys)
    for (int tb = lo; tb < hi; tb += 64 * RS_TILE) {
      unsigned long long kk[RS_TILE];
#pragma unroll
      for (int j = 0; j < RS_TILE; ++j) { const int idx = tb + 64 * j + lane; kk[j] = idx < hi ? a[idx] : 0ull; }
#pragma unroll
      for (int j = 0; j < RS_TILE; ++j) {
        if (tb + 64 * j >= hi) break;                       // wave-uniform: a segment of the segmented form is three chunks, not a tile of eight
        const bool valid = tb + 64 * j + lane < hi;
        const unsigned d = valid ? (unsigned)(kk[j] >> shift) & 255u : 0u;
        const unsigned long long m = rs_match(d, valid);
        if (valid && (m & lt) == 0ull) hist[wave][d] += (unsigned)__popcll(m);      // group leader; distinct digits -> distinct words
      }
    }
    __syncthreads();
    if (tid < 256) {              // digit `tid`: per-wave counts -> exclusive prefix inside the digit; total to colbase
      unsigned s = 0;
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) { const unsigned v = hist[w][tid]; hist[w][tid] = s; s += v; }
      colbase[tid] = s;
    }
    __syncthreads();
    if (wave == 0) {              // exclusive scan of the 256 digit totals: 4 per lane + a wave scan
      unsigned v[4], s = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] = colbase[4 * lane + q]; s += v[q]; }
      unsigned incl = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
      }
      unsigned run = incl - s;
#pragma unroll
      for (int q = 0; q < 4; ++q) { colbase[4 * lane + q] = run; run += v[q]; }
    }
    __syncthreads();
    for (int tb = lo; tb < hi; tb += 64 * RS_TILE) {
      unsigned long long kk[RS_TILE];
#pragma unroll
      for (int j = 0; j < RS_TILE; ++j) { const int idx = tb + 64 * j + lane; kk[j] = idx < hi ? a[idx] : 0ull; }
#pragma unroll
      for (int j = 0; j < RS_TILE; ++j) {
        if (tb + 64 * j >= hi) break;
        const bool valid = tb + 64 * j + lane < hi;
        const unsigned long long key = kk[j];
        const unsigned d = valid ? (unsigned)(key >> shift) & 255u : 0u;
        const unsigned long long m = rs_match(d, valid);
        if (valid) {
          const unsigned off = hist[wave][d];                                       // read by the whole group before its leader bumps it
          b[colbase[d] + off + (unsigned)__popcll(m & lt)] = key;
          if ((m >> lane) == 1ull) hist[wave][d] = off + (unsigned)__popcll(m);     // highest lane of the group
        }
      }
    }
    __syncthreads();
    unsigned long long* t = a; a = b; b = t;
  }
  if constexpr (IN_LDS) {                  // four swaps: the result is in the first LDS buffer
    for (int i = tid; i < n; i += RS_WAVES * 64) ga[i] = a[i];
  }
}

// Small batches (round 5): one image's sort ran on ONE workgroup, and every one of its 4 x n scattered 8-byte stores went through that one
// CU's memory pipeline (124 us for 20 720 keys). The image is cut into MG_SEGS segments sorted by one workgroup each (the kernel above), then
// merged by RANK: a key's place in the whole = its place in its own segment + the number of keys before it in every other segment (ties --
// only the KEY_INVALID tail has any -- go to the lower segment first: a stable merge, unique ranks). The count is a binary search over every
// MG_SP-th key of the other segment, staged in LDS, then a look at the MG_SP - 1 keys behind the splitter it stops at.

// grid (ceil(n / 1024), images): ONE key per thread, so that the kernel is three dependent HBM round trips long (splitters, the key, the
// windows) whatever the segment size; every workgroup stages all splitters (n / MG_SP x 8 B <= 64 KB)
__global__ __launch_bounds__(1024) void merge_rank_kernel(const unsigned long long* __restrict__ keys_all, unsigned long long* __restrict__ out_all,
                                                          int npad, int n_total, int seg_len) {
  __shared__ unsigned long long spl[MG_SEGS][MG_MAXSEG / MG_SP];
  const int tid = threadIdx.x;
  const unsigned long long* keys = keys_all + (size_t)blockIdx.y * npad;
  unsigned long long* out = out_all + (size_t)blockIdx.y * npad;
  const int per = seg_len / MG_SP;                                // splitters per full segment (seg_len is a multiple of 64)
  {
    unsigned long long st[MG_SEGS * (MG_MAXSEG / MG_SP) / 1024];
#pragma unroll
    for (int j = 0; j < MG_SEGS * (MG_MAXSEG / MG_SP) / 1024; ++j) {
      const int e = tid + 1024 * j, g = e / per, i = e - g * per;
      const long long src = (long long)g * seg_len + (long long)i * MG_SP;
      st[j] = (g < MG_SEGS && src < n_total) ? keys[src] : KEY_INVALID;
    }
#pragma unroll
    for (int j = 0; j < MG_SEGS * (MG_MAXSEG / MG_SP) / 1024; ++j) {
      const int e = tid + 1024 * j, g = e / per, i = e - g * per;
      if (g < MG_SEGS) spl[g][i] = st[j];
    }
  }
  const int t = blockIdx.x * 1024 + tid;
  const bool live = t < n_total;
  const unsigned long long key = live ? keys[t] : KEY_INVALID;
  // behind the image's keys the buffer is padding: KEY_INVALID like the unmerged buffer's (gather_kernel looks one key ahead)
  if (blockIdx.x == 0) for (int i = n_total + tid; i < npad; i += 1024) out[i] = KEY_INVALID;
  __syncthreads();
  if (!live) return;
  const int s = t / seg_len, j = t - s * seg_len;
  int ng[MG_SEGS], cnt[MG_SEGS];
#pragma unroll
  for (int g = 0; g < MG_SEGS; ++g) {
    const int r = n_total - g * seg_len;
    ng[g] = r < seg_len ? (r > 0 ? r : 0) : seg_len;
    cnt[g] = (ng[g] + MG_SP - 1) / MG_SP;
  }
  // splitters before the key, all segments side by side: MG_SEGS independent chains of LDS reads per step (ties -- the KEY_INVALID tail --
  // go to the lower segment first)
  int p[MG_SEGS];
#pragma unroll
  for (int g = 0; g < MG_SEGS; ++g) p[g] = 0;
#pragma unroll
  for (int bit = MG_MAXSEG / MG_SP; bit >= 1; bit >>= 1) {
#pragma unroll
    for (int g = 0; g < MG_SEGS; ++g) {
      const int q = p[g] + bit;
      if (q <= cnt[g]) {
        const unsigned long long v = spl[g][q - 1];
        if (g < s ? v <= key : v < key) p[g] = q;
      }
    }
  }
  // keys [0, (p - 1) MG_SP] of the segment are before it; the MG_SP - 1 behind that splitter may be: their loads go out together
  unsigned long long w[MG_SEGS][MG_SP - 1];
#pragma unroll
  for (int g = 0; g < MG_SEGS; ++g) {
    const int c = (p[g] - 1) * MG_SP + 1;
#pragma unroll
    for (int q = 0; q < MG_SP - 1; ++q) w[g][q] = (g != s && p[g] > 0 && c + q < ng[g]) ? keys[(size_t)g * seg_len + c + q] : KEY_INVALID;
  }
  int rank = j;
#pragma unroll
  for (int g = 0; g < MG_SEGS; ++g) {
    if (g == s || p[g] == 0) continue;
    const int c = (p[g] - 1) * MG_SP + 1;
    rank += c;
#pragma unroll
    for (int q = 0; q < MG_SP - 1; ++q) rank += (c + q < ng[g] && (g < s ? w[g][q] <= key : w[g][q] < key)) ? 1 : 0;
  }
  out[rank] = key;
}

bool sort_is_segmented(int n_img, int per_img) {
  const int seg = ((per_img + MG_SEGS - 1) / MG_SEGS + 63) & ~63;
  return n_img <= NMS_MW_MAX_BATCH && per_img > 4096 && seg <= MG_MAXSEG;
}

// sorted keys of every image: in `keys` on return, or -- *in_tmp = 1 -- in `tmp` (the merged form of small batches)
int launch_sort_keys(unsigned long long* keys, unsigned long long* tmp, int n_img, int npad, int per_img, hipStream_t s, int* in_tmp) {
  if (!keys || !tmp) return fail(CTPN_ERR_ARG, "sort: null buffer");
  if (in_tmp) *in_tmp = 0;
  const int seg = ((per_img + MG_SEGS - 1) / MG_SEGS + 63) & ~63;
  if (in_tmp && sort_is_segmented(n_img, per_img)) {
    static bool raised[CTPN_MAX_DEV] = {false};
    int dev = 0;
    CTPN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= CTPN_MAX_DEV) return fail(CTPN_ERR_ARG, "sort: device index out of range");
    const int lds = 2 * MG_MAXSEG * (int)sizeof(unsigned long long);
    int rc = raise_dynamic_lds((const void*)radix_sort_kernel<true>, lds, raised, dev);
    if (rc) return rc;
    hipLaunchKernelGGL(radix_sort_kernel<true>, dim3(MG_SEGS, n_img), dim3(RS_WAVES * 64), lds, s, keys, tmp, npad, per_img, seg);
    hipLaunchKernelGGL(merge_rank_kernel, dim3((per_img + 1023) / 1024, n_img), dim3(1024), 0, s, keys, tmp, npad, per_img, seg);
    *in_tmp = 1;
  } else {
    hipLaunchKernelGGL(radix_sort_kernel<false>, dim3(1, n_img), dim3(RS_WAVES * 64), 0, s, keys, tmp, npad, per_img, per_img);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("radix sort launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// gather the top-`topn` boxes of each image in sorted order; valid keys form a prefix
__global__ __launch_bounds__(256) void gather_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ boxes4,
                                                     float* __restrict__ sorted_boxes, float* __restrict__ sorted_scores,
                                                     int* __restrict__ sorted_anchor, int* __restrict__ valid_counts, int npad,
                                                     int per_img, int topn, unsigned char* __restrict__ colid, int ncols) {
  const int img = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= topn) return;
  const unsigned long long* k = keys + (long long)img * npad;
  const unsigned long long key = (i < npad) ? k[i] : KEY_INVALID;
  const bool valid = key != KEY_INVALID;
  if (valid) {
    const unsigned int idx = (unsigned int)(key & 0xffffffffu);
    const float4 b = *(const float4*)(boxes4 + ((long long)img * per_img + idx) * 4);
    *(float4*)(sorted_boxes + ((long long)img * topn + i) * 4) = b;
    sorted_scores[(long long)img * topn + i] = score_from_order_bits(~(unsigned int)(key >> 32));
    if (sorted_anchor) sorted_anchor[(long long)img * topn + i] = (int)idx;
    // the box's column group for the multi-workgroup NMS (nms_column_groups_kernel): the same function of x1 its siblings evaluate
    if (colid) colid[(size_t)img * ((topn + 15) & ~15) + i] = (unsigned char)nms_col_of(b.x, 1.0f, ncols);
    const bool next_valid = (i + 1 < topn) && (i + 1 < npad) && (k[i + 1] != KEY_INVALID);
    if (!next_valid) valid_counts[img] = i + 1;
  } else if (i == 0) {
    valid_counts[img] = 0;
  }
}

int launch_gather_sorted(const unsigned long long* keys, const float* boxes4, float* sorted_boxes, float* sorted_scores,
                         int* sorted_anchor, int* valid_counts, int n_img, int npad, int n_anchors_total, int topn, hipStream_t s,
                         unsigned char* colid, int ncols) {
  if (colid && (ncols < 1 || ncols > NC_MAXCOL)) return fail(CTPN_ERR_ARG, "gather: column ids need 1..256 columns");
  dim3 grid((topn + 255) / 256, n_img);
  hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, s, keys, boxes4, sorted_boxes, sorted_scores, sorted_anchor, valid_counts, npad,
                     n_anchors_total, topn, colid, ncols);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("gather launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
// greedy NMS, one workgroup (16 waves) per image, no N x N mask in HBM.
//
// Candidates are consumed in score order, 64 at a time (one per lane, replicated in every wave):
//   A. wave w tests the 64 candidates against kept boxes w, w+16, ... (kept list cached in LDS, spill in HBM);
//      __ballot turns "suppressed by an earlier keep" into one 64-bit word per wave;
//   B. wave w also builds rows 4w..4w+3 of the 64x64 intra-block suppression bitmask with __ballot
//      (row i = which later candidates box i would suppress) into LDS;
//   C. wave 0 ORs the 16 words, then walks the 64 candidates in order over the LDS bitmask rows
//      (wave-uniform 64-bit ALU), appends survivors to the kept list, and stops at max_keep.
// Work is sum_blocks 64*(K/16 + 4) IoUs per wave instead of N^2/2, and only keep[] / rois leave the chip.
// The predicate and its fp32 evaluation order are those of devIoU (reference nms_kernel.cu:24-32, :71).
// ---------------------------------------------------------------------------------------------
constexpr int NMS_WAVES = 16;
constexpr int NMS_KCAP = 2048;

__device__ __forceinline__ bool iou_gt(const float4& a, float sa, const float4& b, float sb, float thr) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
  const float inter = width * height;
  return inter / (sa + sb - inter) > thr;
}

__global__ __launch_bounds__(NMS_WAVES * 64) void nms_kernel(const float* __restrict__ sorted_boxes,
                                                             const float* __restrict__ sorted_scores,
                                                             const int* __restrict__ counts_in, int stride, float thr,
                                                             int max_keep, int* __restrict__ keep_idx, int keep_stride,
                                                             int* __restrict__ keep_counts, float* __restrict__ rois_out,
                                                             float4* __restrict__ kept_spill, const int* __restrict__ sorted_anchor,
                                                             int* __restrict__ roi_anchor) {
  __shared__ float4 s_kept[NMS_KCAP];
  __shared__ float s_area[NMS_KCAP];
  __shared__ unsigned long long s_supp[NMS_WAVES];
  __shared__ unsigned long long s_intra[64];
  __shared__ int s_K;

  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = counts_in[img] < stride ? counts_in[img] : stride;
  const float4* boxes = (const float4*)sorted_boxes + (long long)img * stride;
  float4* spill = kept_spill + (long long)img * stride;
  int* keep = keep_idx + (long long)img * keep_stride;
  if (tid == 0) s_K = 0;
  __syncthreads();
  int K = 0;
  const int cap = max_keep < keep_stride ? max_keep : keep_stride;

  // the chunk loop is a serial dependency chain: the candidate boxes (and, where rois are produced, their scores) of chunk
  // c+1 are fetched while chunk c is being resolved, so no global-load latency sits between two chunks
  const float* scs = sorted_scores ? sorted_scores + (long long)img * stride : nullptr;
  float4 nbx = make_float4(0.f, 0.f, 0.f, 0.f);
  float nsc = 0.f;
  if (lane < N) { nbx = boxes[lane]; if (rois_out && scs) nsc = scs[lane]; }
  for (int cb = 0; cb < N && K < cap; cb += 64) {
    const int ci = cb + lane;
    const bool valid = ci < N;
    const float4 bx = nbx;
    const float sc = nsc;
    {
      const int ni = ci + 64;
      nbx = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ni < N) { nbx = boxes[ni]; if (rois_out && scs) nsc = scs[ni]; }
    }
    const float ar = (bx.z - bx.x + 1.f) * (bx.w - bx.y + 1.f);

    // A: against the kept list, strided over waves
    bool supp = false;
    for (int k = wave; k < K; k += NMS_WAVES) {
      float4 kb; float ka;
      if (k < NMS_KCAP) { kb = s_kept[k]; ka = s_area[k]; }
      else { kb = spill[k]; ka = (kb.z - kb.x + 1.f) * (kb.w - kb.y + 1.f); }
      supp = supp || iou_gt(kb, ka, bx, ar, thr);
    }
    const unsigned long long sw = __ballot(supp && valid);
    if (lane == 0) s_supp[wave] = sw;

    // B: rows 4*wave .. 4*wave+3 of the intra-block mask
#pragma unroll
    for (int q = 0; q < 64 / NMS_WAVES; ++q) {
      const int i = wave * (64 / NMS_WAVES) + q;
      float4 bi;
      bi.x = __shfl(bx.x, i); bi.y = __shfl(bx.y, i); bi.z = __shfl(bx.z, i); bi.w = __shfl(bx.w, i);
      const float ai = __shfl(ar, i);
      const bool ov = valid && (lane > i) && (cb + i < N) && iou_gt(bi, ai, bx, ar, thr);
      const unsigned long long wv = __ballot(ov);
      if (lane == 0) s_intra[i] = wv;
    }
    __syncthreads();

    // C: resolve (wave 0)
    if (wave == 0) {
      unsigned long long dead = 0;
#pragma unroll
      for (int w = 0; w < NMS_WAVES; ++w) dead |= s_supp[w];
      const unsigned long long vmask = __ballot(valid);
      unsigned long long alive = vmask & ~dead;
      // Greedy resolution inside the 64: only rows of candidates that are still alive matter, in ascending order. The rows sit
      // one per lane in registers and are fetched with v_readlane (a 64-step loop of dependent LDS reads was ~6k cycles per
      // chunk and most of the kernel's time).
      const unsigned long long myrow = s_intra[lane];
      unsigned long long rem = alive;
      while (rem) {
        const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rem));
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myrow & 0xffffffffull), i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myrow >> 32), i);
        alive &= ~(((unsigned long long)hi << 32) | lo);
        rem = alive & ~((2ull << i) - 1ull);          // still-alive candidates after i
      }
      const bool mine = (alive >> lane) & 1ull;
      const int pos = K + __popcll(alive & ((1ull << lane) - 1ull));
      if (mine && pos < cap) {
        if (pos < NMS_KCAP) { s_kept[pos] = bx; s_area[pos] = ar; }
        else spill[pos] = bx;
        keep[pos] = ci;
        if (rois_out) {
          float* r = rois_out + ((long long)img * max_keep + pos) * 5;
          r[0] = sc;
          r[1] = bx.x; r[2] = bx.y; r[3] = bx.z; r[4] = bx.w;
          // which anchor (y, x, a) produced this roi: the second return of proposal_layer (bbox_deltas[order][keep], :133-157)
          if (roi_anchor) roi_anchor[(long long)img * max_keep + pos] = sorted_anchor[(long long)img * stride + ci];
        }
      }
      int Kn = K + __popcll(alive);
      if (Kn > cap) Kn = cap;
      if (lane == 0) s_K = Kn;
    }
    __syncthreads();
    K = s_K;
  }
  if (tid == 0) keep_counts[img] = K;
}

// ---------------------------------------------------------------------------------------------
// Column-decomposed greedy NMS for the proposal layer (same inputs / outputs as nms_kernel, bit-identical result).
//
// CTPN's anchors are all 16 px wide on a 16 px grid and bbox_transform_inv ignores dx / dw (reference
// lib/fast_rcnn/bbox_transform.py:50,52), so every decoded box spans x in [16 c, 16 c + 16] (clipped): boxes of non-adjacent
// columns are disjoint and adjacent columns share ONE pixel column -- IoU <= 1/33 (one column of two 17-wide boxes) whatever the heights. For a threshold above
// that, "suppressed by an earlier kept box" can only ever come from the candidate's own column group (int(x1) >> 4), i.e. greedy
// NMS over the score-sorted list factorises into independent greedy passes per column, and the global result (first max_keep
// survivors in score order) is their merge. nms_kernel walks the whole list 64 candidates at a time against ALL kept boxes
// behind two workgroup barriers per chunk (0.7 ms per launch on 32 CUs, 1.2 - 1.4 ms when it shares the GPU with the next
// batch's convolutions); here
//   1. the ranks are partitioned by column, order-preserving (per-wave histograms over contiguous rank segments + a digit-major
//      scan, the radix sort's scheme with the column as the digit);
//   2. each of the 16 waves takes columns w, w + 16, ...: 64 candidates at a time against the column's kept boxes (a few dozen,
//      in LDS), then the in-chunk greedy resolution over live candidates only -- no workgroup barrier inside;
//   3. survivors are bits in a rank-indexed mask; a popcount scan emits the first max_keep in rank (= score) order.
// The predicate and its fp32 evaluation order are nms_kernel's / devIoU's (reference nms_kernel.cu:24-32, :71).
// ---------------------------------------------------------------------------------------------

// WAVES x 64 threads per image. Two instantiations:
//   <16, 12288, 128>  the proposal layer's 12 000 candidates, 16 waves, column list in LDS;
//   <4, 1024, 48>     the connector's <= 1000 candidates: 256 threads, <= 64 VGPRs, ~11 KB of LDS: fits on a CU NEXT TO a persistent
//                     convolution workgroup (those hold 144 KB of LDS and 432 of a SIMD's 512 registers), so it starts at once instead of
//                     waiting for a free CU.
// col_scale != nullptr (the connector's NMS 0.2 over boxes / im_scale, detectors.py:28): the column is recovered as
// int(x1 * scale + 0.5) >> 4 -- x1 * scale is within an ulp or two of the multiple of 16 it came from.
template <int WAVES, int MAXN, int KCAP>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 8 : 1) void nms_columns_kernel(
    const float* __restrict__ sorted_boxes, const float* __restrict__ sorted_scores, const int* __restrict__ counts_in, int stride, float thr,
    int max_keep, int* __restrict__ keep_idx, int keep_stride, int* __restrict__ keep_counts, float* __restrict__ rois_out,
    float4* __restrict__ kept_spill, const int* __restrict__ sorted_anchor, int* __restrict__ roi_anchor, int ncols,
    const float* __restrict__ col_scale, int prefix, int dbg) {
  // dbg (diagnostic, option debug_nms; WRONG results -- tools/r6_pipeline_race.py --compare heads looks at the other batch's network outputs only):
  // 1 = no greedy pass (step 2 skipped), 2 = no output stores (step 3's), 4 = step 2 without its box loads (zeros), 8 = return right after step 1
  __shared__ unsigned short s_list[MAXN];
  __shared__ unsigned s_hist[WAVES][NC_MAXCOL];
  __shared__ unsigned s_colbase[NC_MAXCOL + 1];
  __shared__ unsigned s_alive[MAXN / 32];
  __shared__ float4 s_kept[WAVES][KCAP];
  __shared__ float s_karea[WAVES][KCAP];
  __shared__ unsigned s_wcount[WAVES];
  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int Nall = counts_in[img] < stride ? counts_in[img] : stride;
  Nall = Nall > MAXN ? MAXN : Nall;
  const float4* boxes = (const float4*)sorted_boxes + (long long)img * stride;
  float4* spill = kept_spill + (long long)img * stride;
  unsigned short* list = s_list;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const float cs = col_scale ? col_scale[img * 3 + 2] : 1.0f;
  auto col_of = [&](float x1) { int c = (int)(x1 * cs + 0.5f) >> 4; return c < 0 ? 0 : (c > ncols - 1 ? ncols - 1 : c); };
  const int cap = max_keep < keep_stride ? max_keep : keep_stride;

  // PREFIX PASS (round 6). The output is the first `cap` survivors in rank (= score) order, and whether rank r survives depends on ranks
  // below r only: the greedy pass over the first P ranks yields exactly the survivors among them. With 12 000 candidates and cap = 1000
  // the 1000th survivor of the benchmark images sits at rank ~2400 (tools/nms_prefix_stats.py) -- four fifths of the candidates, and
  // more of the work (a column's chunk is tested against ALL its kept boxes), only decide survivors nobody asks for. So: run steps 1 - 2
  // on the first `prefix` ranks; if they hold >= cap survivors the answer is complete (keep_counts = cap either way); else run them
  // again on all N (the prefix pass then cost ~1/9 of a full one). prefix = 0 / >= N: one full pass, as before. Bit-identical by
  // construction; tests/test_gpu_parity.py::test_column_nms_variants_equal_generic_nms holds every form to the generic kernel.
  int N = (prefix > 0 && prefix < Nall) ? prefix : Nall;
  for (;;) {
  // ---- 1. ranks -> column lists, ascending rank inside a column ----
  for (int i = tid; i < WAVES * NC_MAXCOL; i += WAVES * 64) (&s_hist[0][0])[i] = 0u;
  for (int i = tid; i < MAXN / 32; i += WAVES * 64) s_alive[i] = 0u;
  __syncthreads();
  const int seg = (((N + WAVES - 1) / WAVES) + 63) & ~63;
  const int lo = wave * seg < N ? wave * seg : N;
  const int hi = lo + seg < N ? lo + seg : N;
  for (int base = lo; base < hi; base += 64) {
    const int r = base + lane;
    const bool valid = r < hi;
    const unsigned c = valid ? (unsigned)col_of(boxes[r].x) : 0u;
    const unsigned long long m = rs_match(c, valid);
    if (valid && (m & lt) == 0ull) s_hist[wave][c] += (unsigned)__popcll(m);
  }
  __syncthreads();
  for (int col = tid; col < NC_MAXCOL; col += WAVES * 64) {
    unsigned sum = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { const unsigned v = s_hist[w][col]; s_hist[w][col] = sum; sum += v; }
    s_colbase[col] = sum;
  }
  __syncthreads();
  if (wave == 0) {
    unsigned v[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = s_colbase[4 * lane + q]; sum += v[q]; }
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    unsigned run = incl - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) { s_colbase[4 * lane + q] = run; run += v[q]; }
    if (lane == 63) s_colbase[NC_MAXCOL] = run;
  }
  __syncthreads();
  for (int base = lo; base < hi; base += 64) {
    const int r = base + lane;
    const bool valid = r < hi;
    const unsigned c = valid ? (unsigned)col_of(boxes[r].x) : 0u;
    const unsigned long long m = rs_match(c, valid);
    if (valid) {
      const unsigned off = s_hist[wave][c];
      list[s_colbase[c] + off + (unsigned)__popcll(m & lt)] = (unsigned short)r;
      if ((m >> lane) == 1ull) s_hist[wave][c] = off + (unsigned)__popcll(m);
    }
  }
  __syncthreads();

  // ---- 2. greedy NMS per column, one wave per column ----
  if (dbg & 8) return;
  for (int col = wave; col < ((dbg & 1) ? 0 : ncols); col += WAVES) {
    const int start = (int)s_colbase[col], m = (int)s_colbase[col + 1] - start;
    int K = 0;
    for (int cb = 0; cb < m; cb += 64) {
      const int ci = cb + lane;
      const bool valid = ci < m;
      const int rank = valid ? (int)list[start + ci] : 0;
      const float4 bx = (valid && !(dbg & 4)) ? boxes[rank] : make_float4(0.f, 0.f, 16.f * (float)(rank & 63), 16.f);
      const float ar = (bx.z - bx.x + 1.f) * (bx.w - bx.y + 1.f);
      bool supp = false;
      for (int k = 0; k < K; ++k) {                        // against the column's kept boxes (wave-uniform loop, LDS broadcast)
        float4 kb; float ka;
        if (k < KCAP) { kb = s_kept[wave][k]; ka = s_karea[wave][k]; }
        else { kb = spill[start + k]; ka = (kb.z - kb.x + 1.f) * (kb.w - kb.y + 1.f); }
        supp = supp || iou_gt(kb, ka, bx, ar, thr);
      }
      unsigned long long alive = __ballot(valid && !supp);
      unsigned long long rem = alive;
      while (rem) {                                         // in-chunk greedy resolution over live candidates, ascending
        const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rem));
        auto rl = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i)); };
        const float4 bi = make_float4(rl(bx.x), rl(bx.y), rl(bx.z), rl(bx.w));
        const float ai = rl(ar);
        const bool ov = lane > i && ((alive >> lane) & 1ull) && iou_gt(bi, ai, bx, ar, thr);
        alive &= ~__ballot(ov);
        rem = alive & ~((2ull << i) - 1ull);
      }
      const bool mine = (alive >> lane) & 1ull;
      if (mine) {
        const int pos = K + __popcll(alive & lt);
        if (pos < KCAP) { s_kept[wave][pos] = bx; s_karea[wave][pos] = ar; }
        else spill[start + pos] = bx;                       // pos < m: inside this column's own slice of the scratch
        atomicOr(&s_alive[rank >> 5], 1u << (rank & 31));
      }
      K += __popcll(alive);
      // the spill (global) is read back by this wave only, in later chunks: make the stores visible to its own loads
      if (K > KCAP) __threadfence_block();
    }
  }
  __syncthreads();

  // ---- 3. the first max_keep survivors in rank order ----
  const int nwords = (N + 31) >> 5;
  const int wpw = ((nwords + WAVES - 1) / WAVES + 1) & ~1;    // words per wave (contiguous, even: two words = 64 ranks per step)
  const int w_lo = wave * wpw < nwords ? wave * wpw : nwords;
  const int w_hi = w_lo + wpw < nwords ? w_lo + wpw : nwords;
  {
    unsigned cnt = 0;
    for (int j = w_lo + lane; j < w_hi; j += 64) cnt += (unsigned)__popc(s_alive[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) s_wcount[wave] = cnt;
  }
  __syncthreads();
  unsigned basepos = 0, total = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) { const unsigned v = s_wcount[w]; if (w < wave) basepos += v; total += v; }
  if (N < Nall && (int)total < cap) {        // the prefix does not hold `cap` survivors (workgroup-uniform): once more, on everything
    __syncthreads();                          // every wave has read s_wcount / s_alive before step 1 clears them
    N = Nall;
    continue;
  }
  int* keep = keep_idx + (long long)img * keep_stride;
  const float* scs = sorted_scores ? sorted_scores + (long long)img * stride : nullptr;
  for (int j0 = w_lo; j0 < w_hi && (int)basepos < cap; j0 += 2) {     // 64 ranks (two words) per step, one per lane
    const unsigned wlo = s_alive[j0], whi = j0 + 1 < w_hi ? s_alive[j0 + 1] : 0u;
    const unsigned long long bits = ((unsigned long long)whi << 32) | wlo;
    if (((bits >> lane) & 1ull) && !(dbg & 2)) {
      const int pos = (int)basepos + __popcll(bits & lt);
      if (pos < cap) {
        const int rank = j0 * 32 + lane;
        keep[pos] = rank;
        if (rois_out) {
          const float4 b = boxes[rank];
          float* r = rois_out + ((long long)img * max_keep + pos) * 5;
          r[0] = scs ? scs[rank] : 0.f;
          r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
          if (roi_anchor) roi_anchor[(long long)img * max_keep + pos] = sorted_anchor[(long long)img * stride + rank];
        }
      }
    }
    basepos += (unsigned)__popcll(bits);
  }
  if (tid == 0) keep_counts[img] = (int)total < cap ? (int)total : cap;
  break;
  }  // passes
}

// ---------------------------------------------------------------------------------------------
// The same decomposition for SMALL batches (round 5; the reference's own calling convention is one image per call, ctpn/demo.py:55-68, and a
// lone image's proposal tail ran on ONE workgroup = one CU of 256: 333 us on 16 waves that take 3.5 columns each, one after the other).
// Columns are independent, so they spread over the machine: ncols / 4 workgroups of 4 waves per image, ONE COLUMN PER WAVE.
//   1. the wave collects its column's ranks, ascending, from the column id of every rank (one byte each, written by gather_kernel next to the
//      sorted box; 1024 ranks per 16-byte load and lane-step, the loads of a tile in flight together) -- or, the connector's <= 1024 boxes,
//      from the boxes themselves -- into its LDS list: no partition pass, no second launch;
//   2. greedy NMS of the column as in nms_columns_kernel (kept boxes in LDS, beyond MW_KCAP re-read from the sorted boxes by their rank;
//      the next chunk's boxes are fetched while the current one is resolved);
//   3. survivors are bits of a rank-indexed mask in HBM (device-scope atomicOr); the workgroup of the image that finishes LAST (a ticket)
//      runs the popcount scan that emits the first max_keep survivors in rank order and leaves mask and ticket zeroed for the next launch.
// A column holds at most MW_LIST candidates (the caller checks: hf * 10 <= 1024 for the proposal layer, <= 1024 boxes for the connector).
// Same predicate, same order of evaluation per column: bit-identical keep lists and rois (tests/test_gpu_parity.py compares all variants).
// ---------------------------------------------------------------------------------------------
constexpr int MW_WAVES = 4, MW_KCAP = 256, MW_LIST = 1024, MW_TILE = 4;
constexpr size_t MW_ALIVE_OFF = 0, MW_TICKET_OFF = NC_MAXN / 8;
static_assert(MW_TICKET_OFF + 4 <= NMS_MW_FLAG_OFF && NMS_MW_FLAG_OFF + 4 <= NMS_MW_OVERFLOW_OFF && NMS_MW_OVERFLOW_OFF + 4 <= NMS_MW_SCRATCH_BYTES, "per-image scratch block of the multi-workgroup NMS");

__global__ __launch_bounds__(MW_WAVES * 64) void nms_column_groups_kernel(
    const float* __restrict__ sorted_boxes, const float* __restrict__ sorted_scores, const unsigned char* __restrict__ colid, int colid_stride,
    const int* __restrict__ counts_in, int stride, float thr, int max_keep, int* __restrict__ keep_idx, int keep_stride, int* __restrict__ keep_counts,
    float* __restrict__ rois_out, const int* __restrict__ sorted_anchor, int* __restrict__ roi_anchor, int ncols, const float* __restrict__ col_scale,
    int maxn, char* __restrict__ scratch, int n_limit, int stage) {
  // PREFIX PASS (see nms_columns_kernel): stage 1 = this launch looks at the first n_limit ranks only and, if they do not hold `cap` survivors,
  // leaves the block's flag word set and writes nothing; stage 2 = the full launch that follows it in the stream and returns at once when the
  // flag is clear (the usual case: ~3 us); stage 0 = one full launch, as before.
  __shared__ unsigned short s_list[MW_WAVES][MW_LIST];
  __shared__ unsigned short s_krank[MW_WAVES][MW_LIST];
  __shared__ float4 s_kept[MW_WAVES][MW_KCAP];
  __shared__ float s_karea[MW_WAVES][MW_KCAP];
  __shared__ unsigned s_wcount[MW_WAVES];
  __shared__ unsigned s_last;
  const int img = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int N = counts_in[img] < stride ? counts_in[img] : stride;
  N = N > maxn ? maxn : N;
  const int Nfull = N;
  if (n_limit > 0 && n_limit < N) N = n_limit;
  const float4* boxes = (const float4*)sorted_boxes + (long long)img * stride;
  char* blk = scratch + (size_t)img * NMS_MW_SCRATCH_BYTES;
  if (stage == 2 && *(const volatile unsigned*)(blk + NMS_MW_FLAG_OFF) == 0u) return;        // stage 1 answered (workgroup-uniform: written before this launch began)
  unsigned* g_alive = (unsigned*)(blk + MW_ALIVE_OFF);
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned short* list = s_list[wave];

  const int col = blockIdx.x * MW_WAVES + wave;
  if (col < ncols) {
    // ---- 1. the column's ranks, ascending ----
    int m = 0;
    if (colid) {
      const uint4* cid = (const uint4*)(colid + (size_t)img * colid_stride);      // 16 ranks per lane and load, 1024 per wave-step
      const unsigned pat = (unsigned)col * 0x01010101u;
      for (int base = 0; base < N; base += 1024 * MW_TILE) {
        uint4 v[MW_TILE];
#pragma unroll
        for (int j = 0; j < MW_TILE; ++j) {
          const int r0 = base + 1024 * j + 16 * lane;
          v[j] = r0 < N ? cid[r0 >> 4] : make_uint4(~pat, ~pat, ~pat, ~pat);
        }
#pragma unroll
        for (int j = 0; j < MW_TILE; ++j) {
          const int r0 = base + 1024 * j + 16 * lane;
          if (base + 1024 * j >= N) break;                  // wave-uniform
          // bit k of mm: byte k of the lane's 16 equals the column (and its rank is a candidate)
          unsigned mm = 0;
          const unsigned w4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned x = w4[q] ^ pat;                 // zero bytes = matches
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) mm |= (((x >> (8 * bb)) & 0xffu) == 0u ? 1u : 0u) << (4 * q + bb);
          }
          const int left = N - r0;                          // ranks of this lane that exist
          mm = left >= 16 ? mm : (left > 0 ? mm & ((1u << left) - 1u) : 0u);
          const int cnt = __popc(mm);
          int incl = cnt;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
          }
          int pos = m + incl - cnt;
          while (mm) {
            const int k = __builtin_ctz(mm);
            mm &= mm - 1u;
            if (pos < MW_LIST) list[pos] = (unsigned short)(r0 + k);
            ++pos;
          }
          m += __builtin_amdgcn_readlane(incl, 63);
        }
      }
    } else {
      const float cs = col_scale ? col_scale[img * 3 + 2] : 1.0f;
      for (int base = 0; base < N; base += 64 * 8) {
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int r = base + 64 * j + lane; xs[j] = r < N ? boxes[r].x : 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = base + 64 * j + lane;
          const bool hit = r < N && nms_col_of(xs[j], cs, ncols) == col;
          const unsigned long long bal = __ballot(hit);
          const int pos = m + __popcll(bal & lt);
          if (hit && pos < MW_LIST) list[pos] = (unsigned short)r;
          m += __popcll(bal);
        }
      }
    }
    // a column with more candidates than the list holds would lose the rest silently: the callers' preconditions exclude it (hf x 10 <= 1024
    // and no box clipped onto a neighbour's column: enqueue_proposals), and a caller that breaks them finds this STICKY word set (never cleared
    // by the kernel; option nms_check reads it, ctpn_api.hip)
    if (m > MW_LIST && lane == 0) __hip_atomic_fetch_or((unsigned*)(blk + NMS_MW_OVERFLOW_OFF), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    m = m > MW_LIST ? MW_LIST : m;

    // ---- 2. greedy NMS of the column ----
    int K = 0;
    // software pipeline: rank and box of the next chunk are in flight while this one is resolved
    int nrank = lane < m ? (int)list[lane] : 0;
    float4 nbx = lane < m ? boxes[nrank] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cb = 0; cb < m; cb += 64) {
      const int ci = cb + lane;
      const bool valid = ci < m;
      const int rank = nrank;
      const float4 bx = nbx;
      if (ci + 64 < m) { nrank = (int)list[ci + 64]; nbx = boxes[nrank]; }
      const float ar = (bx.z - bx.x + 1.f) * (bx.w - bx.y + 1.f);
      bool supp = false;
      for (int k = 0; k < K; ++k) {                        // against the column's kept boxes (wave-uniform loop, LDS broadcast)
        float4 kb; float ka;
        if (k < MW_KCAP) { kb = s_kept[wave][k]; ka = s_karea[wave][k]; }
        else { kb = boxes[s_krank[wave][k]]; ka = (kb.z - kb.x + 1.f) * (kb.w - kb.y + 1.f); }
        supp = supp || iou_gt(kb, ka, bx, ar, thr);
      }
      unsigned long long alive = __ballot(valid && !supp);
      // in-chunk greedy resolution over live candidates, ascending. (Round 5 also measured the two-part form -- the suppression rows of
      // every candidate that passed the kept list first, lane i parking row i, then a walk that only reads rows: 100 us against 92 for
      // the proposal layer's launch, 15 against 11 for the connector's: the rows of candidates that die inside the chunk are wasted work,
      // and they outnumber what the shorter dependent chain saves.)
      unsigned long long rem = alive;
      while (rem) {
        const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rem));
        auto rl = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i)); };
        const float4 bi = make_float4(rl(bx.x), rl(bx.y), rl(bx.z), rl(bx.w));
        const float ai = rl(ar);
        const bool ov = lane > i && ((alive >> lane) & 1ull) && iou_gt(bi, ai, bx, ar, thr);
        alive &= ~__ballot(ov);
        rem = alive & ~((2ull << i) - 1ull);
      }
      const bool mine = (alive >> lane) & 1ull;
      if (mine) {
        const int pos = K + __popcll(alive & lt);
        if (pos < MW_KCAP) { s_kept[wave][pos] = bx; s_karea[wave][pos] = ar; }
        s_krank[wave][pos] = (unsigned short)rank;          // pos < m <= MW_LIST
        __hip_atomic_fetch_or(&g_alive[rank >> 5], 1u << (rank & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      K += __popcll(alive);
    }
  }
  // ---- the last workgroup of the image to get here merges ----
  __threadfence();                                          // this workgroup's mask bits are visible device-wide before its ticket is
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add((unsigned*)(blk + MW_TICKET_OFF), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();

  // ---- 3. the first max_keep survivors in rank order; mask and ticket back to zero ----
  auto alive_word = [&](int j) { return __hip_atomic_load(&g_alive[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const int cap = max_keep < keep_stride ? max_keep : keep_stride;
  const int nwords = (N + 31) >> 5;
  const int wpw = ((nwords + MW_WAVES - 1) / MW_WAVES + 1) & ~1;    // words per wave (contiguous, even: two words = 64 ranks per step)
  const int w_lo = wave * wpw < nwords ? wave * wpw : nwords;
  const int w_hi = w_lo + wpw < nwords ? w_lo + wpw : nwords;
  {
    unsigned cnt = 0;
    for (int j = w_lo + lane; j < w_hi; j += 64) cnt += (unsigned)__popc(alive_word(j));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) s_wcount[wave] = cnt;
  }
  __syncthreads();
  unsigned basepos = 0, total = 0;
#pragma unroll
  for (int w = 0; w < MW_WAVES; ++w) { const unsigned v = s_wcount[w]; if (w < wave) basepos += v; total += v; }
  if (stage == 1) {
    const bool again = N < Nfull && (int)total < cap;      // the prefix does not hold `cap` survivors: the full launch behind this one does the work
    if (again) {
      __syncthreads();
      for (int j = tid; j < nwords; j += MW_WAVES * 64) g_alive[j] = 0u;
      if (tid == 0) { *(unsigned*)(blk + MW_TICKET_OFF) = 0u; *(unsigned*)(blk + NMS_MW_FLAG_OFF) = 1u; }
      return;
    }
    if (tid == 0) *(unsigned*)(blk + NMS_MW_FLAG_OFF) = 0u;
  } else if (stage == 2 && tid == 0) *(unsigned*)(blk + NMS_MW_FLAG_OFF) = 0u;
  int* keep = keep_idx + (long long)img * keep_stride;
  const float* scs = sorted_scores ? sorted_scores + (long long)img * stride : nullptr;
  for (int j0 = w_lo; j0 < w_hi && (int)basepos < cap; j0 += 2) {     // 64 ranks (two words) per step, one per lane
    const unsigned wlo = alive_word(j0), whi = j0 + 1 < w_hi ? alive_word(j0 + 1) : 0u;
    const unsigned long long bits = ((unsigned long long)whi << 32) | wlo;
    if ((bits >> lane) & 1ull) {
      const int pos = (int)basepos + __popcll(bits & lt);
      if (pos < cap) {
        const int rank = j0 * 32 + lane;
        keep[pos] = rank;
        if (rois_out) {
          const float4 b = boxes[rank];
          float* r = rois_out + ((long long)img * max_keep + pos) * 5;
          r[0] = scs ? scs[rank] : 0.f;
          r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
          if (roi_anchor) roi_anchor[(long long)img * max_keep + pos] = sorted_anchor[(long long)img * stride + rank];
        }
      }
    }
    basepos += (unsigned)__popcll(bits);
  }
  if (tid == 0) keep_counts[img] = (int)total < cap ? (int)total : cap;
  __syncthreads();                                          // every wave has read its words
  for (int j = tid; j < nwords; j += MW_WAVES * 64) g_alive[j] = 0u;
  if (tid == 0) *(unsigned*)(blk + MW_TICKET_OFF) = 0u;
}

// col_scale (im_info rows [h, w, scale], nullable) selects the connector's variant (stride <= 1024, 4 waves).
// PRECONDITION: boxes on the 16-px anchor grid (common.h); arbitrary boxes must go through launch_nms.
int g_debug_nms = 0;        // diagnostic (option debug_nms): process-wide on purpose -- a measurement switch, set right before the launch it applies to
int launch_nms_columns(const float* sorted_boxes, const float* sorted_scores, const int* counts_in, int stride, float thresh, int max_keep,
                       int* keep_idx, int keep_stride, int* keep_counts, float* rois_out, float* kept_spill, int n_img, int ncols, hipStream_t s,
                       const int* sorted_anchor, int* roi_anchor, const float* col_scale, void* mw_scratch, const unsigned char* colid, int prefix) {
  if (!kept_spill) return fail(CTPN_ERR_ARG, "nms: spill buffer (n_img x stride x 4 floats) required");
  if (ncols < 1 || ncols > NC_MAXCOL || stride > NC_MAXN || !(thresh >= 0.1f)) return fail(CTPN_ERR_ARG, "nms_columns: outside the column decomposition's domain");
  if (roi_anchor && (!sorted_anchor || !rois_out)) return fail(CTPN_ERR_ARG, "nms: roi_anchor needs sorted_anchor and rois_out");
  if (col_scale && stride > NC_TL_MAXN) return fail(CTPN_ERR_ARG, "nms_columns: connector variant takes at most 1024 candidates per image");
  if (mw_scratch) {
    // small batches: one column per wave, ncols / 4 workgroups per image (mw_scratch: n_img x NMS_MW_SCRATCH_BYTES, zero on entry and on exit;
    // colid: gather_kernel's column byte per rank, row pitch = stride rounded up to 16 -- null: the columns come from the boxes, <= 1024 of them)
    if (!colid && stride > MW_LIST) return fail(CTPN_ERR_ARG, "nms_columns: the multi-workgroup form needs column ids for more than 1024 candidates");
    const int maxn = col_scale ? NC_TL_MAXN : NC_MAXN;
    const bool two = prefix > 0 && prefix < stride && max_keep < prefix;       // a prefix launch, then the full one that usually finds nothing to do
    for (int stage = two ? 1 : 0; stage <= (two ? 2 : 0); ++stage)
      hipLaunchKernelGGL(nms_column_groups_kernel, dim3((ncols + MW_WAVES - 1) / MW_WAVES, n_img), dim3(MW_WAVES * 64), 0, s, sorted_boxes, sorted_scores,
                         colid, (stride + 15) & ~15, counts_in, stride, thresh, max_keep, keep_idx, keep_stride, keep_counts, rois_out, sorted_anchor, roi_anchor,
                         ncols, col_scale, maxn, (char*)mw_scratch, stage == 1 ? prefix : 0, stage);
  } else if (col_scale) {
    if (stride > NC_TL_MAXN) return fail(CTPN_ERR_ARG, "nms_columns: connector variant takes at most 1024 candidates per image");
    hipLaunchKernelGGL((nms_columns_kernel<4, NC_TL_MAXN, 48>), dim3(n_img), dim3(256), 0, s, sorted_boxes, sorted_scores, counts_in, stride, thresh,
                       max_keep, keep_idx, keep_stride, keep_counts, rois_out, (float4*)kept_spill, sorted_anchor, roi_anchor, ncols, col_scale, 0, 0);
  } else {
    hipLaunchKernelGGL((nms_columns_kernel<16, NC_MAXN, 128>), dim3(n_img), dim3(1024), 0, s, sorted_boxes, sorted_scores, counts_in, stride, thresh,
                       max_keep, keep_idx, keep_stride, keep_counts, rois_out, (float4*)kept_spill, sorted_anchor, roi_anchor, ncols, nullptr,
                       (prefix > 0 && max_keep < prefix) ? prefix : 0, g_debug_nms);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("nms_columns launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}
// DIAGNOSTIC (option debug_hog; round 6, the cross-batch interference of the split path): a kernel with the one-workgroup NMS's FOOTPRINT --
// 1024 threads and 84 KB of LDS per workgroup, one workgroup per image -- that touches no global memory but one word: it spins `usec` on
// s_memrealtime (100 MHz). Whether such a kernel beside the persistent split layers is enough to change their output, or whether it takes
// the NMS kernel's own memory traffic, is what tools/r6_pipeline_race.py asks with it.
__global__ __launch_bounds__(1024, 1) void hog_kernel(unsigned* __restrict__ sink, int usec, int touch, const uint4* __restrict__ src, unsigned n16) {
  __shared__ unsigned s_fill[84 * 256];
  __shared__ unsigned long long s_t0;
  for (int i = threadIdx.x; i < 84 * 256; i += 1024) s_fill[i] = (unsigned)i;
  if (threadIdx.x == 0) s_t0 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  const unsigned long long t0 = s_t0, ticks = (unsigned long long)usec * 100ull;      // s_memrealtime counts at 100 MHz
  unsigned acc = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    acc += s_fill[(threadIdx.x * 7u + acc) % (84u * 256u)];
    if (touch & 1) __hip_atomic_fetch_add(sink + 16 + (blockIdx.x & 15), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((touch & 4) && n16) {        // memory traffic: eight random 16-byte gathers per round (the NMS kernel's access pattern, sustained)
      unsigned r = acc * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9973u + 12345u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        r = r * 1664525u + 1013904223u;
        const uint4 v = src[(r >> 4) % n16];
        acc += v.x ^ v.w;
      }
    }
    if (touch & 2) {        // keep WRITING the whole 84 KB, like the NMS kernel does with its lists, histograms and survivor mask
      for (int i = threadIdx.x; i < 84 * 256; i += 1024) s_fill[i] = acc + (unsigned)i;
      __syncthreads();
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (acc == 0xdeadbeefu) sink[blockIdx.x] = acc;
}
int launch_hog(unsigned* sink, int n_wg, int usec, int touch, hipStream_t s, const void* src, size_t src_bytes) {
  if (!sink || n_wg <= 0 || usec <= 0) return fail(CTPN_ERR_ARG, "hog: bad argument");
  const size_t n16 = src ? src_bytes / 16 : 0;
  hipLaunchKernelGGL(hog_kernel, dim3(n_wg), dim3(1024), 0, s, sink, usec, touch, (const uint4*)src, (unsigned)(n16 > 0xfffffff0ull ? 0xfffffff0ull : n16));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("hog launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

bool nms_columns_ok(int ncols, int stride, float thresh) { return ncols >= 1 && ncols <= NC_MAXCOL && stride <= NC_MAXN && thresh >= 0.1f; }
// the connector's NMS (boxes already divided by im_scale): adjacent columns overlap by one scaled pixel of 16 / scale + 1, so
// IoU <= 1 / (32 / scale + 1) <= 1/9 for scale <= 4 -- far below the 0.2 threshold
bool nms_columns_tl_ok(int ncols, int stride, float thresh, float max_scale) {
  return ncols >= 1 && ncols <= NC_MAXCOL && stride <= NC_TL_MAXN && thresh >= 0.15f && max_scale > 0.f && max_scale <= 4.0f;
}

// text-connector front end on device (reference lib/text_connector/detectors.py:21-26 + lib/fast_rcnn/test.py:57):
// rois are already in descending score order, so "score > 0.7, then sort" is the prefix of rows above the
// threshold; boxes are divided by im_scale exactly as test_ctpn does before the connector sees them.
__global__ __launch_bounds__(256) void lines_prep_kernel(const float* __restrict__ rois, const int* __restrict__ roi_counts,
                                                         const float* __restrict__ im_info, int post, float min_score,
                                                         float* __restrict__ tl_boxes, float* __restrict__ tl_scores,
                                                         int* __restrict__ tl_counts) {
  const int img = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= post) return;
  const int cnt = roi_counts[img];
  const float scale = im_info[img * 3 + 2];
  const float* r = rois + ((long long)img * post + i) * 5;
  const bool ok = i < cnt && r[0] > min_score;
  if (ok) {
    *(float4*)(tl_boxes + ((long long)img * post + i) * 4) = make_float4(r[1] / scale, r[2] / scale, r[3] / scale, r[4] / scale);
    tl_scores[(long long)img * post + i] = r[0];
    const bool next_ok = (i + 1 < cnt) && (i + 1 < post) && (r[5] > min_score);
    if (!next_ok) tl_counts[img] = i + 1;
  } else if (i == 0) {
    tl_counts[img] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// Text-line connector on the device: TextProposalGraphBuilder.build_graph
// (lib/text_connector/text_proposal_graph_builder.py:6-78), Graph.sub_graphs_connected (other.py:20-29),
// TextProposalConnector.get_text_lines H (text_proposal_connector.py:21-64) and O
// (text_proposal_connector_oriented.py:24-105), clip_boxes (other.py:7-13) and TextDetector.filter_boxes
// (detectors.py:37-49) for the proposals that survived the score filter, the sort and NMS 0.2 (rows in descending score
// order). One workgroup per image, the proposals (<= 1000) in LDS; every fp32 / fp64 rounding point is the one of the
// host restatement csrc/text_connector.cpp (the two are compared bit for bit in tests/test_gpu_parity.py).
//   * the reference's column table (proposals bucketed by int(x1), searched column by column up to 50 px away) becomes an
//     all-pairs scan per proposal that keeps the NEAREST matching column and, inside it, the first maximum score in table
//     (= index) order: n^2 <= 1e6 cheap pair tests per image instead of pointer chasing;
//   * a node has at most one out-edge; chains are followed from every root by one thread each, three passes over the chain
//     (sums in chain order: the fp32 / fp64 results do not depend on the thread count);
//   * records of BOTH modes are produced (DETECT_MODE is an argument of ctpn_detect_collect, not of the submit).
// ---------------------------------------------------------------------------------------------
constexpr int CONN_MAX = 1024;          // proposals per image held in LDS (RPN_POST_NMS_TOP_N = 1000)

__device__ __forceinline__ bool conn_meet_v_iou(const float* y1, const float* y2, const float* hh, int a, int b) {
  const float h1 = hh[a], h2 = hh[b];
  const float y0 = fmaxf(y1[b], y1[a]);
  const float y1m = fminf(y2[b], y2[a]);
  const float ov = fmaxf(0.0f, y1m - y0 + 1.0f) / fminf(h1, h2);
  const float sim = fminf(h1, h2) / fmaxf(h1, h2);
  return ov >= 0.7f && sim >= 0.7f;       // MIN_V_OVERLAPS, MIN_SIZE_SIM
}

// np.polyfit(X, Y, 1) over a chain: double least squares, coefficients rounded to fp32 (same op order as polyfit1 on the host)
template <typename FX, typename FY>
__device__ __forceinline__ void conn_polyfit1(const int* succ, int root, int len, FX fx, FY fy, float& c0, float& c1) {
  double mx = 0, my = 0;
  for (int v = root; v >= 0; v = succ[v]) { mx += fx(v); my += fy(v); }
  mx /= (double)len; my /= (double)len;
  double sxx = 0, sxy = 0;
  for (int v = root; v >= 0; v = succ[v]) {
    const double dx = fx(v) - mx;
    sxx += dx * dx;
    sxy += dx * (fy(v) - my);
  }
  const double slope = sxx > 0 ? sxy / sxx : 0.0;
  c0 = (float)slope;
  c1 = (float)(my - slope * mx);
}

__device__ __forceinline__ float conn_clampf(float v, float lo, float hi) { return fmaxf(fminf(v, hi), lo); }

// numpy's pairwise float32 sum above 128 elements: halves (the left one a multiple of 8), recursively; `leaf` consumes the next m
// chain nodes in order. D bounds the depth at compile time (m <= 128 << D).
template <int D, typename Leaf>
__device__ __forceinline__ void conn_sum_rec(int m, float& rs, float& rh, Leaf& leaf) {
  if constexpr (D > 0) {
    if (m > 128) {
      int n2 = m / 2;
      n2 -= n2 % 8;
      float s0, h0, s1, h1;
      conn_sum_rec<D - 1>(n2, s0, h0, leaf);
      conn_sum_rec<D - 1>(m - n2, s1, h1, leaf);
      rs = s0 + s1; rh = h0 + h1;
      return;
    }
  }
  leaf(m, rs, rh);
}

__global__ __launch_bounds__(256) void connect_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                      const int* __restrict__ keep, const int* __restrict__ keep_counts, int stride,
                                                      const float* __restrict__ im_info, double* __restrict__ recs, int* __restrict__ counts,
                                                      double* __restrict__ scratch, int cap) {
  __shared__ float sx1[CONN_MAX], sy1[CONN_MAX], sx2[CONN_MAX], sy2[CONN_MAX], sh[CONN_MAX], ss[CONN_MAX];
  __shared__ float spmax[CONN_MAX];
  __shared__ int ssucc[CONN_MAX];
  __shared__ unsigned char shas_in[CONN_MAX], shas_prec[CONN_MAX];
  __shared__ int sbad;
  const int img = blockIdx.x, tid = threadIdx.x;
  int n = keep_counts[img];
  n = n < 0 ? 0 : (n > CONN_MAX ? CONN_MAX : n);
  const int im_h = (int)im_info[3 * img], im_w = (int)im_info[3 * img + 1];
  if (tid == 0) sbad = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const int src = keep[(size_t)img * stride + i];
    const float4 b = *(const float4*)(boxes + ((size_t)img * stride + src) * 4);
    sx1[i] = b.x; sy1[i] = b.y; sx2[i] = b.z; sy2[i] = b.w;
    sh[i] = b.w - b.y + 1.0f;
    ss[i] = scores[(size_t)img * stride + src];
    ssucc[i] = -1; shas_in[i] = 0;
    const int col = (int)b.x;
    if (col < 0 || col >= im_w) sbad = 1;                 // the reference raises IndexError on boxes_table[int(x1)]
  }
  __syncthreads();
  int* cnt2 = counts + (size_t)img * 3;                    // lines H, lines O, status
  if (sbad) { if (tid == 0) { cnt2[0] = 0; cnt2[1] = 0; cnt2[2] = -1; } return; }

  // precursors of every node: nearest matching column to the left within 50 px, max score in it
  for (int b = tid; b < n; b += 256) {
    const int colb = (int)sx1[b];
    int lo = (int)(sx1[b] - 50.0f);
    lo = lo < 0 ? 0 : lo;
    int cbest = -1;
    float pmax = -INFINITY;
    for (int k = 0; k < n; ++k) {
      const int ck = (int)sx1[k];
      if (ck < lo || ck >= colb || ck < cbest) continue;
      if (!conn_meet_v_iou(sy1, sy2, sh, k, b)) continue;
      if (ck > cbest) { cbest = ck; pmax = ss[k]; }
      else pmax = fmaxf(pmax, ss[k]);
    }
    shas_prec[b] = cbest >= 0;
    spmax[b] = pmax;
  }
  __syncthreads();
  // successors: nearest matching column to the right within 50 px, first maximum score in index order
  for (int i = tid; i < n; i += 256) {
    const int coli = (int)sx1[i];
    const int hi = coli + 50 < im_w - 1 ? coli + 50 : im_w - 1;
    int cbest = 0x7fffffff, best = -1;
    for (int j = 0; j < n; ++j) {
      const int cj = (int)sx1[j];
      if (cj <= coli || cj > hi || cj > cbest) continue;
      if (!conn_meet_v_iou(sy1, sy2, sh, j, i)) continue;
      if (cj < cbest) { cbest = cj; best = j; }
      else if (ss[j] > ss[best]) best = j;
    }
    if (best >= 0 && shas_prec[best] && ss[i] >= spmax[best]) { ssucc[i] = best; shas_in[best] = 1; }
  }
  __syncthreads();

  // chains -> records (both modes) into per-root scratch slots; flag = passes filter_boxes
  double* scr = scratch + (size_t)img * CONN_MAX * 20;      // per root: 9 (H) + 1 (H keep) + 9 (O) + 1 (O keep)
  const float wl = (float)(im_w - 1), hl = (float)(im_h - 1);
  for (int i = tid; i < n; i += 256) {
    double* o = scr + (size_t)i * 20;
    o[9] = 0.0; o[19] = 0.0;
    if (shas_in[i] || ssucc[i] < 0) continue;
    int len = 0;
    float x0 = INFINITY, x1m = -INFINITY;
    bool same_x = true;
    for (int v = i; v >= 0; v = ssucc[v]) {
      ++len;
      x0 = fminf(x0, sx1[v]); x1m = fmaxf(x1m, sx2[v]);
      if (sx1[v] != sx1[i]) same_x = false;
    }
    // line score and mean height: numpy's float32 pairwise add.reduce over the chain in order (see np_sum_f32 in text_connector.cpp).
    // numpy splits recursively above 128 elements (left half a multiple of 8); a chain holds at most the image's <= 1000 kept
    // proposals, i.e. at most three levels (conn_sum_rec<3>: up to 1024).
    float ssum, hsum;
    {
      int v = i;
      auto leaf = [&](int m, float& rs, float& rh) {          // consumes m chain nodes starting at v
        if (m < 8) {
          rs = 0.f; rh = 0.f;
          for (int q = 0; q < m; ++q, v = ssucc[v]) { rs += ss[v]; rh += sy2[v] - sy1[v]; }
          return;
        }
        float as[8], ah[8];
#pragma unroll
        for (int j = 0; j < 8; ++j, v = ssucc[v]) { as[j] = ss[v]; ah[j] = sy2[v] - sy1[v]; }
        int q = 8;
        for (; q < m - (m % 8); q += 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j, v = ssucc[v]) { as[j] += ss[v]; ah[j] += sy2[v] - sy1[v]; }
        }
        rs = ((as[0] + as[1]) + (as[2] + as[3])) + ((as[4] + as[5]) + (as[6] + as[7]));
        rh = ((ah[0] + ah[1]) + (ah[2] + ah[3])) + ((ah[4] + ah[5]) + (ah[6] + ah[7]));
        for (; q < m; ++q, v = ssucc[v]) { rs += ss[v]; rh += sy2[v] - sy1[v]; }
      };
      conn_sum_rec<3>(len, ssum, hsum, leaf);
    }
    const float offset = (sx2[i] - sx1[i]) * 0.5f;
    const float xa = x0 + offset, xb = x1m - offset;
    float lt, rt, lb, rb;
    if (same_x) { lt = rt = sy1[i]; lb = rb = sy2[i]; }
    else {
      float c0, c1;
      conn_polyfit1(ssucc, i, len, [&](int v) { return (double)sx1[v]; }, [&](int v) { return (double)sy1[v]; }, c0, c1);
      lt = c0 * xa + c1; rt = c0 * xb + c1;
      conn_polyfit1(ssucc, i, len, [&](int v) { return (double)sx1[v]; }, [&](int v) { return (double)sy2[v]; }, c0, c1);
      lb = c0 * xa + c1; rb = c0 * xb + c1;
    }
    const float score = ssum / (float)len;
    const float top = fminf(lt, rt), bot = fmaxf(lb, rb);
    {   // H: clip_boxes (also clips the score column: reference quirk), 4-corner layout
      const float xmin = conn_clampf(x0, 0.f, wl), xmax = conn_clampf(x1m, 0.f, wl);
      const float ymin = conn_clampf(top, 0.f, hl), ymax = conn_clampf(bot, 0.f, hl);
      const float sc = conn_clampf(score, 0.f, wl);
      o[0] = xmin; o[1] = ymin; o[2] = xmax; o[3] = ymin; o[4] = xmin; o[5] = ymax; o[6] = xmax; o[7] = ymax; o[8] = sc;
    }
    {   // O: centre-line fit, height = mean(h) + 2.5, parallelogram + skew compensation, no clipping
      float k, b;
      conn_polyfit1(ssucc, i, len, [&](int v) { return (double)((sx1[v] + sx2[v]) / 2.0f); }, [&](int v) { return (double)((sy1[v] + sy2[v]) / 2.0f); }, k, b);
      const float height = hsum / (float)len + 2.5f;
      const float b1 = b - height / 2.0f, b2 = b + height / 2.0f;
      float px1 = x0, py1 = k * x0 + b1;
      float px2 = x1m, py2 = k * x1m + b1;
      float px3 = x0, py3 = k * x0 + b2;
      float px4 = x1m, py4 = k * x1m + b2;
      const float disX = px2 - px1, disY = py2 - py1;
      const float width = sqrtf(disX * disX + disY * disY);
      const float fTmp0 = py3 - py1;
      const float fTmp1 = fTmp0 * disY / width;
      const float dx = fabsf(fTmp1 * disX / width);
      const float dy = fabsf(fTmp1 * disY / width);
      if (k < 0) { px1 -= dx; py1 += dy; px4 += dx; py4 -= dy; }
      else { px2 += dx; py2 += dy; px3 -= dx; py3 -= dy; }
      o[10] = px1; o[11] = py1; o[12] = px2; o[13] = py2; o[14] = px3; o[15] = py3; o[16] = px4; o[17] = py4; o[18] = score;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {   // filter_boxes in float64
      const double* r = o + 10 * m;
      const double heights = (fabs(r[5] - r[1]) + fabs(r[7] - r[3])) / 2.0 + 1;
      const double widths = (fabs(r[2] - r[0]) + fabs(r[6] - r[4])) / 2.0 + 1;
      o[10 * m + 9] = (widths / heights > 0.5 && r[8] > 0.9 && widths > 32.0) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  __threadfence_block();
  // ordered compaction (roots in ascending index = the reference's loop order): two threads, one per mode
  if (tid < 2) {
    const int m = tid;
    int c = 0;
    double* dst = recs + ((size_t)img * 2 + m) * cap * 9;
    for (int i = 0; i < n; ++i) {
      const double* o = scr + (size_t)i * 20 + 10 * m;
      if (o[9] != 0.0) {
        if (c < cap) for (int q = 0; q < 9; ++q) dst[(size_t)c * 9 + q] = o[q];
        ++c;
      }
    }
    cnt2[m] = c;
    if (m == 0) cnt2[2] = 0;
  }
}

int launch_connect(const float* boxes, const float* scores, const int* keep, const int* keep_counts, int stride, const float* im_info,
                   double* recs, int* counts, double* scratch, int cap, int n_img, hipStream_t s) {
  if (stride > CONN_MAX) return fail(CTPN_ERR_ARG, "connect: more proposals per image than the kernel holds in LDS");
  hipLaunchKernelGGL(connect_kernel, dim3(n_img), dim3(256), 0, s, boxes, scores, keep, keep_counts, stride, im_info, recs, counts, scratch, cap);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("connect launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

int launch_lines_prep(const float* rois, const int* roi_counts, const float* im_info, int post, float min_score,
                      float* tl_boxes, float* tl_scores, int* tl_counts, int n_img, hipStream_t s) {
  dim3 grid((post + 255) / 256, n_img);
  hipLaunchKernelGGL(lines_prep_kernel, grid, dim3(256), 0, s, rois, roi_counts, im_info, post, min_score, tl_boxes, tl_scores,
                     tl_counts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("lines_prep launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

int launch_nms(const float* sorted_boxes, const float* sorted_scores, const int* counts_in, int stride, float thresh,
               int max_keep, int* keep_idx, int keep_stride, int* keep_counts, float* rois_out, float* kept_spill, int n_img,
               hipStream_t s, const int* sorted_anchor, int* roi_anchor) {
  if (!kept_spill) return fail(CTPN_ERR_ARG, "nms: spill buffer (n_img x stride x 4 floats) required");
  if (roi_anchor && (!sorted_anchor || !rois_out)) return fail(CTPN_ERR_ARG, "nms: roi_anchor needs sorted_anchor and rois_out");
  hipLaunchKernelGGL(nms_kernel, dim3(n_img), dim3(NMS_WAVES * 64), 0, s, sorted_boxes, sorted_scores, counts_in, stride, thresh,
                     max_keep, keep_idx, keep_stride, keep_counts, rois_out, (float4*)kept_spill, sorted_anchor, roi_anchor);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CTPN_ERR_HIP, std::string("nms launch: ") + hipGetErrorString(e));
  return CTPN_OK;
}

}  // namespace ctpn
