// launch_conv3x3: layer-level decisions of the tap-reuse 3x3 convolution (kernels: conv3x3_impl.h, instantiated per arithmetic type in
// conv3x3_{f32,bf16,f16,split}.hip) -- which kernel family a layer takes, what happens to the ragged pixel columns of a 2D tiling, and the
// per-device resources the kernels share.
//
// Replaces tf.nn.conv2d + bias_add + relu of Network.conv (reference lib/networks/network.py:160-183) and the Network.max_pool that
// follows it where one does (network.py:189-196; VGGnet_test.py:23,26,30,34).
#include "conv3x3_impl.h"

namespace ctpn {

// dump pages (where lanes outside the image store, so that every wave issues the same number of stores) and tile-claim counters of the
// weights-in-registers kernel, per device, shared by its bf16 and fp16 instantiations
int c3_wr_resources(int dev, hipStream_t s, char** dump_out, unsigned** claim_out) {
  static char* dump[C3_MAX_DEV] = {nullptr};
  static unsigned* claims[C3_MAX_DEV] = {nullptr};
  static unsigned ticket[C3_MAX_DEV] = {0};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!dump[dev]) CTPN_HIP_TRY(hipMalloc((void**)&dump[dev], (size_t)1024 * 4096));
  // tile-claim counters: zero when idle (the kernel re-arms them on exit). Every launch takes the next of 64 slots, so two
  // launches in flight on different streams never share one.
  if (!claims[dev]) {
    CTPN_HIP_TRY(hipMalloc((void**)&claims[dev], 64 * 64 * sizeof(unsigned)));       // 64 launch slots x (8 groups x 4 slices x 2 words)
    // zeroed ON THE LAUNCH STREAM and waited for: the ctx streams are non-blocking, so a plain hipMemset (null stream) is not ordered
    // with the launch that follows -- the first conv1_2 of a process could start claiming tiles from counters that were not zero yet (seen in
    // round 3 as a first forward that differed from the second; a one-time wait, the counters re-arm themselves afterwards)
    CTPN_HIP_TRY(hipMemsetAsync(claims[dev], 0, 64 * 64 * sizeof(unsigned), s));
    CTPN_HIP_TRY(hipStreamSynchronize(s));
  }
  *dump_out = dump[dev];
  *claim_out = claims[dev] + (size_t)(ticket[dev]++ % 64) * 64;
  return CTPN_OK;
}

// Guard of the LDS-DMA helpers' m0 contract (conv3x3_impl.h: c3_glds16_saddr declares m0 clobbered, c3_glds16_asm saves and restores it):
// every wave stages `rounds` 1-KiB tiles of `src` into LDS with BOTH forms, interleaved with ordinary LDS traffic and a wave-uniform loop
// the compiler is free to schedule around them, and copies what arrived to out_clobber / out_keep. The two must be the source bytes.
__global__ __launch_bounds__(256) void c3_lds_dma_check_kernel(const char* __restrict__ src, char* __restrict__ out_clobber,
                                                              char* __restrict__ out_keep, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  char* a = smem + wave * 2048;          // clobber form lands here
  char* b = a + 1024;                    // save / restore form here
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)wave * 2048u;
  for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
    const int tu = __builtin_amdgcn_readfirstlane(t);                 // wave-uniform by construction; pinned to an SGPR for the "s" operand
    const char* g = src + (size_t)tu * 1024;
    c3_glds16_saddr(g, (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)lds0));
    c3_glds16_asm(g + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + 1024u)));
    c3_wait_vm<0>();
    __builtin_amdgcn_sched_barrier(0);
    const uint4 va = *(const uint4*)(a + lane * 16), vb = *(const uint4*)(b + lane * 16);
    *(uint4*)(out_clobber + (size_t)t * 1024 + lane * 16) = va;
    *(uint4*)(out_keep + (size_t)t * 1024 + lane * 16) = vb;
    __builtin_amdgcn_s_waitcnt(0);       // the LDS reads above are done before the next round's DMA overwrites the tiles
  }
}

int launch_lds_dma_check(const void* src, void* out_clobber, void* out_keep, int tiles, hipStream_t s) {
  if (tiles <= 0) return fail(CTPN_ERR_ARG, "lds_dma_check: no tiles");
  const int grid = tiles < 1024 ? (tiles + 3) / 4 : 256;
  hipLaunchKernelGGL(c3_lds_dma_check_kernel, dim3(grid), dim3(256), 4 * 2048, s, (const char*)src, (char*)out_clobber, (char*)out_keep, tiles);
  CTPN_HIP_TRY(hipGetLastError());
  return CTPN_OK;
}

// conv1_2 as the launch ctpn_api.hip may hand a q-image to: the weights-in-registers kernel's pooled form without a full-resolution output
bool conv1_fusable(DType t, int n, int h, int w, int ci, int co, bool pool, bool keep_full) {
  return dtype_is_half(t) && ci == 64 && co == 64 && pool && !keep_full && n >= 1;
}

// in/out: bordered NHWC of dtype t; pool_out != nullptr fuses the 2x2/2 VALID max-pool (out may then be nullptr).
// t == SPLIT: ci / co are the layer's channel counts; pixels hold [hi(c) | lo(c)] bf16 planes, weight rows [hi | hi | lo] per tap
// (pack_transpose_split); dup_hi: the output pixel is [hi | lo | hi] (the layer that feeds the LSTM input-projection GEMM).
// q1 / q1_frags (conv1_2 of the 16-bit modes, uint8 feed, production path): conv1_1 is computed inside the launch's window stage from the
// batch's q-image (conv3x3_wr_kernel FUSE); `in` (conv1_1's map) is not touched
int launch_conv3x3(const void* in, const void* wt, const float* bias, void* out, void* pool_out, DType t, int n, int h, int w,
                   int ci, int co, int relu, hipStream_t s, int dup_hi, const void* q1, const void* q1_frags, int split_opts) {
  const int p64 = split_opts & 1;
  const int bke = (t == DType::F32) ? 32 : 64;
  if (ci <= 0 || ci % bke != 0) return fail(CTPN_ERR_ARG, "conv3x3: Ci must be a multiple of the 128-byte strip");
  const int epc = (t == DType::F32) ? 4 : 8;
  if (co % epc != 0) return fail(CTPN_ERR_ARG, "conv3x3: Co must be a multiple of a 16-byte chunk");
  if (!out && !pool_out) return fail(CTPN_ERR_ARG, "conv3x3: no output");
  if (dup_hi && t != DType::SPLIT) return fail(CTPN_ERR_ARG, "conv3x3: dup_hi is a split-precision layout");
  const bool half = dtype_is_half(t);
  Conv3 g{};
  g.in = in; g.wt = wt; g.bias = bias; g.out = out; g.pool_out = pool_out;
  g.N = n; g.H = h; g.W = w; g.Ci = ci; g.Co = co; g.relu = relu;
  g.in_pitch = ci; g.out_pitch = co; g.a_wrap = 0; g.dup_hi = 0; g.opt_p64 = p64; g.opt_small = 1;
  if (t == DType::SPLIT) {
    if (!bias) return fail(CTPN_ERR_ARG, "conv3x3 (split precision): bias required");
    g.Ci = 3 * ci; g.in_pitch = 2 * ci; g.a_wrap = 2 * ci / 64; g.out_pitch = (dup_hi ? 3 : 2) * co; g.dup_hi = dup_hi ? 1 : 0;
  }
  const bool pool = pool_out != nullptr;
  // Ragged last tile column (W = 225 = 7 * 32 + 1 wastes an eighth of the tiles on one pixel column): the 2D launch covers
  // the multiple of 32 and the few remaining columns go through the im2col kernel (igemm.hip) as a [N*H*r] x Co GEMM (fp32).
  // 16-bit modes: one or two columns beyond a multiple of 16 go through conv3x3_edge_kernel, which shares the CUs with the main
  // launch (W = 113 then tiles as 7 x 16 instead of 8 x 16); otherwise up to 8 columns beyond a multiple of 32 go through igemm.
  // Split precision (round 6): the same edge kernel over [hi | lo] planes (three K blocks per tap); no igemm strip there.
  const bool split = t == DType::SPLIT;
  const bool half_or_split = half || (split && (split_opts & 2) != 0);
  const bool can_strip = !pool && !c3_flat_ok(g, pool) && w > 32 && out;
  const bool edge = can_strip && half_or_split && bias && co % 64 == 0 && w % 16 >= 1 && w % 16 <= 2;
  // 16-bit layers with a fused pool and no full-resolution output (conv1_2: W = 900 = 28 * 32 + 4; conv2_2: 450 = 28 * 16 + 2): two or
  // four columns beyond the main launch's tile width go through the pooled form of the edge kernel (whole pooled pixels lie inside them)
  const bool wr_layer = half && ci == 64 && (co == 64 || co == 128) && bias && relu;
  const int rp = w % (wr_layer ? 32 : 16);
  // (when the full-resolution map is kept as well -- keep_acts -- the same columns also go through the plain edge kernel: both
  // forms accumulate in the same order, so the stored pool stays the exact max of the stored map and equal to the production path's)
  // NOT for conv1_2 (Ci = Co = 64 with the pool): since conv1_1 is computed inside its window stage the persistent workgroups fill 456 of a
  // SIMD's 512 registers, ONE edge wave fits next to them instead of three, and the strip -- which also needs conv1_1 for its own columns
  // first -- ended 15 us AFTER the main launch instead of inside it, its one-wave workgroups crowding onto the CUs of the previous batch's NMS
  // (1.3 ms instead of 0.54). The padded 29th tile column costs the main launch 3.5 %; the decision depends on the layer's shape only, so the
  // stored (keep_acts) form takes the same columns through the same kernel family.
  const bool conv1_2_like = wr_layer && co == 64 && pool;
  const bool edge_pool = !conv1_2_like && pool && half_or_split && bias && co % 64 == 0 && w > 64 && h >= 2 && (w & 1) == 0 && (rp == 2 || rp == 4);
  const int r = edge_pool ? rp : (edge ? w % 16 : w % 32);
  // (which columns go to a strip depends on the layer's shape only, never on the batch: the edge kernel and the main kernels sum K in
  // different orders, and a batch must reproduce its images run alone bit for bit)
  bool strip = edge_pool || (can_strip && (edge || (!split && r >= 1 && r <= 8)));
  if (strip) g.w_cover = w - r;
  if (q1) {
    if (!conv1_fusable(t, n, h, w, ci, co, pool, out != nullptr) || !q1_frags || strip) return fail(CTPN_ERR_ARG, "conv3x3: the fused conv1_1 form is conv1_2's pooled 16-bit launch");
    g.q1 = q1; g.q1_frags = q1_frags;
  }
  int rc;
  // The strip (a few dozen workgroups) runs on its own stream, forked after the previous layer and joined before the next,
  // so it shares the machine with the main launch instead of adding 35-50 us of a nearly empty GPU per layer.
  static hipStream_t sstream[16] = {nullptr};
  static hipEvent_t ev_fork[16] = {nullptr}, ev_join[16] = {nullptr};
  static std::mutex strip_mu[16];     // the helper stream and its two events are per device, shared by every ctx on it
  int dev = 0;
  std::unique_lock<std::mutex> strip_lock;
  // One or two images: the edge kernel runs in the layer's own stream, behind the main launch, in its DEEP form (conv3x3_impl.h) -- no fork,
  // no join. WHICH columns it takes is still a function of the layer's shape alone, and the deep form issues the same MFMAs in the same
  // order, so a batch and its images run alone still agree bit for bit.
  // Split precision, any batch: in the stream as well. Beside the main launch (forked, in front of it or behind it) the one-wave workgroups
  // run for as long as the main launch does -- its two waves per SIMD hold 2 x 183 .. 246 of the 512 registers, the edge waves take the places
  // of main workgroups that then start late -- and end 60 .. 270 us after it: -3 % images/s against a padded tile column. Behind the main
  // launch on an empty machine the deep form takes 30 .. 60 us per layer: +1.8 % (tools/r6_split_edge_ab.sh, one context per process).
  const bool instream = strip && (edge || edge_pool) && (n <= 2 || split);
  auto run_edge = [&](void* dst, bool pooled) -> int {
    hipStream_t es = instream ? s : sstream[dev];
    if (split) return c3_edge_split(in, wt, bias, dst, n, h, w, ci, co, relu, r, pooled, es, true, dup_hi);      // always the deep form
    return t == DType::F16 ? c3_edge_f16(in, wt, bias, dst, n, h, w, ci, co, relu, r, pooled, es, instream)
                           : c3_edge_bf16(in, wt, bias, dst, n, h, w, ci, co, relu, r, pooled, es, instream);
  };
  if (strip && !instream) {
    CTPN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return fail(CTPN_ERR_ARG, "conv3x3: device index out of range");
    strip_lock = std::unique_lock<std::mutex>(strip_mu[dev]);   // held until the join is enqueued (event waits capture the record made before them)
    if (!sstream[dev]) {
      CTPN_HIP_TRY(hipStreamCreateWithFlags(&sstream[dev], hipStreamNonBlocking));
      CTPN_HIP_TRY(hipEventCreateWithFlags(&ev_fork[dev], hipEventDisableTiming));
      CTPN_HIP_TRY(hipEventCreateWithFlags(&ev_join[dev], hipEventDisableTiming));
    }
    CTPN_HIP_TRY(hipEventRecord(ev_fork[dev], s));
    CTPN_HIP_TRY(hipStreamWaitEvent(sstream[dev], ev_fork[dev], 0));
    if (edge_pool) {
      if ((rc = run_edge(pool_out, true))) return rc;
      if (out && (rc = run_edge(out, false))) return rc;
    } else if (edge) {
      if ((rc = run_edge(out, false))) return rc;
    } else {
      IGemm ig{};
      ig.a = in; ig.wt = wt; ig.bias = bias; ig.out = out;
      ig.M = (long long)n * h * r; ig.Ci = ci; ig.ntaps = 9; ig.Co = co;
      ig.a_plain = 0; ig.H = h; ig.W = w; ig.rx0 = w - r; ig.rw = r; ig.out_bordered = 1; ig.ldc = co; ig.relu = relu;
      if ((rc = launch_igemm(ig, t, t, sstream[dev]))) return rc;
    }
    CTPN_HIP_TRY(hipEventRecord(ev_join[dev], sstream[dev]));
  }
  switch (t) {
    case DType::F32: rc = c3_run_f32(g, pool, s); break;
    case DType::BF16: rc = c3_run_bf16(g, pool, wr_layer, s); break;
    case DType::F16: rc = c3_run_f16(g, pool, wr_layer, s); break;
    default: rc = c3_run_split(g, pool, s); break;
  }
  if (rc) return rc;
  if (instream) {
    if (edge_pool) {
      if ((rc = run_edge(pool_out, true))) return rc;
      if (out && (rc = run_edge(out, false))) return rc;
    } else if ((rc = run_edge(out, false))) return rc;
  } else if (strip) CTPN_HIP_TRY(hipStreamWaitEvent(s, ev_join[dev], 0));
  return CTPN_OK;
}

}  // namespace ctpn
