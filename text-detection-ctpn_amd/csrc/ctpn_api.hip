// C ABI of libctpn_hip.so: context, weight packing, forward orchestration, proposal layer, NMS, connector.
// See include/ctpn_hip.h for the contract and the reference interfaces each entry point replaces.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include <pthread.h>
#include <sched.h>

#include <dlfcn.h>

#include "common.h"

namespace ctpn {

static thread_local std::string t_err;
void set_error(const std::string& s) { t_err = s; }
int fail(int code, const std::string& s) { t_err = s; return code; }

// ---------------------------------------------------------------------------------------------
// network description (reference lib/networks/VGGnet_test.py:20-43)
// ---------------------------------------------------------------------------------------------
struct ConvSpec { const char* name; int ci, co, level; int pool_after; };
static const ConvSpec kConvs[14] = {
    {"conv1_1", 3, 64, 0, 0},    {"conv1_2", 64, 64, 0, 1},   {"conv2_1", 64, 128, 1, 0},  {"conv2_2", 128, 128, 1, 1},
    {"conv3_1", 128, 256, 2, 0}, {"conv3_2", 256, 256, 2, 0}, {"conv3_3", 256, 256, 2, 1}, {"conv4_1", 256, 512, 3, 0},
    {"conv4_2", 512, 512, 3, 0}, {"conv4_3", 512, 512, 3, 1}, {"conv5_1", 512, 512, 4, 0}, {"conv5_2", 512, 512, 4, 0},
    {"conv5_3", 512, 512, 4, 0}, {"rpn_conv/3x3", 512, 512, 4, 0}};
static const char* kPoolNames[4] = {"pool1", "pool2", "pool3", "pool4"};

struct ManifestEntry { std::string name; int rank; int shape[4]; size_t offset; size_t count; };
static std::vector<ManifestEntry> build_manifest() {
  std::vector<ManifestEntry> m;
  size_t off = 0;
  auto add = [&](const std::string& name, int rank, int a, int b, int c, int d) {
    ManifestEntry e; e.name = name; e.rank = rank; e.shape[0] = a; e.shape[1] = b; e.shape[2] = c; e.shape[3] = d;
    e.offset = off; e.count = (size_t)a * (rank > 1 ? b : 1) * (rank > 2 ? c : 1) * (rank > 3 ? d : 1);
    off += e.count; m.push_back(e);
  };
  for (const auto& c : kConvs) {
    add(std::string(c.name) + "/weights", 4, 3, 3, c.ci, c.co);
    add(std::string(c.name) + "/biases", 1, c.co, 1, 1, 1);
  }
  add("lstm_o/bidirectional_rnn/fw/lstm_cell/kernel", 2, 640, 512, 1, 1);
  add("lstm_o/bidirectional_rnn/fw/lstm_cell/bias", 1, 512, 1, 1, 1);
  add("lstm_o/bidirectional_rnn/bw/lstm_cell/kernel", 2, 640, 512, 1, 1);
  add("lstm_o/bidirectional_rnn/bw/lstm_cell/bias", 1, 512, 1, 1, 1);
  add("lstm_o/weights", 2, 256, 512, 1, 1);
  add("lstm_o/biases", 1, 512, 1, 1, 1);
  add("rpn_bbox_pred/weights", 2, 512, 40, 1, 1);
  add("rpn_bbox_pred/biases", 1, 40, 1, 1, 1);
  add("rpn_cls_score/weights", 2, 512, 20, 1, 1);
  add("rpn_cls_score/biases", 1, 20, 1, 1, 1);
  return m;
}
static const std::vector<ManifestEntry>& manifest() {
  static const std::vector<ManifestEntry> m = build_manifest();
  return m;
}
static const ManifestEntry* find_entry(const std::string& name) {
  for (const auto& e : manifest()) if (e.name == name) return &e;
  return nullptr;
}

struct ProfRec { int kind; hipEvent_t a, b; double work; int launches = 1; };

// ---------------------------------------------------------------------------------------------
// Host worker pool of one ctx: created once, sized by ctpn_host_thread_budget (cores of the node / ranks on the node).
// Runs the per-image host part of the connector (ctpn_detect_collect) and the staging copies of pageable host images.
// Before: up to hardware_concurrency() std::threads were created and joined per collect -- 256 per step on an 8-rank node.
// ---------------------------------------------------------------------------------------------
class HostPool {
 public:
  HostPool(int nthreads, int first_cpu) : n_(nthreads < 1 ? 1 : nthreads) {
    for (int t = 1; t < n_; ++t) {
      th_.emplace_back([this] { loop(); });
      if (first_cpu >= 0) pin(th_.back().native_handle(), first_cpu + t);
    }
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int size() const { return n_; }
  // fn(i) for every i in [0, n); returns when all are done. The calling thread works too. max_par bounds the parallelism.
  void run(int n, const std::function<void(int)>& fn, int max_par = 0) {
    if (n <= 0) return;
    const int par = std::min(n, max_par > 0 ? std::min(max_par, n_) : n_);
    if (par <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::lock_guard<std::mutex> serial(run_mu_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; total_ = n; next_.store(0); done_.store(0); helpers_ = par - 1; ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return done_.load() >= total_ && active_ == 0; });
    helpers_ = 0;      // a worker that wakes up late must not join a finished job
    fn_ = nullptr;
  }

 private:
  static void pin(pthread_t h, int cpu) {
    cpu_set_t set; CPU_ZERO(&set);
    const unsigned ncpu = std::thread::hardware_concurrency();
    CPU_SET((int)(ncpu > 0 ? (unsigned)cpu % ncpu : (unsigned)cpu), &set);
    (void)pthread_setaffinity_np(h, sizeof(set), &set);
  }
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= total_) break;
      (*fn_)(i);
      done_.fetch_add(1);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (helpers_ <= 0) continue;
        --helpers_; ++active_;
      }
      work();
      { std::lock_guard<std::mutex> lk(mu_); --active_; }
      done_cv_.notify_all();
    }
  }
  const int n_;
  std::vector<std::thread> th_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int total_ = 0, helpers_ = 0, active_ = 0;
  std::atomic<int> next_{0}, done_{0};
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

static int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v && *v ? std::atoi(v) : dflt; }

}  // namespace ctpn

using namespace ctpn;

// one block for everything a submitted batch returns (device side and, mirrored, every slot's page-locked host side)
struct PackLayout { size_t tlb, tls, keep, kcnt, rois, rcnt, total; };
static PackLayout pack_layout(size_t mb, size_t post) {
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  PackLayout L; size_t o = 0;
  L.tlb = o; o = al(o + mb * post * 4 * sizeof(float));
  L.tls = o; o = al(o + mb * post * sizeof(float));
  L.keep = o; o = al(o + mb * post * sizeof(int));
  L.kcnt = o; o = al(o + mb * sizeof(int));
  L.rois = o; o = al(o + mb * post * 5 * sizeof(float));
  L.rcnt = o; o = al(o + mb * sizeof(int));
  L.total = o;
  return L;
}

struct ctpn_ctx {
  int device = 0;
  int max_batch = 0, max_h = 0, max_w = 0;
  DType prec = DType::BF16;
  int es = 2;                       // bytes per activation element (split precision: a (hi, lo) bf16 pair = 4)
  hipStream_t stream = nullptr;     // network forward
  hipStream_t stream_p = nullptr;   // proposal layer + connector front end of the asynchronous detect path
  std::vector<void*> allocs;
  // asynchronous detect: two slots of pinned host buffers + events
  struct Slot {
    // the batch's results in ONE page-locked block, laid out like the device's out_pack: a single device-to-host copy per submit
    // (six copies before: ~30 us of host calls per batch, which at batch 1 is 4 % of the step on a slow host)
    char* pack = nullptr;
    float* tlb = nullptr; float* tls = nullptr; int* keep = nullptr; int* kcnt = nullptr; float* rois = nullptr; int* rcnt = nullptr;      // views into pack
    float* im_info = nullptr;
    double* crecs = nullptr; int* ccnt = nullptr;      // device connector results: [n][2 modes][CONN_CAP][9], [n][3]
    hipEvent_t ev_heads = nullptr, ev_decoded = nullptr, ev_done = nullptr;
    int n = 0, h = 0, w = 0; bool busy = false;
  } slot[2];
  hipEvent_t ev_last_decoded = nullptr;  // decode of the most recent submit (it reads `heads`, which the next forward rewrites)
  hipEvent_t ev_last_done = nullptr;     // the whole proposal tail (stream_p) of the most recent submit
  // "tail_confine" (default 0; was 1 in split precision during round 6): forward k + 1 waits, BEHIND its conv1_1, for the proposal tail of batch k,
  // which then overlaps conv1_1 only. History: the reversed-batch test of round 6 found that a batch in flight could change another batch's bits --
  // the proposal NMS of batch k running beside the persistent conv kernels of batch k + 1 (in split precision by default timing, in bf16 as soon as
  // the NMS was delayed into conv3_x / conv4_x). This switch removed the CONDITION. The CAUSE was in the conv kernels: the last k-slice group's
  // fragment reads were in flight across the step barrier while the LDS-DMA behind it recycled the strip they read, ordered by latency only
  // (conv3x3_impl.h, INVARIANT in conv3x3_p_kernel; profiles/r06_barrier_war.txt). Fixed there; the switch stays for A/B runs.
  int tail_confine = 0;
  // asynchronous detect, option tail_overlap = 1 (opt-in): the recurrent tail of batch k (BiLSTM + heads: 0.37 ms of latency-bound kernels
  // on 148 of 256 CUs) runs on stream_p, next to conv1_1 of batch k + 1 (HBM-write-bound) instead of in front of it. Measured, round 3,
  // same box: +0.6 % images/s (3420-3425 vs 3397-3405) -- side by side the BiLSTM takes 0.51-0.73 ms instead of 0.33 and conv1_1 0.60
  // instead of 0.46, and the proposal kernels, which start 0.8 ms later, now run under conv2_x (static persistent tiles) instead of
  // conv1_2 (dynamic tile claims): the conv stack loses 0.9 points of its roofline. Off by default.
  int tail_overlap = 0;
  hipEvent_t ev_conv = nullptr;          // conv stack + lstm_pre of the batch in flight are done (stream -> stream_p)
  hipEvent_t ev_tail = nullptr;          // the tail of the most recent submit is done (stream_p -> stream: before conv1_2 rewrites what it read)
  bool tail_pending = false;

  // weights
  bool weights_loaded = false;
  float* arena = nullptr;            // fp32 copy of the flat arena
  float* w_first = nullptr;          // [27][64]
  void* w_first_frags = nullptr;     // conv1_1 as split-bf16 MFMA A fragments (bf16 mode), 12 KB
  // options (ctpn_set_option; per ctx, never read from the environment)
  int conv1_mfma = 2;                // "conv1_kernel" (16-bit modes): 2 = uint8 feed through the q-image (exact integer pixels x 16-bit weights, one MFMA term; conv1_1 inside
                                     // conv1_2's window stage where "conv1_fuse" allows), 1 = split-bf16 kernel for both feeds, 0 = VALU kernel
  // ctpn_decode_jpeg_batch: two sets of buffers (decode of batch k + 1 while batch k's forward reads its images), allocated on first use and
  // grown on demand; a grown buffer's predecessor is retired, not freed (a pointer handed out earlier stays valid until ctpn_destroy)
  struct JpegBufs {
    int16_t* coef_host = nullptr; uint16_t* qt_host = nullptr;      // page-locked: what the entropy decoders write
    int16_t* coef_dev = nullptr; uint16_t* qt_dev = nullptr; uint8_t* out_dev = nullptr;
    size_t coef_elems = 0, qt_imgs = 0, out_bytes = 0;              // capacities
    int out_n = 0, out_h = 0, out_w = 0;                            // what out_dev holds
    hipEvent_t ev_h2d = nullptr, ev_ready = nullptr, ev_consumed = nullptr;
    bool h2d_valid = false, consumed_valid = false, ready_valid = false;
  } jpeg[2];
  uint8_t* jpeg_planes = nullptr;    // component planes between the two kernels (one set: the kernels of both buffers run on stream_c in order)
  uint8_t* jpeg_raw = nullptr;       // the decoded batch at file size when a resize follows (one set, same reason)
  size_t jpeg_planes_bytes = 0, jpeg_raw_bytes = 0;
  std::vector<void*> jpeg_retired;   // device allocations replaced by larger ones
  int jpeg_flip = 0;
  bool jpeg_ready = false;
  int debug_nms = 0;                 // "debug_nms" (diagnostic, WRONG proposals): parts mask of the one-workgroup proposal NMS, see nms_columns_kernel
  int debug_hog = 0;                 // "debug_hog" (diagnostic, 0 .. 200000; see the launch in enqueue_proposals_impl for the two upper ranges): launch a kernel with the one-workgroup NMS's footprint (1024 threads, 84 KB of LDS, one
                                     // workgroup per image) that spins this many microseconds without memory traffic in front of the proposal NMS. Results are unaffected;
                                     // tools/r6_pipeline_race.py uses it to ask what about the tail disturbs the next batch's persistent split layers
  int nms_prefix = 1;                // "nms_prefix" (round 6): the column NMS of the proposal layer first looks at the 4096 best-scored candidates only; they hold the
                                     // 1000 survivors asked for unless fewer than a quarter survive (then a full pass follows). Same keep list by construction; 0 = always the full pass
  int split_edge = 1;                // "split_edge" (round 6): split precision sends ragged tile columns (W = 225 = 14 x 16 + 1, 113 = 7 x 16 + 1, 450 = 28 x 16 + 2,
                                     // 900 = 28 x 32 + 4) through conv3x3_edge_kernel's split form, like the 16-bit modes, instead of computing a padded tile column
                                     // (an eighth of conv4_1 / conv4_2). 0 = the padded column (ABI 9's arithmetic for those columns: other last bits)
  int conv_p64 = 1;                  // "conv_p64" (round 6): split precision's conv1_2 (Co = 64, no weights-in-registers kernel) through the persistent kernel's 64-channel
                                     // form: 3.56 ms instead of the non-persistent kernel's 4.53 at batch 32 (0 = that kernel, for A/B runs). It made a latent
                                     // race of the conv kernels frequent enough to find (see tail_confine)
  int conv1_fuse = 1;                // "conv1_fuse": with conv1_kernel = 2 and keep_acts = 0, compute conv1_1 inside conv1_2 (conv3x3_wr_kernel FUSE); 0 = stand-alone from the q-image (same bytes)
  void* q_img = nullptr;             // the batch's q-image (common.h), 16-bit modes only
  size_t q_img_bytes = 0;
  int lstm_split = 0;                // "lstm_split": the recurrent product on split-bf16 MFMAs (fp32-class, |d| < 3e-5, 0.32 -> 0.16 ms). Default 1 in
                                     // the 16-bit modes and, since round 6, in split precision (set in create_impl), 0 in fp32 (exact-fp32 MFMA kernel)
  int nms_check = 0;                 // "nms_check": debug -- re-run the generic NMS kernel behind the column-decomposed one and fail on a mismatch
  float* b_conv[14] = {nullptr};     // fp32 biases
  void* wt_conv[14] = {nullptr};     // packed [Co][9*Ci] T (index 0 unused)
  void* wt_x = nullptr;              // [1024][512] T (split precision: [1024][hi(512) | hi(512) | lo(512)] bf16)
  size_t wx_row_bytes = 1024;        // bytes of one wt_x row
  void* wt_xf = nullptr;             // 16-bit modes: wt_x in lstm_pre_kernel's fragment-major order
  float* b_x = nullptr;              // [1024]
  float* wh = nullptr;               // [2][128][512]
  float* wt_fc = nullptr;            // [512][256]
  float* b_fc = nullptr;
  float* wt_h = nullptr;             // [64][512]
  float* b_h = nullptr;              // [64]
  float* wt_fold = nullptr;          // [64][256]: (lstm_o FC) x (heads) folded, bf16 throughput mode only
  float* b_fold = nullptr;           // [64]

  // activations
  void* act_conv[14] = {nullptr};
  void* act_pool[4] = {nullptr};
  bool act_valid[14] = {true, true, true, true, true, true, true, true, true, true, true, true, true, true};
  size_t act_conv_bytes[14] = {0};
  size_t act_pool_bytes[4] = {0};
  uint8_t* img_dev = nullptr;        // staging of host images, buffer 0
  uint8_t* img_dev_b[2] = {nullptr, nullptr};   // ... double-buffered: batch k+1 crosses PCIe on stream_c while batch k is on the convolutions
  hipStream_t stream_c = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  bool consumed_valid[2] = {false, false};
  void* pin_stage[2] = {nullptr, nullptr};      // page-locked staging for pageable caller buffers (lazily allocated)
  size_t pin_stage_bytes[2] = {0, 0};
  hipEvent_t ev_h2d_done[2] = {nullptr, nullptr};
  bool h2d_valid[2] = {false, false};
  int img_flip = 0;
  float* xp = nullptr;      // [M5][1024]
  float* lstm_out = nullptr;  // [M5][256]
  float* fc_out = nullptr;  // [M5][512]
  float* heads = nullptr;   // [M5][64]
  float* cls_prob = nullptr;  // [M5][20]
  float* bbox_pred = nullptr; // [M5][40]
  size_t m5_max = 0;

  // proposal buffers
  int npad_max = 0, topn_max = 12000, post_max = 1000;
  unsigned long long* keys = nullptr; unsigned long long* keys_tmp = nullptr;
  float* boxes4 = nullptr;
  float* sorted_boxes = nullptr;
  float* sorted_scores = nullptr;
  int* valid_counts = nullptr;
  int* keep_idx = nullptr;
  int* keep_counts = nullptr;
  float* rois = nullptr;
  float* kept_spill = nullptr;
  int* sorted_anchor = nullptr;      // [n][12000] anchor index of every sorted row
  int* roi_anchor = nullptr;         // [n][1000]  anchor index of every roi (second return of proposal_layer)
  int last_post = 0, last_prop_n = 0;
  float* im_info_dev = nullptr;
  char* out_pack = nullptr; size_t pack_bytes = 0;      // tl_boxes, tl_scores, tl_keep, tl_keep_counts, rois, keep_counts live here (pack_layout)
  float* tl_boxes = nullptr; float* tl_scores = nullptr; int* tl_counts = nullptr;  // connector front end
  int* tl_keep = nullptr; int* tl_keep_counts = nullptr; float* tl_spill = nullptr;
  double* conn_recs = nullptr; int* conn_counts = nullptr; double* conn_scratch = nullptr;   // device connector (connect_kernel)
  int nms_columns = 1;               // "nms_columns": 1 = column-decomposed NMS (one workgroup per image; batches <= NMS_MW_MAX_BATCH: one column per
                                     // wave over ncols / 4 workgroups per image), 0 = nms_kernel (A/B), 2 / 3 = force the one-workgroup / the multi-workgroup form
  char* nms_mw_scratch = nullptr;    // NMS_MW_CAP_BATCH x NMS_MW_SCRATCH_BYTES
  bool nms_mw_dirty = false;         // the scratch may not be in its zero state (an error between launches, an option change): memset before the next use
  unsigned char* nms_colid = nullptr;  // NMS_MW_CAP_BATCH x (topn_max rounded up to 16): column group of every sorted box (gather_kernel)
  int connect_device = 0;            // "connect_device": 1 = graph build / chains / line fit on the GPU (connect_kernel), 0 = host C++
                                     // (text_connector.cpp; default: it runs on otherwise idle host cores under the next batch's convolutions,
                                     // the kernel shares the GPU with them: 11.15 vs 11.06 ms / step)
  bool proposals_done = false;
  bool postproc_only = false;        // ctpn_create_postproc: proposal / connector buffers only, no network
  std::unique_ptr<ctpn::HostPool> pool;
  int host_threads = 1;
  bool fc_valid = true;
  int keep_acts = 0;      // "keep_acts": 1 = also store the full-resolution output of pool-fused convs, keep lstm_o (layer-wise parity)
  float* cls_in = nullptr;  // staging for proposals_from_host
  float* bbox_in = nullptr;

  // last forward geometry
  int n = 0, h = 0, w = 0;
  int gn = -1, gh = -1, gw = -1;  // geometry the borders are currently zeroed for
  bool forward_done = false;

  // profiling
  bool prof = false;
  int prof_mode = 1;                 // 1: a pair of events around every stage; 2: ONE pair around the 13 conv3x3 launches of a forward only
                                     // (an event pair per launch costs ~0.2 ms of bubbles per step, which a throughput run should not pay)
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> free_events;
  double prof_ms[CTPN_KIND_COUNT] = {0};
  long long prof_n[CTPN_KIND_COUNT] = {0};
  double prof_work[CTPN_KIND_COUNT] = {0};
};

namespace ctpn {

static int dev_alloc(ctpn_ctx* c, void** p, size_t bytes, bool zero) {
  if (bytes == 0) bytes = 256;
  CTPN_HIP_TRY(hipMalloc(p, bytes));
  c->allocs.push_back(*p);
  if (zero) CTPN_HIP_TRY(hipMemsetAsync(*p, 0, bytes, c->stream));
  return CTPN_OK;
}

// 16-bit activation -> fp32 on the host (ctpn_get_tensor, ctpn_debug_conv3x3)
static inline float host_bf16_to_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
static inline float host_f16_to_f32(uint16_t b) { _Float16 h; std::memcpy(&h, &b, 2); return (float)h; }
static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u; std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline uint16_t host_f32_to_f16(float f) { const _Float16 h = (_Float16)f; uint16_t b; std::memcpy(&b, &h, 2); return b; }
static inline DType prec_dtype(int precision) {
  return precision == CTPN_PREC_FP32 ? DType::F32 : precision == CTPN_PREC_FP16 ? DType::F16 : precision == CTPN_PREC_SPLIT ? DType::SPLIT : DType::BF16;
}

static inline int lvl(int v, int level) { for (int i = 0; i < level; ++i) v /= 2; return v; }
// Slack around every activation buffer: the conv kernels fetch input windows without clamping (conv3x3.hip). Behind the last image:
// 2D tiles read up to 17 (+ 8: half items of the tail round) bordered rows + one window row past it (16 x 16 patches), flat mode's last tile a whole window
// (256 + 2 (W + 2) + 2 pixels); in front of the first: flat mode's first tile starts one bordered row + 1 pixel early.
static inline size_t act_slack_pixels(int w) { return (size_t)28 * (w + 2) + 384; }    // behind (+ 8 rows: the second half of a split tail tile)
static inline size_t act_front_pixels(int w) { return (size_t)(w + 2) + 64; }          // in front

static int debug_sync() { static const int v = env_int("CTPN_DEBUG_SYNC", 0); return v; }
// CTPN_ROCTX=1: roctx ranges (rocprofv3 --marker-trace). librocprofiler-sdk-roctx.so is dlopen'ed on first use, so the library has no
// link-time dependency on the profiler SDK: an install without it still loads, and CTPN_ROCTX=1 there is a silent no-op.
struct RoctxApi { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
static const RoctxApi& roctx_api() {
  static const RoctxApi api = [] {
    RoctxApi a;
    if (!env_int("CTPN_ROCTX", 0)) return a;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      const char* rp = std::getenv("ROCM_PATH");
      const std::string p = std::string(rp && *rp ? rp : "/opt/rocm") + "/lib/librocprofiler-sdk-roctx.so";
      h = dlopen(p.c_str(), RTLD_NOW | RTLD_GLOBAL);
    }
    if (h) {
      a.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      a.pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!a.push || !a.pop) { a.push = nullptr; a.pop = nullptr; }
    }
    return a;
  }();
  return api;
}
static int roctx_on() { return roctx_api().push != nullptr; }
static const char* kKindNames[CTPN_KIND_COUNT + 1] = {"ctpn:conv_first", "ctpn:conv_gemm", "ctpn:pool", "ctpn:gemm", "ctpn:bilstm",
                                                     "ctpn:decode", "ctpn:sort", "ctpn:nms", "ctpn:conv_stack"};
struct Timed {
  ctpn_ctx* c; int kind; double work; hipEvent_t a = nullptr, b = nullptr; bool on; hipStream_t st;
  Timed(ctpn_ctx* c_, int kind_, double work_, hipStream_t st_ = nullptr) : c(c_), kind(kind_), work(work_), on(c_->prof && (c_->prof_mode == 1 || kind_ == CTPN_KIND_COUNT)), st(st_ ? st_ : c_->stream) {
    if (debug_sync()) { fprintf(stderr, "[ctpn] launch kind %d work %.3g\n", kind, work); fflush(stderr); }
    // CTPN_ROCTX=1: a roctx range around the enqueue of every stage (rocprofv3 --marker-trace shows them next to the kernels;
    // the reference's only instrumentation is the wall-clock Timer of ctpn/demo.py:56-66)
    if (roctx_on()) (void)roctx_api().push(kKindNames[kind]);
    if (!on) return;
    auto get = [&]() { hipEvent_t e; if (!c->free_events.empty()) { e = c->free_events.back(); c->free_events.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
    a = get(); b = get();
    (void)hipEventRecord(a, st);
  }
  ~Timed() {
    if (roctx_on()) (void)roctx_api().pop();
    if (debug_sync()) { hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[ctpn]   done kind %d: %s\n", kind, hipGetErrorString(e)); fflush(stderr); }
    if (!on) return;
    (void)hipEventRecord(b, st);
    c->pending.push_back({kind, a, b, work});
  }
};

static int prof_drain(ctpn_ctx* c) {
  if (c->pending.empty()) return CTPN_OK;
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream));
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream_p));
  for (auto& r : c->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->prof_ms[r.kind] += ms; c->prof_n[r.kind] += r.launches; c->prof_work[r.kind] += r.work;
    }
    c->free_events.push_back(r.a); c->free_events.push_back(r.b);
  }
  c->pending.clear();
  return CTPN_OK;
}

// ---------------------------------------------------------------------------------------------
static int pack_weights(ctpn_ctx* c) {
  hipStream_t s = c->stream;
  const float* A = c->arena;
  int rc;
  for (int i = 0; i < 14; ++i) {
    const ManifestEntry* we = find_entry(std::string(kConvs[i].name) + "/weights");
    const ManifestEntry* be = find_entry(std::string(kConvs[i].name) + "/biases");
    CTPN_HIP_TRY(hipMemcpyAsync(c->b_conv[i], A + be->offset, be->count * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (i == 0) {
      CTPN_HIP_TRY(hipMemcpyAsync(c->w_first, A + we->offset, we->count * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
      const int K = 9 * kConvs[i].ci, Co = kConvs[i].co;
      // HWIO [K][Co] -> [Co][K] (split precision: [Co][9][hi(Ci) | hi(Ci) | lo(Ci)])
      if (c->prec == DType::SPLIT) { if ((rc = launch_pack_transpose_split(A + we->offset, Co, c->wt_conv[i], 9, kConvs[i].ci, Co, s))) return rc; }
      else if ((rc = launch_pack_transpose(A + we->offset, Co, c->wt_conv[i], K, c->prec, K, Co, s))) return rc;
    }
  }
  const char* dirs[2] = {"fw", "bw"};
  for (int d = 0; d < 2; ++d) {
    const ManifestEntry* ke = find_entry(std::string("lstm_o/bidirectional_rnn/") + dirs[d] + "/lstm_cell/kernel");
    const ManifestEntry* be = find_entry(std::string("lstm_o/bidirectional_rnn/") + dirs[d] + "/lstm_cell/bias");
    // kernel[:512] ([512 in][512 gates]) -> wt_x rows d*512.. ([gate][in]; split precision: [gate][hi | hi | lo] against [hi | lo | hi] pixels)
    char* dst = (char*)c->wt_x + (size_t)d * 512 * c->wx_row_bytes;
    if (c->prec == DType::SPLIT) { if ((rc = launch_pack_transpose_split(A + ke->offset, 512, dst, 1, 512, 512, s))) return rc; }
    else if ((rc = launch_pack_transpose(A + ke->offset, 512, dst, 512, c->prec, 512, 512, s))) return rc;
    CTPN_HIP_TRY(hipMemcpyAsync(c->wh + (size_t)d * 128 * 512, A + ke->offset + (size_t)512 * 512, (size_t)128 * 512 * sizeof(float), hipMemcpyDeviceToDevice, s));
    CTPN_HIP_TRY(hipMemcpyAsync(c->b_x + d * 512, A + be->offset, 512 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  {
    // gate columns of lstm_pre in the recurrence kernel's order (bilstm.hip: a lane's 4 gates x 4 units = one 64-byte run): permute
    // the rows of the packed [1024][512] input-projection matrix and its bias once, here
    void* tmp = nullptr;
    const size_t wbytes = (size_t)1024 * c->wx_row_bytes;
    CTPN_HIP_TRY(hipMalloc(&tmp, wbytes));
    CTPN_HIP_TRY(hipMemcpyAsync(tmp, c->wt_x, wbytes, hipMemcpyDeviceToDevice, s));
    if ((rc = launch_lstm_permute_rows(tmp, c->wt_x, (int)c->wx_row_bytes, s))) { (void)hipFree(tmp); return rc; }
    CTPN_HIP_TRY(hipMemcpyAsync(tmp, c->b_x, 1024 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if ((rc = launch_lstm_permute_rows(tmp, c->b_x, 4, s))) { (void)hipFree(tmp); return rc; }
    if (c->wt_xf && (rc = launch_lstm_pre_pack(c->wt_x, c->wt_xf, s))) { (void)hipFree(tmp); return rc; }
    CTPN_HIP_TRY(hipStreamSynchronize(s));
    CTPN_HIP_TRY(hipFree(tmp));
  }
  {
    const ManifestEntry* we = find_entry("lstm_o/weights");
    const ManifestEntry* be = find_entry("lstm_o/biases");
    if ((rc = launch_pack_transpose(A + we->offset, 512, c->wt_fc, 256, DType::F32, 256, 512, s))) return rc;
    CTPN_HIP_TRY(hipMemcpyAsync(c->b_fc, A + be->offset, 512 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  {
    const ManifestEntry* wb = find_entry("rpn_bbox_pred/weights");
    const ManifestEntry* bb = find_entry("rpn_bbox_pred/biases");
    const ManifestEntry* wc = find_entry("rpn_cls_score/weights");
    const ManifestEntry* bc = find_entry("rpn_cls_score/biases");
    CTPN_HIP_TRY(hipMemsetAsync(c->wt_h, 0, (size_t)64 * 512 * sizeof(float), s));
    CTPN_HIP_TRY(hipMemsetAsync(c->b_h, 0, 64 * sizeof(float), s));
    if ((rc = launch_pack_transpose(A + wb->offset, 40, c->wt_h, 512, DType::F32, 512, 40, s))) return rc;
    if ((rc = launch_pack_transpose(A + wc->offset, 20, c->wt_h + (size_t)40 * 512, 512, DType::F32, 512, 20, s))) return rc;
    CTPN_HIP_TRY(hipMemcpyAsync(c->b_h, A + bb->offset, 40 * sizeof(float), hipMemcpyDeviceToDevice, s));
    CTPN_HIP_TRY(hipMemcpyAsync(c->b_h + 40, A + bc->offset, 20 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  CTPN_HIP_TRY(hipStreamSynchronize(s));
  if ((rc = pack_conv1_frags(c->w_first, c->b_conv[0], (uint4*)c->w_first_frags))) return rc;
  {
    // lstm_o has no activation after its FC (reference network.py:110-113), so FC (256 -> 512) and the two heads
    // (512 -> 40 | 20) compose into one 256 -> 60 map: W' = W_fc W_h, b' = b_fc W_h + b_h, folded here in double.
    // Used by the bf16 throughput mode only; the fp32 gate keeps the reference's two-GEMM op order.
    const ManifestEntry* wf = find_entry("lstm_o/weights");
    const ManifestEntry* bf = find_entry("lstm_o/biases");
    const ManifestEntry* wb = find_entry("rpn_bbox_pred/weights");
    const ManifestEntry* bb = find_entry("rpn_bbox_pred/biases");
    const ManifestEntry* wc = find_entry("rpn_cls_score/weights");
    const ManifestEntry* bc = find_entry("rpn_cls_score/biases");
    std::vector<float> hfc(256 * 512), hbf(512), hwb(512 * 40), hbb(40), hwc(512 * 20), hbc(20);
    CTPN_HIP_TRY(hipMemcpy(hfc.data(), A + wf->offset, hfc.size() * 4, hipMemcpyDeviceToHost));
    CTPN_HIP_TRY(hipMemcpy(hbf.data(), A + bf->offset, hbf.size() * 4, hipMemcpyDeviceToHost));
    CTPN_HIP_TRY(hipMemcpy(hwb.data(), A + wb->offset, hwb.size() * 4, hipMemcpyDeviceToHost));
    CTPN_HIP_TRY(hipMemcpy(hbb.data(), A + bb->offset, hbb.size() * 4, hipMemcpyDeviceToHost));
    CTPN_HIP_TRY(hipMemcpy(hwc.data(), A + wc->offset, hwc.size() * 4, hipMemcpyDeviceToHost));
    CTPN_HIP_TRY(hipMemcpy(hbc.data(), A + bc->offset, hbc.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> fold((size_t)64 * 256, 0.f), bfold(64, 0.f);
    for (int o = 0; o < 60; ++o) {
      auto wh = [&](int j) -> double { return o < 40 ? hwb[(size_t)j * 40 + o] : hwc[(size_t)j * 20 + (o - 40)]; };
      for (int k = 0; k < 256; ++k) {
        double acc = 0;
        for (int j = 0; j < 512; ++j) acc += (double)hfc[(size_t)k * 512 + j] * wh(j);
        fold[(size_t)o * 256 + k] = (float)acc;
      }
      double accb = o < 40 ? hbb[o] : hbc[o - 40];
      for (int j = 0; j < 512; ++j) accb += (double)hbf[j] * wh(j);
      bfold[o] = (float)accb;
    }
    CTPN_HIP_TRY(hipMemcpy(c->wt_fold, fold.data(), fold.size() * 4, hipMemcpyHostToDevice));
    CTPN_HIP_TRY(hipMemcpy(c->b_fold, bfold.data(), bfold.size() * 4, hipMemcpyHostToDevice));
  }
  c->weights_loaded = true;
  return CTPN_OK;
}

// the column NMS of a small batch spreads its columns over the machine (proposal.hip: nms_column_groups_kernel); option nms_columns = 2 / 3
// pins one form for A/B runs and the tests
// hf: rows of the feature map (a column holds hf x 10 candidates at most, the kernel's list 1024), 0 for the connector's <= 1024 boxes
static inline bool nms_multi_wg(const ctpn_ctx* c, int n, int hf) {
  return c->nms_mw_scratch && hf * 10 <= 1024 && ((c->nms_columns == 3 && n <= NMS_MW_CAP_BATCH) || (c->nms_columns == 1 && n <= NMS_MW_MAX_BATCH));
}

static int enqueue_proposals_impl(ctpn_ctx* c, const float* heads, int heads_are_probs, int n, int hf, int wf, const float* im_info,
                                  int pre_nms_topn, int post_nms_topn, float nms_thresh, float min_size, hipStream_t s, hipEvent_t ev_decoded);
static int enqueue_proposals(ctpn_ctx* c, const float* heads, int heads_are_probs, int n, int hf, int wf, const float* im_info,
                             int pre_nms_topn, int post_nms_topn, float nms_thresh, float min_size, hipStream_t s = nullptr,
                             hipEvent_t ev_decoded = nullptr) {
  const int rc = enqueue_proposals_impl(c, heads, heads_are_probs, n, hf, wf, im_info, pre_nms_topn, post_nms_topn, nms_thresh, min_size, s, ev_decoded);
  if (rc != CTPN_OK) c->nms_mw_dirty = true;       // whatever failed, nobody vouches for the multi-workgroup NMS's scratch any more (common.h)
  return rc;
}
static int enqueue_proposals_impl(ctpn_ctx* c, const float* heads, int heads_are_probs, int n, int hf, int wf, const float* im_info,
                                  int pre_nms_topn, int post_nms_topn, float nms_thresh, float min_size, hipStream_t s, hipEvent_t ev_decoded) {
  if (!s) s = c->stream;
  if (!im_info) return fail(CTPN_ERR_ARG, "proposals: null pointer");
  if (pre_nms_topn <= 0 || pre_nms_topn > c->topn_max) return fail(CTPN_ERR_CAPACITY, "proposals: pre_nms_topn must be in 1..12000");
  if (post_nms_topn <= 0 || post_nms_topn > c->post_max) return fail(CTPN_ERR_CAPACITY, "proposals: post_nms_topn must be in 1..1000");
  const int per_img = hf * wf * 10;
  const int npad = next_pow2(per_img);
  if (npad > c->npad_max) return fail(CTPN_ERR_CAPACITY, "proposals: feature map larger than the ctx was created for");
  if (n > 4) CTPN_HIP_TRY(hipMemcpyAsync(c->im_info_dev, im_info, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, s));      // (<= 4: in decode_kernel's arguments)
  ProposalCfg pc{n, hf, wf, pre_nms_topn, post_nms_topn, nms_thresh, min_size};
  int rc;
  bool mw = nms_multi_wg(c, n, hf) && nms_columns_ok(wf, pre_nms_topn, nms_thresh);
  if (c->nms_mw_scratch && c->nms_mw_dirty) {
    // in stream order in front of everything that follows; both streams that ever use the block are drained by whoever set the flag
    CTPN_HIP_TRY(hipMemsetAsync(c->nms_mw_scratch, 0, (size_t)NMS_MW_CAP_BATCH * NMS_MW_SCRATCH_BYTES, s));
    c->nms_mw_dirty = false;
  }
  const bool seg_sort = !(c->nms_columns == 2 || c->nms_columns == 0);      // options 0 / 2 pin the one-workgroup forms of sort and NMS
  const double nanch = (double)n * per_img;
  {
    Timed t(c, CTPN_KIND_DECODE, nanch * (60.0 * 4 / 10 + 8 + 16), s);
    if ((rc = launch_decode(heads, 64, heads_are_probs, c->cls_in, c->bbox_in, c->im_info_dev, heads_are_probs ? nullptr : c->cls_prob,
                            heads_are_probs ? nullptr : c->bbox_pred, c->keys, c->boxes4, pc, npad, s, seg_sort && sort_is_segmented(n, per_img), n <= 4 ? im_info : nullptr))) return rc;
  }
  if (ev_decoded) CTPN_HIP_TRY(hipEventRecord(ev_decoded, s));
  {
    Timed t(c, CTPN_KIND_SORT, (double)n * npad * 16.0, s);
    int in_tmp = 0;
    if ((rc = launch_sort_keys(c->keys, c->keys_tmp, n, npad, per_img, s, seg_sort ? &in_tmp : nullptr))) return rc;
    const unsigned long long* sorted_keys = in_tmp ? c->keys_tmp : c->keys;
    // boxes whose x was clipped onto the image's last pixel column (im_info narrower than the feature map: only ctpn_proposals_from_host can
    // say so) pile up in ONE column group, which may then exceed the multi-workgroup kernel's list: those calls keep the one-workgroup form
    for (int i = 0; i < n; ++i) mw = mw && im_info[3 * i + 1] >= (float)((wf - 1) * 16 + 1);
    if ((rc = launch_gather_sorted(sorted_keys, c->boxes4, c->sorted_boxes, c->sorted_scores, c->sorted_anchor, c->valid_counts, n, npad, per_img, pre_nms_topn, s,
                                   mw ? c->nms_colid : nullptr, wf))) return rc;
  }
  if (c->debug_hog > 0 && c->nms_mw_scratch) {
    // values above 50000: the hog also keeps writing its 84 KB of LDS (usec = value - 50000); above 100000: it gathers random 16-byte pieces of the
    // largest activation buffer instead (usec = value - 100000, twice the workgroups): the NMS kernel's memory traffic for as long as asked
    const int hv = c->debug_hog;
    const int usec = hv > 100000 ? hv - 100000 : hv > 50000 ? hv - 50000 : hv, touch = hv > 100000 ? 4 : hv > 50000 ? 2 : 0;
    const void* src = nullptr; size_t src_bytes = 0;
    for (int i = 0; i < 14; ++i) if (c->act_conv[i] && c->act_conv_bytes[i] > src_bytes) { src = c->act_conv[i]; src_bytes = c->act_conv_bytes[i]; }
    if ((rc = launch_hog((unsigned*)(c->nms_colid), touch == 4 ? 2 * n : n, usec, touch, s, src, src_bytes))) return rc;       // (sink: never written; any device pointer)
  }
  {
    Timed t(c, CTPN_KIND_NMS, (double)n * pre_nms_topn * 24.0, s);
    g_debug_nms = c->debug_nms;
    if (c->nms_columns && nms_columns_ok(wf, pre_nms_topn, nms_thresh)) {
      // 16 waves per image (a 4-wave footprint that co-resides with the persistent convolutions took 1.9 ms instead of 0.66 ms and slowed
      // conv1_2 by 8 % through the shared SIMDs in round 2: removed)
      if ((rc = launch_nms_columns(c->sorted_boxes, c->sorted_scores, c->valid_counts, pre_nms_topn, nms_thresh, post_nms_topn, c->keep_idx,
                                   c->topn_max, c->keep_counts, c->rois, c->kept_spill, n, wf, s, c->sorted_anchor, c->roi_anchor, nullptr,
                                   mw ? c->nms_mw_scratch : nullptr, mw ? c->nms_colid : nullptr, c->nms_prefix ? 4096 : 0))) return rc;
      if (c->nms_check) {
        // option "nms_check" (debug; synchronises the stream): the column decomposition presumes boxes on the 16-px anchor grid (common.h). Re-run the generic
        // kernel on the same candidates and fail loudly if the keep lists differ.
        std::vector<int> k1((size_t)n * c->topn_max), c1(n), k2((size_t)n * c->topn_max), c2(n);
        int* keep2 = nullptr; int* cnt2 = nullptr; float* spill2 = nullptr;
        struct Free3 { int*& a; int*& b; float*& c; ~Free3() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (c) (void)hipFree(c); } } guard{keep2, cnt2, spill2};   // every early return frees
        CTPN_HIP_TRY(hipStreamSynchronize(s));
        CTPN_HIP_TRY(hipMemcpy(k1.data(), c->keep_idx, k1.size() * sizeof(int), hipMemcpyDeviceToHost));
        CTPN_HIP_TRY(hipMemcpy(c1.data(), c->keep_counts, c1.size() * sizeof(int), hipMemcpyDeviceToHost));
        CTPN_HIP_TRY(hipMalloc((void**)&keep2, k2.size() * sizeof(int)));
        CTPN_HIP_TRY(hipMalloc((void**)&cnt2, c2.size() * sizeof(int)));
        CTPN_HIP_TRY(hipMalloc((void**)&spill2, (size_t)n * c->topn_max * 4 * sizeof(float)));
        rc = launch_nms(c->sorted_boxes, c->sorted_scores, c->valid_counts, pre_nms_topn, nms_thresh, post_nms_topn, keep2, c->topn_max, cnt2, nullptr, spill2, n, s);
        if (rc == CTPN_OK && (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(k2.data(), keep2, k2.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(c2.data(), cnt2, c2.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess))
          rc = fail(CTPN_ERR_HIP, "nms_check: copy back failed");
        if (rc) return rc;
        if (mw) {
          // the multi-workgroup form's sticky overflow words (a column with more candidates than the kernel's list)
          for (int i = 0; i < n; ++i) {
            unsigned ov = 0;
            CTPN_HIP_TRY(hipMemcpy(&ov, c->nms_mw_scratch + (size_t)i * NMS_MW_SCRATCH_BYTES + NMS_MW_OVERFLOW_OFF, 4, hipMemcpyDeviceToHost));
            if (ov) {
              c->nms_mw_dirty = true;
              return fail(CTPN_ERR_STATE, "nms_check: a column held more candidates than the multi-workgroup NMS's list (1024): keep lists are incomplete");
            }
          }
        }
        for (int i = 0; i < n; ++i) {
          bool same = c1[i] == c2[i];
          for (int k = 0; same && k < c1[i]; ++k) same = k1[(size_t)i * c->topn_max + k] == k2[(size_t)i * c->topn_max + k];
          if (!same) return fail(CTPN_ERR_STATE, "nms_check: column-decomposed NMS differs from the generic kernel (boxes off the 16-px anchor grid?)");
        }
      }
    } else if ((rc = launch_nms(c->sorted_boxes, c->sorted_scores, c->valid_counts, pre_nms_topn, nms_thresh, post_nms_topn, c->keep_idx,
                                c->topn_max, c->keep_counts, c->rois, c->kept_spill, n, s, c->sorted_anchor, c->roi_anchor))) return rc;
  }
  c->last_post = post_nms_topn; c->last_prop_n = n;
  if (!heads_are_probs) c->proposals_done = true;
  return CTPN_OK;
}

static int run_proposals(ctpn_ctx* c, const float* heads, int heads_are_probs, int n, int hf, int wf, const float* im_info,
                         int pre_nms_topn, int post_nms_topn, float nms_thresh, float min_size, float* rois_out, int* counts_out) {
  if (!rois_out || !counts_out) return fail(CTPN_ERR_ARG, "proposals: null pointer");
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream_p));   // the asynchronous detect path shares the proposal buffers
  int rc = enqueue_proposals(c, heads, heads_are_probs, n, hf, wf, im_info, pre_nms_topn, post_nms_topn, nms_thresh, min_size);
  if (rc) return rc;
  hipStream_t s = c->stream;
  CTPN_HIP_TRY(hipMemcpyAsync(counts_out, c->keep_counts, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
  CTPN_HIP_TRY(hipMemcpyAsync(rois_out, c->rois, (size_t)n * post_nms_topn * 5 * sizeof(float), hipMemcpyDeviceToHost, s));
  CTPN_HIP_TRY(hipStreamSynchronize(s));
  return CTPN_OK;
}

}  // namespace ctpn

// =============================================================================================
extern "C" {

int ctpn_abi_version(void) { return CTPN_ABI_VERSION; }
const char* ctpn_last_error(void) { return t_err.c_str(); }
int ctpn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ctpn_weight_count(void) { return (int)manifest().size(); }
int ctpn_weight_manifest(int index, const char** name, int* rank, int shape4[4], size_t* offset_floats) {
  const auto& m = manifest();
  if (index < 0 || index >= (int)m.size()) return fail(CTPN_ERR_ARG, "manifest index out of range");
  if (name) *name = m[index].name.c_str();
  if (rank) *rank = m[index].rank;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = m[index].shape[i];
  if (offset_floats) *offset_floats = m[index].offset;
  return CTPN_OK;
}

int ctpn_host_thread_budget(int cpu_count, int local_world_size, int requested) {
  if (requested > 0) return requested > 256 ? 256 : requested;
  if (cpu_count < 1) cpu_count = 1;
  if (local_world_size < 1) local_world_size = 1;
  int b = cpu_count / local_world_size;
  return b < 1 ? 1 : (b > 32 ? 32 : b);      // 32 = images per batch of the benchmark configuration: more threads have nothing to do
}

static int create_impl(ctpn_ctx** out, int device_id, int max_batch, int max_h, int max_w, int precision, bool postproc_only) {
  if (!out) return fail(CTPN_ERR_ARG, "ctpn_create: out is null");
  *out = nullptr;
  if (max_batch <= 0 || max_h < 16 || max_w < 16) return fail(CTPN_ERR_ARG, "ctpn_create: max_batch > 0 and max_h, max_w >= 16 required");
  if (precision < CTPN_PREC_FP32 || precision > CTPN_PREC_SPLIT) return fail(CTPN_ERR_ARG, "ctpn_create: unknown precision");
  int ndev = ctpn_device_count();
  if (ndev <= 0) return fail(CTPN_ERR_NODEVICE, "ctpn_create: no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) {
    // the usual cause on a multi-GPU node: a rank whose LOCAL_RANK is not among the devices its environment lets it see
    const char* hv = getenv("HIP_VISIBLE_DEVICES");
    const char* rv = getenv("ROCR_VISIBLE_DEVICES");
    return fail(CTPN_ERR_ARG, "ctpn_create: device_id " + std::to_string(device_id) + " out of range: " + std::to_string(ndev) +
                " device(s) visible (HIP_VISIBLE_DEVICES=" + (hv ? hv : "unset") + ", ROCR_VISIBLE_DEVICES=" + (rv ? rv : "unset") + "); one process per GPU needs LOCAL_RANK < that count");
  }
  CTPN_HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  CTPN_HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(CTPN_ERR_NODEVICE, std::string("ctpn_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");

  ctpn_ctx* c = new ctpn_ctx();
  c->device = device_id; c->max_batch = max_batch; c->max_h = max_h; c->max_w = max_w;
  c->prec = prec_dtype(precision);
  c->es = dtype_bytes(c->prec);
  c->wx_row_bytes = c->prec == DType::SPLIT ? (size_t)3 * 512 * 2 : (size_t)512 * c->es;
  // 16-bit throughput modes: the recurrent product h Wh on split-bf16 MFMAs by default (state, gates, accumulation fp32; three bf16 terms per
  // product: |lstm_out - exact-fp32 kernel| < 3e-5, two orders below the modes' own conv rounding; 0.32 -> 0.16 ms per 32-image batch).
  // split precision takes the split-bf16 recurrence too since round 6: it IS this mode's arithmetic ((hi, lo) bf16 pairs, three MFMA terms, fp32
  // accumulate: what its convolutions do), the bench's accuracy object does not move (cls_prob 3.48e-5, 100 % lines either way) and a lone image
  // saves 0.2 ms of its 2.05 (343 -> 138 us at batch 32). fp32 keeps the exact kernel; option lstm_split = 0 restores it anywhere.
  c->lstm_split = (dtype_is_half(c->prec) || c->prec == DType::SPLIT) ? 1 : 0;
  c->tail_confine = 0;
  c->postproc_only = postproc_only;
  {
    // host workers: the node's cores divided by the ranks that share it (torchrun exports LOCAL_WORLD_SIZE), CTPN_HOST_THREADS
    // overrides; CTPN_AFFINITY=1 pins them to the block of cores [local_rank * budget, ...)
    const unsigned hw = std::thread::hardware_concurrency();
    c->host_threads = ctpn_host_thread_budget((int)(hw ? hw : 1), env_int("LOCAL_WORLD_SIZE", 1), env_int("CTPN_HOST_THREADS", 0));
    const int first_cpu = env_int("CTPN_AFFINITY", 0) ? env_int("LOCAL_RANK", 0) * c->host_threads : -1;
    c->pool.reset(new HostPool(c->host_threads, first_cpu));
  }
  int rc = CTPN_OK;
  auto A = [&](void** p, size_t bytes, bool zero) { if (rc == CTPN_OK) rc = dev_alloc(c, p, bytes, zero); };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(CTPN_ERR_HIP, "hipStreamCreate failed"); }
  // (the proposal stream at the highest stream priority was measured in round 2: no effect -- placement is by free resources)
  // (the proposal stream at the highest stream priority: measured in round 6 with the tail confined -- 1163 against 1164 images/s in split precision,
  // -0.2 % in bf16: the dispatcher does not hand CUs to the 1024-thread NMS workgroups any sooner. Not used.)
  if (hipStreamCreateWithFlags(&c->stream_p, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return fail(CTPN_ERR_HIP, "hipStreamCreate failed"); }
  if (hipEventCreateWithFlags(&c->ev_conv, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess) {
    ctpn_destroy(c); return fail(CTPN_ERR_HIP, "ctpn_create: events");
  }
  for (auto& sl : c->slot) {
    const size_t mb = (size_t)max_batch;
    const PackLayout L = pack_layout(mb, (size_t)c->post_max);
    bool ok = hipHostMalloc((void**)&sl.pack, L.total) == hipSuccess;
    if (ok) {
      sl.tlb = (float*)(sl.pack + L.tlb); sl.tls = (float*)(sl.pack + L.tls); sl.keep = (int*)(sl.pack + L.keep); sl.kcnt = (int*)(sl.pack + L.kcnt);
      sl.rois = (float*)(sl.pack + L.rois); sl.rcnt = (int*)(sl.pack + L.rcnt);
    }
    ok = ok &&
              hipHostMalloc((void**)&sl.im_info, mb * 3 * sizeof(float)) == hipSuccess &&
              hipHostMalloc((void**)&sl.crecs, mb * 2 * CONN_CAP * 9 * sizeof(double)) == hipSuccess &&
              hipHostMalloc((void**)&sl.ccnt, mb * 3 * sizeof(int)) == hipSuccess &&
              hipEventCreateWithFlags(&sl.ev_heads, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&sl.ev_decoded, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) { ctpn_destroy(c); return fail(CTPN_ERR_HIP, "ctpn_create: pinned host buffers / events"); }
  }

  const int hf = lvl(max_h, 4), wf = lvl(max_w, 4);
  c->m5_max = (size_t)max_batch * hf * wf;
  if (!postproc_only) {
  A((void**)&c->arena, (size_t)CTPN_WEIGHT_FLOATS * sizeof(float), false);
  A((void**)&c->w_first, 27 * 64 * sizeof(float), false);
  A(&c->w_first_frags, CF_FRAGS_TOTAL, true);
  for (int i = 0; i < 14; ++i) {
    A((void**)&c->b_conv[i], (size_t)kConvs[i].co * sizeof(float), true);
    // 16-bit / fp32: [Co][9 Ci] elements; split precision: [Co][9][3 Ci] bf16
    if (i > 0) A(&c->wt_conv[i], (size_t)kConvs[i].co * 9 * kConvs[i].ci * (c->prec == DType::SPLIT ? 6 : c->es), true);
  }
  A(&c->wt_x, (size_t)1024 * c->wx_row_bytes, true);
  if (dtype_is_half(c->prec)) A(&c->wt_xf, (size_t)1024 * 512 * 2, true);
  A((void**)&c->b_x, 1024 * sizeof(float), true);
  A((void**)&c->wh, (size_t)2 * 128 * 512 * sizeof(float), true);
  A((void**)&c->wt_fc, (size_t)512 * 256 * sizeof(float), true);
  A((void**)&c->b_fc, 512 * sizeof(float), true);
  A((void**)&c->wt_h, (size_t)64 * 512 * sizeof(float), true);
  A((void**)&c->b_h, 64 * sizeof(float), true);
  A((void**)&c->wt_fold, (size_t)64 * 256 * sizeof(float), true);
  A((void**)&c->b_fold, 64 * sizeof(float), true);

  for (int i = 0; i < 14; ++i) {
    const int hl = lvl(max_h, kConvs[i].level), wl = lvl(max_w, kConvs[i].level);
    // + slack: the weights-in-registers conv kernel fetches edge tiles' input windows without clamping (conv3x3.hip), i.e. up to
    // 8 bordered rows + one window row past the last image; those pixels only feed outputs that are never stored
    // bytes per pixel: channels x element size; split precision: [hi | lo] planes = 4 bytes per channel, and rpn_conv/3x3 (which feeds the
    // LSTM projection GEMM) [hi | lo | hi] = 6
    const size_t pix_b = (size_t)kConvs[i].co * ((c->prec == DType::SPLIT && i == 13) ? 6 : c->es);
    c->act_conv_bytes[i] = ((size_t)max_batch * (hl + 2) * (wl + 2) + act_slack_pixels(wl)) * pix_b;
    const size_t front = act_front_pixels(wl) * pix_b;
    A(&c->act_conv[i], front + c->act_conv_bytes[i], true);
    if (c->act_conv[i]) c->act_conv[i] = (char*)c->act_conv[i] + front;     // allocs[] keeps the pointer hipFree needs
  }
  {
    const int pool_src[4] = {1, 3, 6, 9};
    for (int p = 0; p < 4; ++p) {
      const int hl = lvl(max_h, p + 1), wl = lvl(max_w, p + 1);
      c->act_pool_bytes[p] = ((size_t)max_batch * (hl + 2) * (wl + 2) + act_slack_pixels(wl)) * kConvs[pool_src[p]].co * c->es;
      const size_t front = act_front_pixels(wl) * kConvs[pool_src[p]].co * c->es;
      A(&c->act_pool[p], front + c->act_pool_bytes[p], true);
      if (c->act_pool[p]) c->act_pool[p] = (char*)c->act_pool[p] + front;
    }
  }
  if (dtype_is_half(c->prec)) { c->q_img_bytes = conv1_q_bytes(max_batch, max_h, max_w); A(&c->q_img, c->q_img_bytes, true); }
  A((void**)&c->img_dev, (size_t)max_batch * max_h * max_w * 3 * sizeof(float), false);
  c->img_dev_b[0] = c->img_dev;
  A((void**)&c->img_dev_b[1], (size_t)max_batch * max_h * max_w * 3 * sizeof(float), false);
  if (rc == CTPN_OK) {
    bool ok = hipStreamCreateWithFlags(&c->stream_c, hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b)
      ok = hipEventCreateWithFlags(&c->ev_copied[b], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_consumed[b], hipEventDisableTiming) == hipSuccess;
    if (!ok) rc = fail(CTPN_ERR_HIP, "ctpn_create: copy stream / events");
  }
  A((void**)&c->xp, c->m5_max * 1024 * sizeof(float), false);
  A((void**)&c->lstm_out, c->m5_max * 256 * sizeof(float), false);
  A((void**)&c->fc_out, c->m5_max * 512 * sizeof(float), false);
  A((void**)&c->heads, c->m5_max * 64 * sizeof(float), true);
  }  // !postproc_only
  A((void**)&c->cls_prob, c->m5_max * 20 * sizeof(float), false);
  A((void**)&c->bbox_pred, c->m5_max * 40 * sizeof(float), false);
  A((void**)&c->cls_in, c->m5_max * 20 * sizeof(float), false);
  A((void**)&c->bbox_in, c->m5_max * 40 * sizeof(float), false);
  c->npad_max = next_pow2(hf * wf * 10);
  A((void**)&c->keys, (size_t)max_batch * c->npad_max * sizeof(unsigned long long), false);
  A((void**)&c->keys_tmp, (size_t)max_batch * c->npad_max * sizeof(unsigned long long), false);
  A((void**)&c->boxes4, (size_t)max_batch * hf * wf * 10 * 4 * sizeof(float), false);
  A((void**)&c->sorted_boxes, (size_t)max_batch * c->topn_max * 4 * sizeof(float), false);
  A((void**)&c->sorted_scores, (size_t)max_batch * c->topn_max * sizeof(float), false);
  A((void**)&c->valid_counts, (size_t)max_batch * sizeof(int), true);
  A((void**)&c->keep_idx, (size_t)max_batch * c->topn_max * sizeof(int), false);
  {
    // what a submitted batch hands back to the host -- connector front end (tl_*), rois and their counts -- as views into one block
    const PackLayout L = pack_layout((size_t)max_batch, (size_t)c->post_max);
    c->pack_bytes = L.total;
    A((void**)&c->out_pack, L.total, true);
    if (c->out_pack) {
      c->tl_boxes = (float*)(c->out_pack + L.tlb); c->tl_scores = (float*)(c->out_pack + L.tls); c->tl_keep = (int*)(c->out_pack + L.keep);
      c->tl_keep_counts = (int*)(c->out_pack + L.kcnt); c->rois = (float*)(c->out_pack + L.rois); c->keep_counts = (int*)(c->out_pack + L.rcnt);
    }
  }
  A((void**)&c->kept_spill, (size_t)max_batch * c->topn_max * 4 * sizeof(float), false);
  A((void**)&c->sorted_anchor, (size_t)max_batch * c->topn_max * sizeof(int), false);
  A((void**)&c->roi_anchor, (size_t)max_batch * c->post_max * sizeof(int), true);
  A((void**)&c->tl_counts, (size_t)max_batch * sizeof(int), true);
  A((void**)&c->tl_spill, (size_t)max_batch * c->post_max * 4 * sizeof(float), false);
  A((void**)&c->nms_mw_scratch, (size_t)NMS_MW_CAP_BATCH * NMS_MW_SCRATCH_BYTES, true);
  A((void**)&c->nms_colid, (size_t)NMS_MW_CAP_BATCH * ((c->topn_max + 15) & ~15), true);
  A((void**)&c->conn_recs, (size_t)max_batch * 2 * CONN_CAP * 9 * sizeof(double), false);
  A((void**)&c->conn_counts, (size_t)max_batch * 3 * sizeof(int), true);
  A((void**)&c->conn_scratch, (size_t)max_batch * 1024 * 20 * sizeof(double), false);
  A((void**)&c->im_info_dev, (size_t)max_batch * 3 * sizeof(float), true);
  if (rc == CTPN_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(CTPN_ERR_HIP, "ctpn_create: sync failed");
  if (rc != CTPN_OK) { ctpn_destroy(c); return rc; }
  *out = c;
  return CTPN_OK;
}

int ctpn_create(ctpn_ctx** out, int device_id, int max_batch, int max_h, int max_w, int precision) {
  return create_impl(out, device_id, max_batch, max_h, max_w, precision, false);
}

int ctpn_create_postproc(ctpn_ctx** out, int device_id, int max_batch, int max_hf, int max_wf) {
  if (max_hf < 1 || max_wf < 1 || max_hf > (1 << 20) / 16 || max_wf > (1 << 20) / 16) return fail(CTPN_ERR_ARG, "ctpn_create_postproc: feature-map size out of range");
  return create_impl(out, device_id, max_batch, max_hf * 16, max_wf * 16, CTPN_PREC_FP32, true);
}

// ---- options: behaviour switches of ONE ctx (two ctxs in a process can choose differently; nothing here is read from the environment) ----
static int* option_slot(ctpn_ctx* c, const std::string& k) {
  if (k == "keep_acts") return &c->keep_acts;
  if (k == "conv1_kernel") return &c->conv1_mfma;
  if (k == "conv1_fuse") return &c->conv1_fuse;
  if (k == "conv_p64") return &c->conv_p64;
  if (k == "tail_confine") return &c->tail_confine;
  if (k == "nms_prefix") return &c->nms_prefix;
  if (k == "debug_hog") return &c->debug_hog;
  if (k == "debug_nms") return &c->debug_nms;
  if (k == "split_edge") return &c->split_edge;
  if (k == "lstm_split") return &c->lstm_split;
  if (k == "nms_columns") return &c->nms_columns;
  if (k == "nms_check") return &c->nms_check;
  if (k == "connect_device") return &c->connect_device;
  if (k == "tail_overlap") return &c->tail_overlap;
  return nullptr;
}
static const char* kOptionNames[] = {"keep_acts", "conv1_kernel", "conv1_fuse", "lstm_split", "nms_columns", "nms_check", "connect_device", "tail_overlap", "conv_p64", "tail_confine", "nms_prefix", "debug_hog", "debug_nms", "split_edge"};
int ctpn_option_count(void) { return (int)(sizeof(kOptionNames) / sizeof(kOptionNames[0])); }
const char* ctpn_option_name(int index) { return index >= 0 && index < ctpn_option_count() ? kOptionNames[index] : nullptr; }
int ctpn_set_option(ctpn_ctx* c, const char* key, int value) {
  if (!c || !key) return fail(CTPN_ERR_ARG, "ctpn_set_option: null pointer");
  int* slot = option_slot(c, key);
  if (!slot) return fail(CTPN_ERR_ARG, std::string("ctpn_set_option: unknown option ") + key);
  const std::string k(key);
  if (k == "conv1_kernel" ? (value < 0 || value > 2) : k == "nms_columns" ? (value < 0 || value > 3) : k == "debug_hog" ? (value < 0 || value > 200000) : k == "debug_nms" ? (value < 0 || value > 15) : (value != 0 && value != 1)) return fail(CTPN_ERR_ARG, std::string("ctpn_set_option: value out of range for ") + key);
  if (*slot == value) return CTPN_OK;
  // a switch changes what the queued work would read / which stream runs it: drain first
  CTPN_HIP_TRY(hipSetDevice(c->device));
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream));
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream_p));
  for (auto& sl : c->slot) if (sl.busy) return fail(CTPN_ERR_STATE, "ctpn_set_option: a submitted batch has not been collected");
  c->tail_pending = false;
  c->nms_mw_dirty = true;           // both streams are drained: the next proposal launch re-zeroes the multi-workgroup NMS's scratch (8 KB memset)
  *slot = value;
  return CTPN_OK;
}
int ctpn_get_option(ctpn_ctx* c, const char* key, int* value_out) {
  if (!c || !key || !value_out) return fail(CTPN_ERR_ARG, "ctpn_get_option: null pointer");
  int* slot = option_slot(c, key);
  if (!slot) return fail(CTPN_ERR_ARG, std::string("ctpn_get_option: unknown option ") + key);
  *value_out = *slot;
  return CTPN_OK;
}

int ctpn_host_threads(ctpn_ctx* c, int* threads_out) {
  if (!c || !threads_out) return fail(CTPN_ERR_ARG, "null pointer");
  *threads_out = c->host_threads;
  return CTPN_OK;
}

int ctpn_destroy(ctpn_ctx* c) {
  if (!c) return CTPN_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream_p) (void)hipStreamSynchronize(c->stream_p);
  if (c->stream_c) { (void)hipStreamSynchronize(c->stream_c); (void)hipStreamDestroy(c->stream_c); }
  for (int b = 0; b < 2; ++b) { if (c->pin_stage[b]) (void)hipHostFree(c->pin_stage[b]); if (c->ev_h2d_done[b]) (void)hipEventDestroy(c->ev_h2d_done[b]); }
  for (int b = 0; b < 2; ++b) { if (c->ev_copied[b]) (void)hipEventDestroy(c->ev_copied[b]); if (c->ev_consumed[b]) (void)hipEventDestroy(c->ev_consumed[b]); }
  for (auto& sl : c->slot) {
    for (void* p : {(void*)sl.pack, (void*)sl.im_info, (void*)sl.crecs, (void*)sl.ccnt}) if (p) (void)hipHostFree(p);
    for (hipEvent_t e : {sl.ev_heads, sl.ev_decoded, sl.ev_done}) if (e) (void)hipEventDestroy(e);
  }
  for (hipEvent_t e : {c->ev_conv, c->ev_tail}) if (e) (void)hipEventDestroy(e);
  for (auto& j : c->jpeg) {
    if (j.coef_host) (void)hipHostFree(j.coef_host);
    if (j.qt_host) (void)hipHostFree(j.qt_host);
    for (void* p : {(void*)j.coef_dev, (void*)j.qt_dev, (void*)j.out_dev}) if (p) (void)hipFree(p);
    for (hipEvent_t e : {j.ev_h2d, j.ev_ready, j.ev_consumed}) if (e) (void)hipEventDestroy(e);
  }
  for (void* p : {(void*)c->jpeg_planes, (void*)c->jpeg_raw}) if (p) (void)hipFree(p);
  for (void* p : c->jpeg_retired) (void)hipFree(p);
  if (c->stream_p) (void)hipStreamDestroy(c->stream_p);
  for (auto& r : c->pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return CTPN_OK;
}

int ctpn_sync(ctpn_ctx* c) {
  if (!c) return fail(CTPN_ERR_ARG, "null ctx");
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream));
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream_p));
  if (c->stream_c) CTPN_HIP_TRY(hipStreamSynchronize(c->stream_c));      // staged copies, ctpn_decode_jpeg_batch
  return CTPN_OK;
}
int ctpn_stream(ctpn_ctx* c, void** stream_out) {
  if (!c || !stream_out) return fail(CTPN_ERR_ARG, "null pointer");
  *stream_out = (void*)c->stream;
  return CTPN_OK;
}

int ctpn_load_weights_host(ctpn_ctx* c, const float* arena_host) {
  if (!c || !arena_host) return fail(CTPN_ERR_ARG, "null pointer");
  if (c->postproc_only) return fail(CTPN_ERR_STATE, "ctpn_load_weights_host: post-processing-only ctx (ctpn_create_postproc) has no network");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  CTPN_HIP_TRY(hipMemcpyAsync(c->arena, arena_host, (size_t)CTPN_WEIGHT_FLOATS * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return pack_weights(c);
}
int ctpn_load_weights_device(ctpn_ctx* c, const void* arena_dev) {
  if (!c || !arena_dev) return fail(CTPN_ERR_ARG, "null pointer");
  if (c->postproc_only) return fail(CTPN_ERR_STATE, "ctpn_load_weights_device: post-processing-only ctx (ctpn_create_postproc) has no network");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  CTPN_HIP_TRY(hipMemcpyAsync(c->arena, arena_dev, (size_t)CTPN_WEIGHT_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return pack_weights(c);
}

// ---------------------------------------------------------------------------------------------
// Weight broadcast over RCCL: the ONLY collective on the path -- 71.57 MB of fp32 once at start-up, nothing per
// batch. librccl is dlopen'ed by soname on first use (no link-time dependency; a process that has imported torch gets torch's copy,
// exactly like libamdhip64), the types below restate the five entry points of rccl.h that are used.
// ---------------------------------------------------------------------------------------------
extern "C++" {
namespace {
typedef struct rcclComm* rccl_comm_t;
struct rccl_unique_id { char internal[128]; };
struct RcclApi {
  int (*get_unique_id)(rccl_unique_id*) = nullptr;
  int (*comm_init_rank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
  int (*comm_init_all)(rccl_comm_t*, int, const int*) = nullptr;
  int (*comm_destroy)(rccl_comm_t) = nullptr;
  int (*bcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
  const char* (*err_string)(int) = nullptr;
  bool ok = false;
};
constexpr int kRcclFloat = 7;      // ncclFloat32 (rccl.h ncclDataType_t)
const RcclApi& rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    void* h = nullptr;
    const char* override_path = std::getenv("CTPN_RCCL_LIB");
    if (override_path && *override_path) h = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      const char* rp = std::getenv("ROCM_PATH");
      const std::string p = std::string(rp && *rp ? rp : "/opt/rocm") + "/lib/librccl.so";
      h = dlopen(p.c_str(), RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) return a;
    a.get_unique_id = (decltype(a.get_unique_id))dlsym(h, "ncclGetUniqueId");
    a.comm_init_rank = (decltype(a.comm_init_rank))dlsym(h, "ncclCommInitRank");
    a.comm_init_all = (decltype(a.comm_init_all))dlsym(h, "ncclCommInitAll");
    a.comm_destroy = (decltype(a.comm_destroy))dlsym(h, "ncclCommDestroy");
    a.bcast = (decltype(a.bcast))dlsym(h, "ncclBroadcast");
    a.group_start = (decltype(a.group_start))dlsym(h, "ncclGroupStart");
    a.group_end = (decltype(a.group_end))dlsym(h, "ncclGroupEnd");
    a.err_string = (decltype(a.err_string))dlsym(h, "ncclGetErrorString");
    a.ok = a.get_unique_id && a.comm_init_rank && a.comm_init_all && a.comm_destroy && a.bcast && a.group_start && a.group_end;
    return a;
  }();
  return api;
}
int rccl_fail(const char* what, int code) {
  const RcclApi& r = rccl_api();
  return fail(CTPN_ERR_HIP, std::string(what) + ": RCCL error " + std::to_string(code) + (r.err_string ? std::string(" (") + r.err_string(code) + ")" : std::string()));
}
}  // namespace
}  // extern "C++"

int ctpn_comm_unique_id(char* id_out, size_t capacity) {
  if (!id_out || capacity < CTPN_COMM_ID_BYTES) return fail(CTPN_ERR_ARG, "ctpn_comm_unique_id: buffer of at least CTPN_COMM_ID_BYTES required");
  const RcclApi& r = rccl_api();
  if (!r.ok) return fail(CTPN_ERR_NODEVICE, "ctpn_comm_unique_id: librccl.so could not be loaded (set CTPN_RCCL_LIB or ROCM_PATH)");
  rccl_unique_id id;
  const int e = r.get_unique_id(&id);
  if (e) return rccl_fail("ncclGetUniqueId", e);
  std::memcpy(id_out, id.internal, CTPN_COMM_ID_BYTES);
  return CTPN_OK;
}

int ctpn_broadcast_weights_rank(ctpn_ctx* c, const char* unique_id, int rank, int world, int root) {
  if (!c || !unique_id) return fail(CTPN_ERR_ARG, "null pointer");
  if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return fail(CTPN_ERR_ARG, "ctpn_broadcast_weights_rank: rank / root outside [0, world)");
  if (c->postproc_only) return fail(CTPN_ERR_STATE, "ctpn_broadcast_weights_rank: post-processing-only ctx has no network");
  if (rank == root && !c->weights_loaded) return fail(CTPN_ERR_STATE, "ctpn_broadcast_weights_rank: the root's weights are not loaded");
  const RcclApi& r = rccl_api();
  if (!r.ok) return fail(CTPN_ERR_NODEVICE, "ctpn_broadcast_weights_rank: librccl.so could not be loaded (set CTPN_RCCL_LIB or ROCM_PATH)");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  rccl_unique_id id;
  std::memcpy(id.internal, unique_id, CTPN_COMM_ID_BYTES);
  rccl_comm_t comm = nullptr;
  int e = r.comm_init_rank(&comm, world, id, rank);
  if (e) return rccl_fail("ncclCommInitRank", e);
  e = r.bcast(c->arena, c->arena, (size_t)CTPN_WEIGHT_FLOATS, kRcclFloat, root, comm, c->stream);
  const hipError_t he = hipStreamSynchronize(c->stream);
  (void)r.comm_destroy(comm);
  if (e) return rccl_fail("ncclBroadcast", e);
  if (he != hipSuccess) return fail(CTPN_ERR_HIP, std::string("ctpn_broadcast_weights_rank: ") + hipGetErrorString(he));
  return rank == root ? CTPN_OK : pack_weights(c);
}

int ctpn_broadcast_weights(ctpn_ctx** handles, int n) {
  if (!handles || n < 1) return fail(CTPN_ERR_ARG, "ctpn_broadcast_weights: handles / n");
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!handles[i] || handles[i]->postproc_only) return fail(CTPN_ERR_ARG, "ctpn_broadcast_weights: null or post-processing-only ctx");
    devs[i] = handles[i]->device;
    for (int j = 0; j < i; ++j) if (devs[j] == devs[i]) return fail(CTPN_ERR_ARG, "ctpn_broadcast_weights: two ctxs on the same device (RCCL needs one rank per GPU)");
  }
  if (!handles[0]->weights_loaded) return fail(CTPN_ERR_STATE, "ctpn_broadcast_weights: handles[0] has no weights loaded");
  if (n == 1) return CTPN_OK;
  const RcclApi& r = rccl_api();
  if (!r.ok) return fail(CTPN_ERR_NODEVICE, "ctpn_broadcast_weights: librccl.so could not be loaded (set CTPN_RCCL_LIB or ROCM_PATH)");
  std::vector<rccl_comm_t> comms(n, nullptr);
  int e = r.comm_init_all(comms.data(), n, devs.data());
  if (e) return rccl_fail("ncclCommInitAll", e);
  int rc = CTPN_OK;
  e = r.group_start();
  for (int i = 0; i < n && !e; ++i) {
    if (hipSetDevice(devs[i]) != hipSuccess) { rc = fail(CTPN_ERR_HIP, "ctpn_broadcast_weights: hipSetDevice"); break; }
    e = r.bcast(handles[i]->arena, handles[i]->arena, (size_t)CTPN_WEIGHT_FLOATS, kRcclFloat, 0, comms[i], handles[i]->stream);
  }
  const int e2 = r.group_end();
  if (!e) e = e2;
  for (int i = 0; i < n; ++i) {
    (void)hipSetDevice(devs[i]);
    if (hipStreamSynchronize(handles[i]->stream) != hipSuccess && rc == CTPN_OK) rc = fail(CTPN_ERR_HIP, "ctpn_broadcast_weights: stream sync");
  }
  for (int i = 0; i < n; ++i) (void)r.comm_destroy(comms[i]);
  if (e) return rccl_fail("ncclBroadcast", e);
  if (rc) return rc;
  for (int i = 1; i < n; ++i) {
    CTPN_HIP_TRY(hipSetDevice(devs[i]));
    if ((rc = pack_weights(handles[i]))) return rc;
  }
  return CTPN_OK;
}

// host copy on a few pool threads: one core moves ~10 GB/s, a 52 MB batch would cost 5 ms of the submitting thread
static void parallel_memcpy(HostPool* pool, void* dst, const void* src, size_t bytes) {
  const size_t chunk = (size_t)8 << 20;
  const int nt = (int)std::min<size_t>(8, (bytes + chunk - 1) / chunk);
  if (nt <= 1 || !pool) { std::memcpy(dst, src, bytes); return; }
  const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
  pool->run(nt, [=](int i) {
    const size_t lo = (size_t)i * per, hi = std::min(bytes, lo + per);
    if (lo < hi) std::memcpy((char*)dst + lo, (const char*)src + lo, hi - lo);
  }, 8);
}

static int forward_impl(ctpn_ctx* c, const void* images, int is_f32, int images_on_device, int n, int h, int w, bool tail_on_p = false) {
  if (!c || !images) return fail(CTPN_ERR_ARG, "null pointer");
  if (c->postproc_only) return fail(CTPN_ERR_STATE, "ctpn_forward: post-processing-only ctx (ctpn_create_postproc) has no network");
  if (!c->weights_loaded) return fail(CTPN_ERR_STATE, "ctpn_forward: weights not loaded");
  if (n <= 0 || n > c->max_batch || h < 16 || w < 16 || h > c->max_h || w > c->max_w)
    return fail(CTPN_ERR_CAPACITY, "ctpn_forward: batch/size outside what the ctx was created for (h, w >= 16)");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  int rc;
  hipStream_t ts = tail_on_p ? c->stream_p : s;      // stream of the recurrent tail
  if (!tail_on_p) {
    // a forward that keeps everything on `s` rewrites xp / lstm_out / heads there: after their readers on stream_p -- the previous
    // asynchronous batch's tail (if it ran there) and its decode kernel. The decode kernel reads `heads` only, so that wait sits in front of
    // the GEMM that writes `heads`, at the END of this forward (at its start it put the cross-stream round trip heads -> decode -> next
    // forward, ~45 us, between every two batches: round 4's batch-1 and batch-32 timelines)
    if (c->tail_pending) { CTPN_HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0)); c->tail_pending = false; }
  }
  auto wait_decoded = [&]() -> int {
    if (!tail_on_p && c->ev_last_decoded) CTPN_HIP_TRY(hipStreamWaitEvent(ts, c->ev_last_decoded, 0));
    return CTPN_OK;
  };
  // borders must be zero for this geometry
  if (c->gn != n || c->gh != h || c->gw != w) {
    if (c->tail_pending) { CTPN_HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0)); c->tail_pending = false; }   // it still reads rpn_conv's output
    for (int i = 0; i < 14; ++i) CTPN_HIP_TRY(hipMemsetAsync(c->act_conv[i], 0, c->act_conv_bytes[i], s));
    for (int p = 0; p < 4; ++p) CTPN_HIP_TRY(hipMemsetAsync(c->act_pool[p], 0, c->act_pool_bytes[p], s));
    if (c->q_img) CTPN_HIP_TRY(hipMemsetAsync(c->q_img, 0, c->q_img_bytes, s));      // the zero frame around every image (only image pixels are rewritten)
    c->gn = n; c->gh = h; c->gw = w;
  }
  const void* img = images;
  int staged = -1;
  if (!images_on_device) {
    // Host images cross PCIe on their own stream into one of two staging buffers, so the copy of this batch overlaps the
    // previous batch's convolutions (the forward stream only waits for the copy event). A buffer is reused two calls later,
    // after the conv1_1 launch that read it (ev_consumed).
    staged = c->img_flip;
    c->img_flip ^= 1;
    if (c->consumed_valid[staged]) CTPN_HIP_TRY(hipStreamWaitEvent(c->stream_c, c->ev_consumed[staged], 0));
    const size_t bytes = (size_t)n * h * w * 3 * (is_f32 ? 4 : 1);
    const void* src = images;
    hipPointerAttribute_t attr;
    const bool locked = hipPointerGetAttributes(&attr, images) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!locked) {
      // A pageable source makes the runtime stage the copy itself and serialise it with the other streams (measured: 13.5
      // instead of 11.4 ms / step). Stage it here instead: host memcpy into a page-locked buffer of the ctx (this thread,
      // while the GPU works on the previous batch), then a truly asynchronous copy.
      (void)hipGetLastError();
      if (c->pin_stage_bytes[staged] < bytes) {
        if (c->pin_stage[staged]) { CTPN_HIP_TRY(hipStreamSynchronize(c->stream_c)); CTPN_HIP_TRY(hipHostFree(c->pin_stage[staged])); c->pin_stage[staged] = nullptr; }
        CTPN_HIP_TRY(hipHostMalloc(&c->pin_stage[staged], bytes));
        c->pin_stage_bytes[staged] = bytes;
        if (!c->ev_h2d_done[staged]) CTPN_HIP_TRY(hipEventCreateWithFlags(&c->ev_h2d_done[staged], hipEventDisableTiming));
        c->h2d_valid[staged] = false;
      }
      if (c->h2d_valid[staged]) CTPN_HIP_TRY(hipEventSynchronize(c->ev_h2d_done[staged]));   // the copy that last read this staging buffer
      parallel_memcpy(c->pool.get(), c->pin_stage[staged], images, bytes);
      src = c->pin_stage[staged];
    }
    CTPN_HIP_TRY(hipMemcpyAsync(c->img_dev_b[staged], src, bytes, hipMemcpyHostToDevice, c->stream_c));
    if (!locked) { CTPN_HIP_TRY(hipEventRecord(c->ev_h2d_done[staged], c->stream_c)); c->h2d_valid[staged] = true; }
    CTPN_HIP_TRY(hipEventRecord(c->ev_copied[staged], c->stream_c));
    CTPN_HIP_TRY(hipStreamWaitEvent(s, c->ev_copied[staged], 0));
    img = c->img_dev_b[staged];
  }
  c->n = n; c->h = h; c->w = w;
  int jpeg_src = -1;
  if (images_on_device && c->jpeg_ready)
    for (int b = 0; b < 2; ++b)
      if (c->jpeg[b].ready_valid && images == (const void*)c->jpeg[b].out_dev) {      // decoded on stream_c: the forward waits for its kernels, not the host
        CTPN_HIP_TRY(hipStreamWaitEvent(s, c->jpeg[b].ev_ready, 0));
        jpeg_src = b;
      }
  bool via_q = false, fuse1 = false;
  {
    Timed t(c, CTPN_KIND_CONV_FIRST, (double)n * h * w * (3.0 + 64.0 * c->es));
    // 16-bit modes: "conv1_kernel" picks exact-pixel MFMA through the q-image (2, uint8 feed) / split-operand MFMA (1) / VALU (0); split precision always takes the
    // split-operand MFMA kernel (fp32-class sums, stored as (hi, lo) planes); fp32: the VALU kernel
    const bool frags = c->prec == DType::SPLIT || (c->conv1_mfma && dtype_is_half(c->prec));
    // uint8 feed of the 16-bit modes ("conv1_kernel" = 2): bytes -> q-image; conv1_1 then runs inside conv1_2's window stage (the production
    // path: its 69 MB per image are never stored) or, with keep_acts / "conv1_fuse" = 0, stand-alone from the q-image: the same bytes
    via_q = !is_f32 && dtype_is_half(c->prec) && c->conv1_mfma >= 2 && c->q_img != nullptr;
    fuse1 = via_q && c->conv1_fuse && !c->keep_acts && conv1_fusable(c->prec, n, h, w, 64, 64, true, false);
    if (via_q) {
      if ((rc = launch_image_to_q((const uint8_t*)img, c->q_img, c->prec, n, h, w, s))) return rc;
      if (!fuse1 && (rc = launch_conv_first_from_q(c->q_img, conv1_p_frags(c->w_first_frags, c->prec), c->act_conv[0], c->prec, n, h, w, 0, w, s))) return rc;
    } else if ((rc = launch_conv_first(img, is_f32, c->w_first, c->b_conv[0], c->act_conv[0], c->prec, n, h, w, s,
                                       frags ? c->w_first_frags : nullptr))) return rc;
  }
  c->act_valid[0] = !fuse1;      // fused: conv1_1's map exists only inside conv1_2's LDS windows
  if (jpeg_src >= 0) {           // the images came from ctpn_decode_jpeg_batch: its buffer may be rewritten once the first layer has read it
    CTPN_HIP_TRY(hipEventRecord(c->jpeg[jpeg_src].ev_consumed, s));
    c->jpeg[jpeg_src].consumed_valid = true;
  }
  if (staged >= 0) {
    CTPN_HIP_TRY(hipEventRecord(c->ev_consumed[staged], s));
    c->consumed_valid[staged] = true;
  }
  // the previous batch's tail (stream_p) overlaps conv1_1 only: the conv stack starts on an otherwise idle chip (its timed window too)
  // and rpn_conv's output, which lstm_pre reads, is not rewritten under it
  if (c->tail_pending) { CTPN_HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0)); c->tail_pending = false; }
  if (c->tail_confine && c->ev_last_done) CTPN_HIP_TRY(hipStreamWaitEvent(s, c->ev_last_done, 0));      // option "tail_confine": see its declaration
  const void* cur = c->act_conv[0];
  int pool_i = 0;
  hipEvent_t stack_a = nullptr, stack_b = nullptr;
  double stack_flops = 0.0;
  const bool stack_timed = c->prof && c->prof_mode == 2;
  if (stack_timed) {
    auto get = [&]() { hipEvent_t e; if (!c->free_events.empty()) { e = c->free_events.back(); c->free_events.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
    stack_a = get(); stack_b = get();
    CTPN_HIP_TRY(hipEventRecord(stack_a, s));
  }
  for (int i = 1; i < 14; ++i) {
    const int hl = lvl(h, kConvs[i].level), wl = lvl(w, kConvs[i].level);
    double flops = 2.0 * (double)n * hl * wl * 9.0 * kConvs[i].ci * kConvs[i].co;
    if (fuse1 && i == 1) flops += 2.0 * (double)n * h * w * 27.0 * 64.0;      // conv1_1 is part of this launch
    stack_flops += flops;
    const bool fuse = kConvs[i].pool_after != 0;
    void* full = (!fuse || c->keep_acts) ? c->act_conv[i] : nullptr;
    {
      Timed t(c, CTPN_KIND_CONV_GEMM, flops);
      const bool f1 = fuse1 && i == 1;
      if ((rc = launch_conv3x3(cur, c->wt_conv[i], c->b_conv[i], full, fuse ? c->act_pool[pool_i] : nullptr, c->prec, n, hl, wl,
                               kConvs[i].ci, kConvs[i].co, 1, s, (c->prec == DType::SPLIT && i == 13) ? 1 : 0,
                               f1 ? c->q_img : nullptr, f1 ? conv1_p_frags(c->w_first_frags, c->prec) : nullptr, (c->conv_p64 ? 1 : 0) | (c->split_edge ? 2 : 0)))) return rc;
    }
    c->act_valid[i] = full != nullptr;
    cur = fuse ? c->act_pool[pool_i] : c->act_conv[i];
    if (fuse) ++pool_i;
  }
  if (stack_timed) {
    CTPN_HIP_TRY(hipEventRecord(stack_b, s));
    ProfRec r{CTPN_KIND_CONV_GEMM, stack_a, stack_b, stack_flops};
    r.launches = 13;
    c->pending.push_back(r);
  }
  const int hf = lvl(h, 4), wf = lvl(w, 4);
  const long long M5 = (long long)n * hf * wf;
  {  // lstm_pre: x_t @ kernel[:512] + bias for both directions (on `s` also when the tail overlaps: next to conv1_1 this MFMA GEMM took
     // 644 us instead of 174, measured -- only the latency-bound recurrence and the small heads GEMM move to stream_p)
    IGemm g{};
    // split precision: rpn_conv/3x3 stored [hi | lo | hi] pixels, wt_x rows are [hi | hi | lo]: a plain bf16 GEMM over K = 1536
    const bool sp = c->prec == DType::SPLIT;
    g.a = cur; g.wt = c->wt_x; g.bias = c->b_x; g.out = c->xp;
    g.M = M5; g.Ci = sp ? 1536 : 512; g.ntaps = 1; g.Co = 1024; g.a_plain = 0; g.H = hf; g.W = wf; g.tap_base_y = 1; g.tap_base_x = 1;
    g.out_bordered = 0; g.ldc = 1024; g.relu = 0;
    Timed t(c, CTPN_KIND_GEMM, 2.0 * (double)M5 * 512 * 1024);
    // 16-bit modes: lstm_pre is stored as fp16 (half the 272 MB round trip between this GEMM and the recurrence; see bilstm.hip) and
    // computed by the resident-weight-slice kernel (lstm_pre.hip); fp32 and split precision: the im2col GEMM
    if (dtype_is_half(c->prec)) { if ((rc = launch_lstm_pre(cur, c->wt_xf, c->b_x, c->xp, c->prec, n, hf, wf, s))) return rc; }
    else if ((rc = launch_igemm(g, sp ? DType::BF16 : c->prec, DType::F32, s))) return rc;
  }
  if (tail_on_p) {
    CTPN_HIP_TRY(hipEventRecord(c->ev_conv, s));
    CTPN_HIP_TRY(hipStreamWaitEvent(ts, c->ev_conv, 0));
  }
  {
    Timed t(c, CTPN_KIND_BILSTM, (double)M5 * (1024.0 + 256.0) * 4.0, ts);
    // "lstm_split": the recurrent product on split-bf16 MFMAs (fp32-class; v_exp / v_rcp gate math, 1 ulp each) in every mode but the fp32 gate,
    // whose kernel (and split precision's with lstm_split = 0) is exact-fp32 MFMA with exact gates
    const bool half = dtype_is_half(c->prec);
    if ((rc = launch_bilstm(c->xp, half ? 1 : 0, c->wh, c->lstm_out, n * hf, wf, ts, (c->lstm_split && c->prec != DType::F32) ? 1 : 0, half ? 1 : 0))) return rc;
  }
  const bool fold_heads = dtype_is_half(c->prec) && !c->keep_acts;
  if (fold_heads) {  // lstm_out (256) -> bbox (40) | cls (20) through the pre-multiplied FC x heads matrix
    IGemm g{};
    g.a = c->lstm_out; g.wt = c->wt_fold; g.bias = c->b_fold; g.out = c->heads;
    g.M = M5; g.Ci = 256; g.ntaps = 1; g.Co = 60; g.a_plain = 1; g.lda = 256; g.out_bordered = 0; g.ldc = 64; g.relu = 0;
    if ((rc = wait_decoded())) return rc;
    Timed t(c, CTPN_KIND_GEMM, 2.0 * (double)M5 * 256 * 60, ts);
    if ((rc = launch_igemm(g, DType::F32, DType::F32, ts))) return rc;
  } else {
  {  // lstm_o FC 256 -> 512 (no activation, reference network.py:110-113)
    IGemm g{};
    g.a = c->lstm_out; g.wt = c->wt_fc; g.bias = c->b_fc; g.out = c->fc_out;
    g.M = M5; g.Ci = 256; g.ntaps = 1; g.Co = 512; g.a_plain = 1; g.lda = 256; g.out_bordered = 0; g.ldc = 512; g.relu = 0;
    Timed t(c, CTPN_KIND_GEMM, 2.0 * (double)M5 * 256 * 512, ts);
    if ((rc = launch_igemm(g, DType::F32, DType::F32, ts))) return rc;
  }
  {  // rpn_bbox_pred (40) | rpn_cls_score (20) in one 512 -> 60 GEMM
    IGemm g{};
    g.a = c->fc_out; g.wt = c->wt_h; g.bias = c->b_h; g.out = c->heads;
    g.M = M5; g.Ci = 512; g.ntaps = 1; g.Co = 60; g.a_plain = 1; g.lda = 512; g.out_bordered = 0; g.ldc = 64; g.relu = 0;
    if ((rc = wait_decoded())) return rc;
    Timed t(c, CTPN_KIND_GEMM, 2.0 * (double)M5 * 512 * 60, ts);
    if ((rc = launch_igemm(g, DType::F32, DType::F32, ts))) return rc;
  }
  }
  if (tail_on_p) { CTPN_HIP_TRY(hipEventRecord(c->ev_tail, ts)); c->tail_pending = true; }
  c->fc_valid = !fold_heads;
  c->forward_done = true;
  c->proposals_done = false;
  return CTPN_OK;
}

int ctpn_forward(ctpn_ctx* c, const uint8_t* images, int images_on_device, int n, int h, int w) {
  return forward_impl(c, images, 0, images_on_device, n, h, w);
}
int ctpn_forward_blob(ctpn_ctx* c, const float* blob, int blob_on_device, int n, int h, int w) {
  return forward_impl(c, blob, 1, blob_on_device, n, h, w);
}

int ctpn_feat_shape(ctpn_ctx* c, int* n, int* hf, int* wf) {
  if (!c) return fail(CTPN_ERR_ARG, "null ctx");
  if (!c->forward_done) return fail(CTPN_ERR_STATE, "no forward yet");
  if (n) *n = c->n; if (hf) *hf = lvl(c->h, 4); if (wf) *wf = lvl(c->w, 4);
  return CTPN_OK;
}

int ctpn_get_tensor(ctpn_ctx* c, const char* name, float* out_host, size_t capacity, int shape4[4]) {
  if (!c || !name || !out_host) return fail(CTPN_ERR_ARG, "null pointer");
  if (!c->forward_done) return fail(CTPN_ERR_STATE, "ctpn_get_tensor: no forward yet");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  const std::string nm(name);
  const int n = c->n, hf = lvl(c->h, 4), wf = lvl(c->w, 4);
  const void* src = nullptr; int H = 0, W = 0, C = 0, ld = 0; bool bordered = false; DType t = DType::F32;
  for (int i = 0; i < 14; ++i) if (nm == kConvs[i].name && !c->act_valid[i])
    return fail(CTPN_ERR_STATE, "ctpn_get_tensor: " + nm + (i == 0 ? " is computed inside conv1_2's window stage" : " is fused with its max-pool") + " and not stored; set the ctx option keep_acts = 1 (ctpn_set_option)");
  for (int i = 0; i < 14; ++i) if (nm == kConvs[i].name) { src = c->act_conv[i]; H = lvl(c->h, kConvs[i].level); W = lvl(c->w, kConvs[i].level); C = kConvs[i].co; ld = C; bordered = true; t = c->prec; }
  const int pool_src[4] = {1, 3, 6, 9};
  for (int p = 0; p < 4; ++p) if (nm == kPoolNames[p]) { src = c->act_pool[p]; H = lvl(c->h, p + 1); W = lvl(c->w, p + 1); C = kConvs[pool_src[p]].co; ld = C; bordered = true; t = c->prec; }
  if (nm == "lstm_pre") { src = c->xp; H = hf; W = wf; C = 1024; ld = 1024; if (dtype_is_half(c->prec)) t = DType::F16; }
  if (nm == "lstm_out") { src = c->lstm_out; H = hf; W = wf; C = 256; ld = 256; }
  if (nm == "lstm_o" && !c->fc_valid)
    return fail(CTPN_ERR_STATE, "ctpn_get_tensor: lstm_o is folded into the heads GEMM in the 16-bit modes; set the ctx option keep_acts = 1 (ctpn_set_option)");
  if (nm == "lstm_o") { src = c->fc_out; H = hf; W = wf; C = 512; ld = 512; }
  if (nm == "heads") { src = c->heads; H = hf; W = wf; C = 60; ld = 64; }
  if (nm == "rpn_cls_prob_reshape") { src = c->cls_prob; H = hf; W = wf; C = 20; ld = 20; }
  if (nm == "rpn_bbox_pred") { src = c->bbox_pred; H = hf; W = wf; C = 40; ld = 40; }
  if (!src) return fail(CTPN_ERR_ARG, "ctpn_get_tensor: unknown tensor name " + nm);
  if ((src == c->cls_prob || src == c->bbox_pred) && !c->proposals_done)
    return fail(CTPN_ERR_STATE, "ctpn_get_tensor: " + nm + " is produced by ctpn_proposals (the softmax is fused into the decode kernel)");
  const size_t need = (size_t)n * H * W * C;
  if (shape4) { shape4[0] = n; shape4[1] = H; shape4[2] = W; shape4[3] = C; }
  if (capacity < need) return fail(CTPN_ERR_CAPACITY, "ctpn_get_tensor: output buffer too small");
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream));
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream_p));
  // split precision: a pixel is [hi(C) | lo(C)] bf16 (rpn_conv/3x3: [hi | lo | hi]); the value is hi + lo
  const bool split = t == DType::SPLIT;
  if (split) ld = (src == c->act_conv[13] ? 3 : 2) * C;
  const int es = split ? 2 : dtype_bytes(t);
  const int Hs = bordered ? H + 2 : H, Ws = bordered ? W + 2 : W;
  const size_t bytes = (size_t)n * Hs * Ws * ld * es;
  std::vector<char> tmp(bytes);
  CTPN_HIP_TRY(hipMemcpy(tmp.data(), src, bytes, hipMemcpyDeviceToHost));
  const int o = bordered ? 1 : 0;
  for (int in = 0; in < n; ++in)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t sp = (((size_t)in * Hs + y + o) * Ws + x + o) * ld;
        float* d = out_host + (((size_t)in * H + y) * W + x) * C;
        if (src == c->xp) {                      // device layout: permuted gate columns -> TF's i | j | f | o order (fp16 in the 16-bit modes)
          if (es == 2) {
            const uint16_t* sh = (const uint16_t*)tmp.data() + sp;
            for (int ch = 0; ch < 1024; ++ch) d[ch] = host_f16_to_f32(sh[(ch & ~511) + lstm_gate_col(ch & 511)]);
          } else {
            const float* sf = (const float*)tmp.data() + sp;
            for (int ch = 0; ch < 1024; ++ch) d[ch] = sf[(ch & ~511) + lstm_gate_col(ch & 511)];
          }
        } else if (es == 4) {
          std::memcpy(d, (const float*)tmp.data() + sp, (size_t)C * 4);
        } else {
          const uint16_t* sb = (const uint16_t*)tmp.data() + sp;
          if (split) for (int ch = 0; ch < C; ++ch) d[ch] = host_bf16_to_f32(sb[ch]) + host_bf16_to_f32(sb[C + ch]);
          else if (t == DType::F16) for (int ch = 0; ch < C; ++ch) d[ch] = host_f16_to_f32(sb[ch]);
          else for (int ch = 0; ch < C; ++ch) d[ch] = host_bf16_to_f32(sb[ch]);
        }
      }
  return CTPN_OK;
}

int ctpn_proposals(ctpn_ctx* c, const float* im_info, int pre_nms_topn, int post_nms_topn, float nms_thresh, float min_size,
                   float* rois_out, int* counts_out) {
  if (!c) return fail(CTPN_ERR_ARG, "null ctx");
  if (!c->forward_done) return fail(CTPN_ERR_STATE, "ctpn_proposals: no forward yet");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  return run_proposals(c, c->heads, 0, c->n, lvl(c->h, 4), lvl(c->w, 4), im_info, pre_nms_topn, post_nms_topn, nms_thresh, min_size,
                       rois_out, counts_out);
}

int ctpn_proposals_from_host(ctpn_ctx* c, const float* cls_prob, const float* bbox_pred, int n, int hf, int wf, const float* im_info,
                             int pre_nms_topn, int post_nms_topn, float nms_thresh, float min_size, float* rois_out, int* counts_out) {
  if (!c || !cls_prob || !bbox_pred) return fail(CTPN_ERR_ARG, "null pointer");
  if (n <= 0 || n > c->max_batch || hf <= 0 || wf <= 0 || (size_t)n * hf * wf > c->m5_max)
    return fail(CTPN_ERR_CAPACITY, "ctpn_proposals_from_host: shape outside what the ctx was created for");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  const size_t m = (size_t)n * hf * wf;
  CTPN_HIP_TRY(hipMemcpyAsync(c->cls_in, cls_prob, m * 20 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  CTPN_HIP_TRY(hipMemcpyAsync(c->bbox_in, bbox_pred, m * 40 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return run_proposals(c, nullptr, 1, n, hf, wf, im_info, pre_nms_topn, post_nms_topn, nms_thresh, min_size, rois_out, counts_out);
}

int ctpn_proposal_anchors(ctpn_ctx* c, int* anchors_out, int post_nms_topn) {
  if (!c || !anchors_out) return fail(CTPN_ERR_ARG, "null pointer");
  if (c->last_prop_n <= 0) return fail(CTPN_ERR_STATE, "ctpn_proposal_anchors: no ctpn_proposals / ctpn_proposals_from_host call yet");
  if (post_nms_topn != c->last_post) return fail(CTPN_ERR_ARG, "ctpn_proposal_anchors: post_nms_topn differs from the proposals call");
  CTPN_HIP_TRY(hipSetDevice(c->device));
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream_p));
  CTPN_HIP_TRY(hipMemcpyAsync(anchors_out, c->roi_anchor, (size_t)c->last_prop_n * post_nms_topn * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  CTPN_HIP_TRY(hipStreamSynchronize(c->stream));
  return CTPN_OK;
}

// TextDetector.detect entirely on the device for one image's rois (the asynchronous detect path with option connect_device = 1):
// lines_prep_kernel (score > 0.7 prefix, boxes / scale) -> nms_kernel (0.2) -> connect_kernel. Test hook: lets the parity
// tests feed the reference-generated rois straight into connect_kernel.
int ctpn_debug_connect(int device_id, const float* rois, int r, int im_h, int im_w, float scale, int mode, double* recs_out,
                       int capacity, int* count_out) {
  if (!count_out) return fail(CTPN_ERR_ARG, "ctpn_debug_connect: count_out is null");
  *count_out = 0;
  if (mode != CTPN_MODE_H && mode != CTPN_MODE_O) return fail(CTPN_ERR_ARG, "ctpn_debug_connect: mode must be H(0) or O(1)");
  if (r < 0 || r > 1000 || (r > 0 && !rois)) return fail(CTPN_ERR_ARG, "ctpn_debug_connect: 0 <= r <= 1000 rows of [score,x1,y1,x2,y2]");
  const int ndev = ctpn_device_count();
  if (ndev <= 0) return fail(CTPN_ERR_NODEVICE, "ctpn_debug_connect: no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(CTPN_ERR_ARG, "ctpn_debug_connect: device_id out of range");
  CTPN_HIP_TRY(hipSetDevice(device_id));
  const int post = 1000;
  char* buf = nullptr;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_rois = take((size_t)(post + 1) * 5 * 4), o_cnt = take(16), o_info = take(16), o_tlb = take((size_t)post * 16), o_tls = take((size_t)post * 4),
               o_tlc = take(16), o_keep = take((size_t)post * 4), o_kc = take(16), o_spill = take((size_t)post * 16),
               o_recs = take((size_t)2 * CONN_CAP * 9 * 8), o_cc = take(16), o_scr = take((size_t)1024 * 20 * 8);
  CTPN_HIP_TRY(hipMalloc((void**)&buf, off));
  hipStream_t st = nullptr;
  int rc = CTPN_OK;
  auto done = [&](int code) { if (st) (void)hipStreamDestroy(st); (void)hipFree(buf); return code; };
#define DC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return done(fail(CTPN_ERR_HIP, std::string("ctpn_debug_connect: ") + hipGetErrorString(e_))); } while (0)
  DC_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  DC_TRY(hipMemsetAsync(buf, 0, off, st));
  const float info[3] = {(float)im_h, (float)im_w, scale};
  if (r) DC_TRY(hipMemcpyAsync(buf + o_rois, rois, (size_t)r * 5 * 4, hipMemcpyHostToDevice, st));
  DC_TRY(hipMemcpyAsync(buf + o_cnt, &r, 4, hipMemcpyHostToDevice, st));
  DC_TRY(hipMemcpyAsync(buf + o_info, info, 12, hipMemcpyHostToDevice, st));
  if ((rc = launch_lines_prep((const float*)(buf + o_rois), (const int*)(buf + o_cnt), (const float*)(buf + o_info), post, 0.7f, (float*)(buf + o_tlb),
                              (float*)(buf + o_tls), (int*)(buf + o_tlc), 1, st))) return done(rc);
  if ((rc = launch_nms((const float*)(buf + o_tlb), (const float*)(buf + o_tls), (const int*)(buf + o_tlc), post, 0.2f, post, (int*)(buf + o_keep), post,
                       (int*)(buf + o_kc), nullptr, (float*)(buf + o_spill), 1, st))) return done(rc);
  if ((rc = launch_connect((const float*)(buf + o_tlb), (const float*)(buf + o_tls), (const int*)(buf + o_keep), (const int*)(buf + o_kc), post,
                           (const float*)(buf + o_info), (double*)(buf + o_recs), (int*)(buf + o_cc), (double*)(buf + o_scr), CONN_CAP, 1, st))) return done(rc);
  int cc[3] = {0, 0, 0};
  DC_TRY(hipMemcpyAsync(cc, buf + o_cc, 12, hipMemcpyDeviceToHost, st));
  DC_TRY(hipStreamSynchronize(st));
  if (cc[2] != 0) return done(fail(CTPN_ERR_ARG, "text_lines: proposal x1 outside the image (reference raises IndexError)"));
  const int cnt = cc[mode == CTPN_MODE_O ? 1 : 0];
  *count_out = cnt;
  if (cnt > capacity || cnt > CONN_CAP) return done(fail(CTPN_ERR_CAPACITY, "ctpn_debug_connect: more lines than capacity"));
  if (cnt && !recs_out) return done(fail(CTPN_ERR_ARG, "ctpn_debug_connect: recs_out is null"));
  if (cnt) DC_TRY(hipMemcpy(recs_out, buf + o_recs + (size_t)(mode == CTPN_MODE_O ? 1 : 0) * CONN_CAP * 9 * 8, (size_t)cnt * 9 * 8, hipMemcpyDeviceToHost));
#undef DC_TRY
  return done(CTPN_OK);
}

// ---- standalone NMS (B1 seam) ----------------------------------------------------------------
namespace {
struct NmsCache {
  std::mutex mu;
  hipStream_t stream = nullptr;
  int cap = 0;
  float* boxes = nullptr; float* spill = nullptr; int* keep = nullptr; int* counts = nullptr;  // counts[0] = n in, counts[1] = n kept
};
NmsCache g_nms[16];
}  // namespace

int ctpn_nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float thresh, int device_id) {
  if (!keep_out || !num_out) return fail(CTPN_ERR_ARG, "ctpn_nms: null output");
  *num_out = 0;
  if (boxes_num == 0) return CTPN_OK;
  if (!boxes_host || boxes_num < 0 || boxes_dim < 4) return fail(CTPN_ERR_ARG, "ctpn_nms: boxes must be N x (>=4)");
  const int ndev = ctpn_device_count();
  if (ndev <= 0) return fail(CTPN_ERR_NODEVICE, "ctpn_nms: no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev || device_id >= 16) return fail(CTPN_ERR_ARG, "ctpn_nms: device_id out of range");
  NmsCache& nc = g_nms[device_id];
  std::lock_guard<std::mutex> lk(nc.mu);
  CTPN_HIP_TRY(hipSetDevice(device_id));
  if (!nc.stream) CTPN_HIP_TRY(hipStreamCreateWithFlags(&nc.stream, hipStreamNonBlocking));
  if (nc.cap < boxes_num) {
    if (nc.boxes) { (void)hipFree(nc.boxes); (void)hipFree(nc.spill); (void)hipFree(nc.keep); (void)hipFree(nc.counts); nc.boxes = nullptr; nc.cap = 0; }
    const int cap = boxes_num < 16384 ? 16384 : boxes_num;
    CTPN_HIP_TRY(hipMalloc((void**)&nc.boxes, (size_t)cap * 4 * sizeof(float)));
    CTPN_HIP_TRY(hipMalloc((void**)&nc.spill, (size_t)cap * 4 * sizeof(float)));
    CTPN_HIP_TRY(hipMalloc((void**)&nc.keep, (size_t)cap * sizeof(int)));
    CTPN_HIP_TRY(hipMalloc((void**)&nc.counts, 2 * sizeof(int)));
    nc.cap = cap;
  }
  std::vector<float> b4((size_t)boxes_num * 4);
  for (int i = 0; i < boxes_num; ++i) std::memcpy(&b4[(size_t)i * 4], boxes_host + (size_t)i * boxes_dim, 4 * sizeof(float));
  CTPN_HIP_TRY(hipMemcpyAsync(nc.boxes, b4.data(), b4.size() * sizeof(float), hipMemcpyHostToDevice, nc.stream));
  CTPN_HIP_TRY(hipMemcpyAsync(nc.counts, &boxes_num, sizeof(int), hipMemcpyHostToDevice, nc.stream));
  int rc = launch_nms(nc.boxes, nullptr, nc.counts, boxes_num, thresh, boxes_num, nc.keep, boxes_num, nc.counts + 1, nullptr, nc.spill, 1, nc.stream);
  if (rc) return rc;
  int nk = 0;
  CTPN_HIP_TRY(hipMemcpyAsync(&nk, nc.counts + 1, sizeof(int), hipMemcpyDeviceToHost, nc.stream));
  CTPN_HIP_TRY(hipStreamSynchronize(nc.stream));
  if (nk < 0 || nk > boxes_num) return fail(CTPN_ERR_HIP, "ctpn_nms: device returned an impossible keep count");
  CTPN_HIP_TRY(hipMemcpy(keep_out, nc.keep, (size_t)nk * sizeof(int), hipMemcpyDeviceToHost));
  *num_out = nk;
  return CTPN_OK;
}

// ---- JPEG: host entropy decode + device pixels (jpeg.hip) ----
int ctpn_jpeg_probe(const uint8_t* data, size_t len, int* h, int* w, int* ncomp, int* luma_sampling) {
  if (!data) return fail(CTPN_ERR_ARG, "ctpn_jpeg_probe: null pointer");
  return jpeg_probe(data, len, h, w, ncomp, luma_sampling);
}
size_t ctpn_jpeg_coef_capacity(int h, int w) { return (h > 0 && w > 0) ? jpeg_coef_capacity(h, w) : 0; }
int ctpn_jpeg_entropy_decode(const uint8_t* data, size_t len, int16_t* coef, size_t coef_capacity, uint16_t* qt, int* layout8) {
  if (!data || !coef || !qt || !layout8) return fail(CTPN_ERR_ARG, "ctpn_jpeg_entropy_decode: null pointer");
  JpegGeom g;
  const int rc = jpeg_entropy_decode(data, len, coef, coef_capacity, qt, &g);
  if (rc) return rc;
  const int l8[8] = {g.h, g.w, g.ncomp, g.hs0 | ((g.orient - 1) << 8), g.bw[0], g.bw[1], g.bh[0], g.bh[1]};
  std::memcpy(layout8, l8, sizeof(l8));
  return CTPN_OK;
}

// grow a device buffer to `need` bytes; the old allocation is retired (it may still be read by work in flight, or be held by the caller)
static int jpeg_grow_dev(ctpn_ctx* c, void** p, size_t& have, size_t need) {
  if (need <= have) return CTPN_OK;
  void* q = nullptr;
  CTPN_HIP_TRY(hipMalloc(&q, need));
  if (*p) c->jpeg_retired.push_back(*p);
  *p = q; have = need;
  return CTPN_OK;
}

static int jpeg_reserve(ctpn_ctx* c, ctpn_ctx::JpegBufs& J, size_t n, size_t cap, size_t raw_bytes, size_t out_bytes) {
  if (!c->jpeg_ready) {
    for (auto& j : c->jpeg)
      for (hipEvent_t* e : {&j.ev_h2d, &j.ev_ready, &j.ev_consumed}) CTPN_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    c->jpeg_ready = true;
  }
  int rc;
  if (n * cap > J.coef_elems) {        // (the copy that last read the page-locked block has been waited for by the caller)
    if (J.coef_host) CTPN_HIP_TRY(hipHostFree(J.coef_host));
    J.coef_host = nullptr;
    J.coef_elems = 0;                  // (a failed allocation below must not leave the old size standing next to a null block)
    CTPN_HIP_TRY(hipHostMalloc((void**)&J.coef_host, n * cap * sizeof(int16_t)));
    size_t have = J.coef_elems * sizeof(int16_t);
    if ((rc = jpeg_grow_dev(c, (void**)&J.coef_dev, have, n * cap * sizeof(int16_t)))) return rc;
    J.coef_elems = n * cap;
  }
  if (n > J.qt_imgs) {
    if (J.qt_host) CTPN_HIP_TRY(hipHostFree(J.qt_host));
    J.qt_host = nullptr;
    J.qt_imgs = 0;
    CTPN_HIP_TRY(hipHostMalloc((void**)&J.qt_host, n * 192 * sizeof(uint16_t)));
    size_t have = J.qt_imgs * 192 * sizeof(uint16_t);
    if ((rc = jpeg_grow_dev(c, (void**)&J.qt_dev, have, n * 192 * sizeof(uint16_t)))) return rc;
    J.qt_imgs = n;
  }
  if ((rc = jpeg_grow_dev(c, (void**)&J.out_dev, J.out_bytes, out_bytes + 256))) return rc;
  if ((rc = jpeg_grow_dev(c, (void**)&c->jpeg_planes, c->jpeg_planes_bytes, n * cap))) return rc;      // one byte per coefficient
  if (raw_bytes && (rc = jpeg_grow_dev(c, (void**)&c->jpeg_raw, c->jpeg_raw_bytes, raw_bytes + 256))) return rc;
  return CTPN_OK;
}

// read a whole file; false if it cannot be read
static bool jpeg_read_file(const char* path, std::vector<uint8_t>& buf, size_t limit = 0) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  bool ok = false;
  if (limit) {
    buf.resize(limit);
    const size_t got = std::fread(buf.data(), 1, limit, f);
    buf.resize(got);
    ok = got > 0;
  } else if (std::fseek(f, 0, SEEK_END) == 0) {
    const long sz = std::ftell(f);
    if (sz > 0 && sz <= (1L << 30) && std::fseek(f, 0, SEEK_SET) == 0) {
      buf.resize((size_t)sz);
      ok = std::fread(buf.data(), 1, (size_t)sz, f) == (size_t)sz;
    }
  }
  std::fclose(f);
  return ok;
}

int ctpn_jpeg_probe_files(const char* const* paths, int n, int* info4, int threads) {
  if (!paths || !info4 || n < 0) return fail(CTPN_ERR_ARG, "ctpn_jpeg_probe_files: bad arguments");
  for (int i = 0; i < n; ++i) if (!paths[i]) return fail(CTPN_ERR_ARG, "ctpn_jpeg_probe_files: null path");
  if (threads <= 0) { const unsigned hw = std::thread::hardware_concurrency(); threads = (int)std::min<unsigned>(16u, hw ? hw : 1u); }
  threads = std::max(1, std::min(threads, n));
  std::atomic<int> next(0);
  auto work = [&]() {
    std::vector<uint8_t> buf;
    for (int i; (i = next.fetch_add(1)) < n;) {
      int* o = info4 + 4 * (size_t)i;
      o[0] = o[1] = o[2] = o[3] = 0;
      // the headers normally end within the first 64 KB; a file with larger APPn segments is read whole
      for (const size_t limit : {(size_t)1 << 16, (size_t)0}) {
        if (!jpeg_read_file(paths[i], buf, limit)) break;
        int h = 0, w = 0, nc = 0, hs = 0;
        const int rc = jpeg_probe(buf.data(), buf.size(), &h, &w, &nc, &hs);
        if (rc == CTPN_OK) { o[0] = h; o[1] = w; o[2] = nc; o[3] = hs; break; }
        if (rc == CTPN_ERR_UNSUPPORTED || buf.size() < ((size_t)1 << 16)) break;
      }
    }
  };
  std::vector<std::thread> team;
  for (int t = 1; t < threads; ++t) team.emplace_back(work);
  work();
  for (auto& t : team) t.join();
  return CTPN_OK;
}

// one image's bytes for the host half: from the caller's memory, or read from the file inside the worker thread
struct JpegSource {
  const uint8_t* const* mem = nullptr; const size_t* sizes = nullptr;
  const char* const* paths = nullptr;
};

static int jpeg_decode_impl(ctpn_ctx* c, const JpegSource& src, int n, int h, int w, double fx, double fy, const uint8_t** images_dev_out, int* out_h, int* out_w);

int ctpn_decode_jpeg_batch(ctpn_ctx* c, const uint8_t* const* files, const size_t* sizes, int n, int h, int w, double fx, double fy,
                           const uint8_t** images_dev_out, int* out_h, int* out_w) {
  if (!c || !files || !sizes || !images_dev_out) return fail(CTPN_ERR_ARG, "ctpn_decode_jpeg_batch: null pointer");
  for (int i = 0; i < n; ++i) if (!files[i]) return fail(CTPN_ERR_ARG, "ctpn_decode_jpeg_batch: null file pointer");
  JpegSource src; src.mem = files; src.sizes = sizes;
  return jpeg_decode_impl(c, src, n, h, w, fx, fy, images_dev_out, out_h, out_w);
}

int ctpn_decode_jpeg_files(ctpn_ctx* c, const char* const* paths, int n, int h, int w, double fx, double fy, const uint8_t** images_dev_out, int* out_h, int* out_w) {
  if (!c || !paths || !images_dev_out) return fail(CTPN_ERR_ARG, "ctpn_decode_jpeg_files: null pointer");
  for (int i = 0; i < n; ++i) if (!paths[i]) return fail(CTPN_ERR_ARG, "ctpn_decode_jpeg_files: null path");
  JpegSource src; src.paths = paths;
  return jpeg_decode_impl(c, src, n, h, w, fx, fy, images_dev_out, out_h, out_w);
}

static int jpeg_decode_impl(ctpn_ctx* c, const JpegSource& src, int n, int h, int w, double fx, double fy, const uint8_t** images_dev_out, int* out_h, int* out_w) {
  if (c->postproc_only) return fail(CTPN_ERR_STATE, "ctpn_decode_jpeg_batch: post-processing-only ctx");
  if (n <= 0 || h <= 0 || w <= 0 || h > 65535 || w > 65535) return fail(CTPN_ERR_ARG, "ctpn_decode_jpeg_batch: empty batch / bad size");
  const bool resize = (fx > 0.0 && fx != 1.0) || (fy > 0.0 && fy != 1.0);
  if (!(fx > 0.0)) fx = 1.0;
  if (!(fy > 0.0)) fy = 1.0;
  int dh = h, dw = w;
  if (resize) { dh = resize_out_dim(h, fy); dw = resize_out_dim(w, fx); if (dh <= 0 || dw <= 0) return fail(CTPN_ERR_ARG, "ctpn_decode_jpeg_batch: empty output"); }
  CTPN_HIP_TRY(hipSetDevice(c->device));
  const int b = c->jpeg_flip;
  auto& J = c->jpeg[b];
  // the page-locked coefficient block was last read by the copy of two calls ago
  if (J.h2d_valid) CTPN_HIP_TRY(hipEventSynchronize(J.ev_h2d));
  const size_t cap = jpeg_coef_capacity(h, w);
  int rc = jpeg_reserve(c, J, (size_t)n, cap, resize ? (size_t)n * h * w * 3 : 0, (size_t)n * dh * dw * 3);
  if (rc) return rc;
  // host half: one image per worker thread
  std::vector<JpegGeom> geo((size_t)n);
  std::vector<int> st((size_t)n, CTPN_OK);
  std::vector<std::string> msg((size_t)n);
  c->pool->run(n, [&](int i) {
    const uint8_t* data = nullptr; size_t len = 0;
    static thread_local std::vector<uint8_t> filebuf;      // one per worker thread, reused from batch to batch
    try {
      if (src.paths) {
        if (!jpeg_read_file(src.paths[i], filebuf)) { st[i] = CTPN_ERR_ARG; msg[i] = std::string("cannot read ") + src.paths[i]; return; }
        data = filebuf.data(); len = filebuf.size();
      } else { data = src.mem[i]; len = src.sizes[i]; }
      st[i] = jpeg_entropy_decode(data, len, J.coef_host + (size_t)i * cap, cap, J.qt_host + (size_t)i * 192, &geo[i]);
      if (st[i]) msg[i] = ctpn_last_error();      // (the error text is per thread)
      if (filebuf.capacity() > ((size_t)8 << 20)) std::vector<uint8_t>().swap(filebuf);      // one huge file must not pin its size per worker for the run
    } catch (const std::exception& e) { st[i] = CTPN_ERR_CAPACITY; msg[i] = e.what(); }      // nothing may leave a worker thread
  });
  for (int i = 0; i < n; ++i) if (st[i]) return fail(st[i], "ctpn_decode_jpeg_batch: file " + std::to_string(i) + ": " + msg[i]);
  const JpegGeom& g = geo[0];
  for (int i = 0; i < n; ++i) {
    if (geo[i].oh != h || geo[i].ow != w) return fail(CTPN_ERR_ARG, "ctpn_decode_jpeg_batch: file " + std::to_string(i) + " is not " + std::to_string(h) + " x " + std::to_string(w) + " (as cv2.imread returns it: EXIF orientation applied)");
    if (geo[i].ncomp != g.ncomp || geo[i].hs0 != g.hs0 || geo[i].vs0 != g.vs0 || geo[i].orient != g.orient) return fail(CTPN_ERR_UNSUPPORTED, "ctpn_decode_jpeg_batch: the files of one batch must share one component layout and one EXIF orientation");
  }
  hipStream_t qs = c->stream_c;
  // the device buffers of this set: the forward that read out_dev two calls ago has passed its first layer
  if (J.consumed_valid) CTPN_HIP_TRY(hipStreamWaitEvent(qs, J.ev_consumed, 0));
  CTPN_HIP_TRY(hipMemcpy2DAsync(J.coef_dev, (size_t)g.coef_per_img * sizeof(int16_t), J.coef_host, cap * sizeof(int16_t), (size_t)g.coef_per_img * sizeof(int16_t), (size_t)n,
                                hipMemcpyHostToDevice, qs));
  CTPN_HIP_TRY(hipMemcpyAsync(J.qt_dev, J.qt_host, (size_t)n * 192 * sizeof(uint16_t), hipMemcpyHostToDevice, qs));
  CTPN_HIP_TRY(hipEventRecord(J.ev_h2d, qs));
  J.h2d_valid = true;
  if ((rc = launch_jpeg_pixels(J.coef_dev, J.qt_dev, c->jpeg_planes, resize ? c->jpeg_raw : J.out_dev, g, n, qs))) return rc;
  // resize_im (reference ctpn/demo.py:21-25: cv2.resize, INTER_LINEAR) of the decoded batch, in the same queue
  if (resize && (rc = launch_resize_linear(c->jpeg_raw, J.out_dev, 0, n, h, w, dh, dw, fx, fy, qs))) return rc;
  CTPN_HIP_TRY(hipEventRecord(J.ev_ready, qs));
  J.ready_valid = true;
  J.consumed_valid = false;      // until a forward reads this buffer
  J.out_n = n; J.out_h = dh; J.out_w = dw;
  c->jpeg_flip ^= 1;
  *images_dev_out = J.out_dev;
  if (out_h) *out_h = dh;
  if (out_w) *out_w = dw;
  return CTPN_OK;
}

int ctpn_jpeg_batch_fetch(ctpn_ctx* c, const uint8_t* images_dev, uint8_t* host_out, size_t capacity) {
  if (!c || !images_dev || !host_out) return fail(CTPN_ERR_ARG, "ctpn_jpeg_batch_fetch: null pointer");
  for (auto& J : c->jpeg)
    if (J.ready_valid && J.out_dev == images_dev) {
      const size_t bytes = (size_t)J.out_n * J.out_h * J.out_w * 3;
      if (capacity < bytes) return fail(CTPN_ERR_CAPACITY, "ctpn_jpeg_batch_fetch: capacity too small");
      CTPN_HIP_TRY(hipSetDevice(c->device));
      CTPN_HIP_TRY(hipEventSynchronize(J.ev_ready));
      CTPN_HIP_TRY(hipMemcpy(host_out, J.out_dev, bytes, hipMemcpyDeviceToHost));
      return CTPN_OK;
    }
  return fail(CTPN_ERR_STATE, "ctpn_jpeg_batch_fetch: not a live batch of ctpn_decode_jpeg_batch");
}

int ctpn_resize_dims(int h, int w, double fx, double fy, int* out_h, int* out_w) {
  if (!out_h || !out_w || h <= 0 || w <= 0 || !(fx > 0.0) || !(fy > 0.0)) return fail(CTPN_ERR_ARG, "ctpn_resize_dims: bad arguments");
  *out_h = resize_out_dim(h, fy);
  *out_w = resize_out_dim(w, fx);
  if (*out_h <= 0 || *out_w <= 0) return fail(CTPN_ERR_ARG, "ctpn_resize_dims: empty output");
  return CTPN_OK;
}

int ctpn_resize(int device_id, const void* src, int src_is_f32, int src_on_device, int n, int h, int w, double fx, double fy, void* dst,
                int dst_on_device, long long dst_capacity, int* out_h, int* out_w) {
  int dh = 0, dw = 0;
  int rc = ctpn_resize_dims(h, w, fx, fy, &dh, &dw);
  if (rc) return rc;
  if (out_h) *out_h = dh;
  if (out_w) *out_w = dw;
  if (!dst) return CTPN_OK;
  if (!src || n <= 0) return fail(CTPN_ERR_ARG, "ctpn_resize: null source / empty batch");
  const long long need = (long long)n * dh * dw * 3;
  if (dst_capacity < need) return fail(CTPN_ERR_CAPACITY, "ctpn_resize: dst_capacity too small");
  const int ndev = ctpn_device_count();
  if (ndev <= 0) return fail(CTPN_ERR_NODEVICE, "ctpn_resize: no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(CTPN_ERR_ARG, "ctpn_resize: device_id out of range");
  CTPN_HIP_TRY(hipSetDevice(device_id));
  const size_t es = src_is_f32 ? 4 : 1;
  const size_t sbytes = (size_t)n * h * w * 3 * es, dbytes = (size_t)need * es;
  void *ds = nullptr, *dd = nullptr;
  hipStream_t st = nullptr;
  auto cleanup = [&]() { if (ds) (void)hipFree(ds); if (dd) (void)hipFree(dd); if (st) (void)hipStreamDestroy(st); };
#define RS_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(CTPN_ERR_HIP, std::string("ctpn_resize: ") + hipGetErrorString(e_)); } } while (0)
  RS_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const void* s_in = src;
  if (!src_on_device) { RS_TRY(hipMalloc(&ds, sbytes)); RS_TRY(hipMemcpyAsync(ds, src, sbytes, hipMemcpyHostToDevice, st)); s_in = ds; }
  void* d_out = dst;
  if (!dst_on_device) { RS_TRY(hipMalloc(&dd, dbytes)); d_out = dd; }
  rc = launch_resize_linear(s_in, d_out, src_is_f32, n, h, w, dh, dw, fx, fy, st);
  if (rc) { cleanup(); return rc; }
  if (!dst_on_device) RS_TRY(hipMemcpyAsync(dst, dd, dbytes, hipMemcpyDeviceToHost, st));
  RS_TRY(hipStreamSynchronize(st));
#undef RS_TRY
  cleanup();
  return CTPN_OK;
}

int ctpn_text_lines(const float* boxes, const float* scores, int r, int im_h, int im_w, int mode, int device_id, double* recs_out,
                    int capacity, int* count_out) {
  if (!count_out) return fail(CTPN_ERR_ARG, "ctpn_text_lines: count_out is null");
  *count_out = 0;
  if (r > 0 && (!boxes || !scores)) return fail(CTPN_ERR_ARG, "ctpn_text_lines: null input");
  std::vector<double> recs;
  int rc = text_lines_host(boxes, scores, r, im_h, im_w, mode, device_id, recs);
  if (rc) return rc;
  const int cnt = (int)(recs.size() / 9);
  *count_out = cnt;
  if (cnt > capacity) return fail(CTPN_ERR_CAPACITY, "ctpn_text_lines: more lines than capacity");
  if (cnt && !recs_out) return fail(CTPN_ERR_ARG, "ctpn_text_lines: recs_out is null");
  if (cnt) std::memcpy(recs_out, recs.data(), recs.size() * sizeof(double));
  return CTPN_OK;
}

// lone: a synchronous ctpn_detect with nothing else in flight on this ctx -- the proposal layer and the connector front end then follow the
// forward on ITS stream instead of hopping to the proposal stream (an event record + a cross-queue wait: ~12 us of a lone image's millisecond;
// the second stream exists to run the tail under the NEXT batch's convolutions, and there is no next batch here)
static int detect_submit_body(ctpn_ctx* c, const uint8_t* images, int images_on_device, int n, int h, int w, const float* scales, int slot, bool lone);
static int detect_submit_impl(ctpn_ctx* c, const uint8_t* images, int images_on_device, int n, int h, int w, const float* scales, int slot, bool lone) {
  const int rc = detect_submit_body(c, images, images_on_device, n, h, w, scales, slot, lone);
  if (rc != CTPN_OK && c) c->nms_mw_dirty = true;        // an error anywhere in a submit (the connector NMS's launch included): see common.h, NMS_MW_OVERFLOW_OFF
  return rc;
}
static int detect_submit_body(ctpn_ctx* c, const uint8_t* images, int images_on_device, int n, int h, int w, const float* scales, int slot, bool lone) {
  if (!c) return fail(CTPN_ERR_ARG, "ctpn_detect_submit: null ctx");
  if (slot < 0 || slot > 1) return fail(CTPN_ERR_ARG, "ctpn_detect_submit: slot must be 0 or 1");
  ctpn_ctx::Slot& sl = c->slot[slot];
  if (sl.busy) return fail(CTPN_ERR_STATE, "ctpn_detect_submit: slot still holds an uncollected batch");
  int rc = forward_impl(c, images, 0, images_on_device, n, h, w, c->tail_overlap != 0);
  if (rc) {
    // a forward that failed midway may have consumed the cross-stream hand-over state (tail_pending is cleared when the wait is ENQUEUED,
    // the events are recorded later): drain both streams so that a retry starts from a quiet ctx, keeping the first error text
    const std::string first = ctpn_last_error();
    (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->stream_p);
    c->tail_pending = false; c->ev_last_decoded = nullptr; c->ev_last_done = nullptr;
    return fail(rc, first);
  }
  for (int i = 0; i < n; ++i) { sl.im_info[3 * i] = (float)h; sl.im_info[3 * i + 1] = (float)w; sl.im_info[3 * i + 2] = scales ? scales[i] : 1.0f; }
  const int post = c->post_max;
  hipStream_t p = lone ? c->stream : c->stream_p;
  if (!lone) {
    CTPN_HIP_TRY(hipEventRecord(sl.ev_heads, c->stream));
    CTPN_HIP_TRY(hipStreamWaitEvent(p, sl.ev_heads, 0));
  }
  // cfg.TEST.* defaults (reference lib/fast_rcnn/config.py:175-183)
  rc = enqueue_proposals(c, c->heads, 0, n, lvl(h, 4), lvl(w, 4), sl.im_info, 12000, post, 0.7f, 8.0f, p, sl.ev_decoded);
  if (rc) return rc;
  c->ev_last_decoded = sl.ev_decoded;
  // TextDetector.detect front end on device: score > 0.7 prefix, boxes / scale, NMS 0.2 (detectors.py:21-30)
  {
    Timed t(c, CTPN_KIND_NMS, (double)n * post * 24.0, p);
    if ((rc = launch_lines_prep(c->rois, c->keep_counts, c->im_info_dev, post, 0.7f, c->tl_boxes, c->tl_scores, c->tl_counts, n, p))) return rc;
    float max_scale = 0.f;
    for (int i = 0; i < n; ++i) max_scale = sl.im_info[3 * i + 2] > max_scale ? sl.im_info[3 * i + 2] : max_scale;
    if (c->nms_columns && nms_columns_tl_ok(lvl(w, 4), post, 0.2f, max_scale)) {
      if ((rc = launch_nms_columns(c->tl_boxes, c->tl_scores, c->tl_counts, post, 0.2f, post, c->tl_keep, post, c->tl_keep_counts, nullptr,
                                   c->tl_spill, n, lvl(w, 4), p, nullptr, nullptr, c->im_info_dev, nms_multi_wg(c, n, 0) ? c->nms_mw_scratch : nullptr))) return rc;
    } else if ((rc = launch_nms(c->tl_boxes, c->tl_scores, c->tl_counts, post, 0.2f, post, c->tl_keep, post, c->tl_keep_counts, nullptr,
                                c->tl_spill, n, p))) return rc;
  }
  if (c->connect_device) {
    // graph build, chains, line fit and filter_boxes on the device too, for both DETECT_MODEs (the mode is chosen at collect)
    if ((rc = launch_connect(c->tl_boxes, c->tl_scores, c->tl_keep, c->tl_keep_counts, post, c->im_info_dev, c->conn_recs, c->conn_counts,
                             c->conn_scratch, CONN_CAP, n, p))) return rc;
    CTPN_HIP_TRY(hipMemcpyAsync(sl.crecs, c->conn_recs, (size_t)n * 2 * CONN_CAP * 9 * sizeof(double), hipMemcpyDeviceToHost, p));
    CTPN_HIP_TRY(hipMemcpyAsync(sl.ccnt, c->conn_counts, (size_t)n * 3 * sizeof(int), hipMemcpyDeviceToHost, p));
  }
  CTPN_HIP_TRY(hipMemcpyAsync(sl.pack, c->out_pack, c->pack_bytes, hipMemcpyDeviceToHost, p));      // tl_* (host connector), rois, counts: one copy
  CTPN_HIP_TRY(hipEventRecord(sl.ev_done, p));
  c->ev_last_done = sl.ev_done;
  sl.n = n; sl.h = h; sl.w = w; sl.busy = true;
  return CTPN_OK;
}

int ctpn_detect_submit(ctpn_ctx* c, const uint8_t* images, int images_on_device, int n, int h, int w, const float* scales, int slot) {
  return detect_submit_impl(c, images, images_on_device, n, h, w, scales, slot, false);
}

int ctpn_detect_collect(ctpn_ctx* c, int slot, int mode, double* recs_out, int line_capacity, int* line_counts, float* rois_out,
                        int* roi_counts) {
  if (!c || !recs_out || !line_counts) return fail(CTPN_ERR_ARG, "ctpn_detect_collect: null pointer");
  if (slot < 0 || slot > 1) return fail(CTPN_ERR_ARG, "ctpn_detect_collect: slot must be 0 or 1");
  if (mode != CTPN_MODE_H && mode != CTPN_MODE_O) return fail(CTPN_ERR_ARG, "ctpn_detect_collect: mode must be H(0) or O(1)");
  ctpn_ctx::Slot& sl = c->slot[slot];
  if (!sl.busy) return fail(CTPN_ERR_STATE, "ctpn_detect_collect: nothing was submitted to this slot");
  CTPN_HIP_TRY(hipEventSynchronize(sl.ev_done));
  sl.busy = false;
  const int n = sl.n, h = sl.h, w = sl.w, post = c->post_max;
  if (rois_out) std::memcpy(rois_out, sl.rois, (size_t)n * post * 5 * sizeof(float));
  if (roi_counts) std::memcpy(roi_counts, sl.rcnt, (size_t)n * sizeof(int));
  if (c->connect_device) {
    for (int i = 0; i < n; ++i) {
      if (sl.ccnt[3 * i + 2] != 0) return fail(CTPN_ERR_ARG, "text_lines: proposal x1 outside the image (reference raises IndexError)");
      const int cnt = sl.ccnt[3 * i + (mode == CTPN_MODE_O ? 1 : 0)];
      line_counts[i] = cnt;
      if (cnt > line_capacity || cnt > CONN_CAP) return fail(CTPN_ERR_CAPACITY, "ctpn_detect: more lines than line_capacity");
      if (cnt) std::memcpy(recs_out + (size_t)i * line_capacity * 9, sl.crecs + ((size_t)i * 2 + (mode == CTPN_MODE_O ? 1 : 0)) * CONN_CAP * 9, (size_t)cnt * 9 * sizeof(double));
    }
    return CTPN_OK;
  }
  std::vector<int> status(n, 0);
  std::vector<std::string> errs(n);
  auto work = [&](int i) {
    const int nk = sl.kcnt[i];
    std::vector<float> kb((size_t)nk * 4), ks(nk);
    for (int j = 0; j < nk; ++j) {
      const int src = sl.keep[(size_t)i * post + j];
      std::memcpy(&kb[4 * j], &sl.tlb[((size_t)i * post + src) * 4], 4 * sizeof(float));
      ks[j] = sl.tls[(size_t)i * post + src];
    }
    std::vector<double> recs;
    int st = connect_lines(kb.data(), ks.data(), nk, h, w, mode, recs);
    if (st) { status[i] = st; errs[i] = ctpn_last_error(); return; }
    const int cnt = (int)(recs.size() / 9);
    line_counts[i] = cnt;
    if (cnt > line_capacity) { status[i] = CTPN_ERR_CAPACITY; errs[i] = "ctpn_detect: more lines than line_capacity"; return; }
    if (cnt) std::memcpy(recs_out + (size_t)i * line_capacity * 9, recs.data(), recs.size() * sizeof(double));
  };
  c->pool->run(n, work);      // persistent workers of the ctx (ctpn_host_thread_budget), one image per task
  for (int i = 0; i < n; ++i) if (status[i]) return fail(status[i], errs[i]);
  return CTPN_OK;
}

int ctpn_detect(ctpn_ctx* c, const uint8_t* images, int images_on_device, int n, int h, int w, const float* scales, int mode,
                double* recs_out, int line_capacity, int* line_counts, float* rois_out, int* roi_counts) {
  if (!c || !recs_out || !line_counts) return fail(CTPN_ERR_ARG, "ctpn_detect: null pointer");
  if (mode != CTPN_MODE_H && mode != CTPN_MODE_O) return fail(CTPN_ERR_ARG, "ctpn_detect: mode must be H(0) or O(1)");
  int slot = c->slot[0].busy ? 1 : 0;
  const bool lone = !c->slot[0].busy && !c->slot[1].busy && c->tail_overlap == 0;
  int rc = detect_submit_impl(c, images, images_on_device, n, h, w, scales, slot, lone);
  if (rc) return rc;
  return ctpn_detect_collect(c, slot, mode, recs_out, line_capacity, line_counts, rois_out, roi_counts);
}

// ---- diagnostics ---------------------------------------------------------------------------------------------
// One 3x3 conv (+bias+ReLU, optionally + 2x2 max-pool) on caller-supplied dense tensors: the unit-test hook for the
// conv kernels on shapes the VGG trunk never produces (odd sizes, tails, single rows). Not on the product path.
int ctpn_debug_cvt_bf16(int device_id, const float* in, uint16_t* out, int n, int use_hw_instruction) {
  if (!in || !out || n <= 0) return fail(CTPN_ERR_ARG, "ctpn_debug_cvt_bf16: bad argument");
  if (ctpn_device_count() <= 0) return fail(CTPN_ERR_NODEVICE, "ctpn_debug_cvt_bf16: no HIP device visible (no CPU fallback)");
  CTPN_HIP_TRY(hipSetDevice(device_id));
  float* d_in = nullptr; uint16_t* d_out = nullptr;
  CTPN_HIP_TRY(hipMalloc((void**)&d_in, (size_t)n * 4));
  CTPN_HIP_TRY(hipMalloc((void**)&d_out, (size_t)n * 2 + 4));
  CTPN_HIP_TRY(hipMemcpy(d_in, in, (size_t)n * 4, hipMemcpyHostToDevice));
  int rc = launch_cvt_bf16(d_in, d_out, n, use_hw_instruction, nullptr);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(CTPN_ERR_HIP, "ctpn_debug_cvt_bf16: kernel failed");
  if (!rc && hipMemcpy(out, d_out, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(CTPN_ERR_HIP, "ctpn_debug_cvt_bf16: copy failed");
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc;
}

int ctpn_debug_lds_dma(int device_id, const uint8_t* src, size_t bytes, uint8_t* out_clobber, uint8_t* out_keep) {
  if (!src || !out_clobber || !out_keep || bytes == 0 || bytes % 1024 != 0 || bytes > ((size_t)1 << 30)) return fail(CTPN_ERR_ARG, "ctpn_debug_lds_dma: bytes must be a positive multiple of 1024 (<= 1 GiB)");
  if (ctpn_device_count() <= 0) return fail(CTPN_ERR_NODEVICE, "ctpn_debug_lds_dma: no HIP device visible (no CPU fallback)");
  CTPN_HIP_TRY(hipSetDevice(device_id));
  char *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)d_in, (void*)d_a, (void*)d_b}) if (p) (void)hipFree(p); };
  struct Guard { decltype(cleanup)& f; ~Guard() { f(); } } guard{cleanup};
  CTPN_HIP_TRY(hipMalloc((void**)&d_in, bytes));
  CTPN_HIP_TRY(hipMalloc((void**)&d_a, bytes));
  CTPN_HIP_TRY(hipMalloc((void**)&d_b, bytes));
  CTPN_HIP_TRY(hipMemcpy(d_in, src, bytes, hipMemcpyHostToDevice));
  CTPN_HIP_TRY(hipMemset(d_a, 0xA5, bytes));
  CTPN_HIP_TRY(hipMemset(d_b, 0x5A, bytes));
  CTPN_HIP_TRY(hipDeviceSynchronize());
  int rc = launch_lds_dma_check(d_in, d_a, d_b, (int)(bytes / 1024), nullptr);
  if (rc) return rc;
  if (hipDeviceSynchronize() != hipSuccess) return fail(CTPN_ERR_HIP, "ctpn_debug_lds_dma: kernel failed");
  CTPN_HIP_TRY(hipMemcpy(out_clobber, d_a, bytes, hipMemcpyDeviceToHost));
  CTPN_HIP_TRY(hipMemcpy(out_keep, d_b, bytes, hipMemcpyDeviceToHost));
  return CTPN_OK;
}

int ctpn_debug_conv3x3(int device_id, const float* in_nhwc, const float* w_hwio, const float* bias, int n, int h, int w, int ci,
                       int co, int precision, int impl, int fuse_pool, float* out_full, float* out_pool) {
  if (!in_nhwc || !w_hwio || !bias) return fail(CTPN_ERR_ARG, "null pointer");
  if (precision < CTPN_PREC_FP32 || precision > CTPN_PREC_SPLIT) return fail(CTPN_ERR_ARG, "ctpn_debug_conv3x3: unknown precision");
  if (ctpn_device_count() <= 0) return fail(CTPN_ERR_NODEVICE, "ctpn_debug_conv3x3: no HIP device visible (no CPU fallback)");
  CTPN_HIP_TRY(hipSetDevice(device_id));
  const DType t = prec_dtype(precision);
  const bool split = t == DType::SPLIT;
  if (impl != 0 && impl != 1) return fail(CTPN_ERR_ARG, "ctpn_debug_conv3x3: impl is 0 (im2col GEMM) or 1 (the product kernels)");
  if (split && impl != 1) return fail(CTPN_ERR_ARG, "ctpn_debug_conv3x3: split precision exists in the tap-reuse kernels only (impl 1)");
  const int es = split ? 2 : dtype_bytes(t);                 // bytes per stored scalar
  const int cin_p = split ? 2 * ci : ci, cout_p = split ? 2 * co : co;      // scalars per pixel: split precision stores [hi | lo] planes
  const int Hp = h + 2, Wp = w + 2, ho = h / 2, wo = w / 2;
  const size_t in_elems = ((size_t)n * Hp * Wp + act_slack_pixels(w)) * cin_p, out_elems = (size_t)n * Hp * Wp * cout_p, pool_elems = (size_t)n * (ho + 2) * (wo + 2) * cout_p;
  const int co_pad = (co + 127) / 128 * 128;
  std::vector<char> hin(in_elems * es, 0);
  for (int in = 0; in < n; ++in) for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) for (int c = 0; c < ci; ++c) {
    const float v = in_nhwc[(((size_t)in * h + y) * w + x) * ci + c];
    const size_t o = (((size_t)in * Hp + y + 1) * Wp + x + 1) * cin_p + c;
    if (t == DType::F32) std::memcpy(&hin[o * 4], &v, 4);
    else if (t == DType::F16) { const uint16_t b = host_f32_to_f16(v); std::memcpy(&hin[o * 2], &b, 2); }
    else {
      const uint16_t b = host_f32_to_bf16(v); std::memcpy(&hin[o * 2], &b, 2);
      if (split) { const uint16_t l = host_f32_to_bf16(v - host_bf16_to_f32(b)); std::memcpy(&hin[(o + ci) * 2], &l, 2); }
    }
  }
  void *d_in = nullptr, *d_out = nullptr, *d_pool = nullptr, *d_wt = nullptr; float *d_w = nullptr, *d_b = nullptr;
  char* d_in_alloc = nullptr;
  hipStream_t s = nullptr;
  int rc = CTPN_OK;
  auto cleanup = [&]() { for (void* p : {(void*)d_in_alloc, d_out, d_pool, d_wt, (void*)d_w, (void*)d_b}) if (p) (void)hipFree(p); };
  struct Guard { decltype(cleanup)& f; ~Guard() { f(); } } guard{cleanup};
  const size_t in_front = act_front_pixels(w) * cin_p * es;
  const size_t wt_bytes = (size_t)co_pad * 9 * ci * (split ? 6 : es);
  CTPN_HIP_TRY(hipMalloc((void**)&d_in_alloc, in_front + in_elems * es));
  CTPN_HIP_TRY(hipMemset(d_in_alloc, 0, in_front));
  d_in = d_in_alloc + in_front;
  CTPN_HIP_TRY(hipMalloc(&d_out, out_elems * es));
  CTPN_HIP_TRY(hipMalloc(&d_pool, pool_elems * es + 256));
  CTPN_HIP_TRY(hipMalloc(&d_wt, wt_bytes));
  CTPN_HIP_TRY(hipMalloc((void**)&d_w, (size_t)9 * ci * co * 4));
  CTPN_HIP_TRY(hipMalloc((void**)&d_b, (size_t)co_pad * 4));
  CTPN_HIP_TRY(hipMemset(d_out, 0, out_elems * es));
  CTPN_HIP_TRY(hipMemset(d_pool, 0, pool_elems * es + 256));
  CTPN_HIP_TRY(hipMemset(d_wt, 0, wt_bytes));
  CTPN_HIP_TRY(hipMemset(d_b, 0, (size_t)co_pad * 4));
  CTPN_HIP_TRY(hipMemcpy(d_in, hin.data(), in_elems * es, hipMemcpyHostToDevice));
  CTPN_HIP_TRY(hipMemcpy(d_w, w_hwio, (size_t)9 * ci * co * 4, hipMemcpyHostToDevice));
  CTPN_HIP_TRY(hipMemcpy(d_b, bias, (size_t)co * 4, hipMemcpyHostToDevice));
  rc = split ? launch_pack_transpose_split(d_w, co, d_wt, 9, ci, co, s) : launch_pack_transpose(d_w, co, d_wt, 9 * ci, t, 9 * ci, co, s);
  bool host_pool = false;
  if (!rc) {
    if (impl == 1) {
      rc = launch_conv3x3(d_in, d_wt, d_b, (out_full || !fuse_pool) ? d_out : nullptr, fuse_pool ? d_pool : nullptr, t, n, h, w, ci, co, 1, s, 0);
    } else {
      // impl 0: the im2col GEMM (igemm.hip) as an independent reference of the same layer; its pool is taken on the host from the stored map
      IGemm g{};
      g.a = d_in; g.wt = d_wt; g.bias = d_b; g.out = d_out; g.M = (long long)n * h * w; g.Ci = ci; g.ntaps = 9; g.Co = co;
      g.H = h; g.W = w; g.out_bordered = 1; g.ldc = co; g.relu = 1;
      rc = launch_igemm(g, t, t, s);
      host_pool = fuse_pool != 0;
    }
  }
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(CTPN_ERR_HIP, "ctpn_debug_conv3x3: kernel failed");
  auto fetch = [&](void* dsrc, int H2, int W2, float* dst) -> int {
    const size_t elems = (size_t)n * (H2 + 2) * (W2 + 2) * cout_p;
    std::vector<char> tmp(elems * es);
    CTPN_HIP_TRY(hipMemcpy(tmp.data(), dsrc, elems * es, hipMemcpyDeviceToHost));
    for (int in = 0; in < n; ++in) for (int y = 0; y < H2; ++y) for (int x = 0; x < W2; ++x) for (int c = 0; c < co; ++c) {
      const size_t o = (((size_t)in * (H2 + 2) + y + 1) * (W2 + 2) + x + 1) * cout_p + c;
      float v;
      if (t == DType::F32) std::memcpy(&v, &tmp[o * 4], 4);
      else {
        uint16_t b; std::memcpy(&b, &tmp[o * 2], 2);
        if (t == DType::F16) v = host_f16_to_f32(b);
        else {
          v = host_bf16_to_f32(b);
          if (split) { uint16_t l; std::memcpy(&l, &tmp[(o + co) * 2], 2); v += host_bf16_to_f32(l); }
        }
      }
      dst[(((size_t)in * H2 + y) * W2 + x) * co + c] = v;
    }
    return CTPN_OK;
  };
  if (!rc && (out_full || host_pool)) {
    std::vector<float> full_tmp;
    float* fdst = out_full;
    if (!fdst) { full_tmp.resize((size_t)n * h * w * co); fdst = full_tmp.data(); }
    rc = fetch(d_out, h, w, fdst);
    if (!rc && host_pool && out_pool)
      for (int in = 0; in < n; ++in) for (int y = 0; y < ho; ++y) for (int x = 0; x < wo; ++x) for (int c = 0; c < co; ++c) {
        auto at = [&](int yy, int xx) { return fdst[(((size_t)in * h + yy) * w + xx) * co + c]; };
        out_pool[(((size_t)in * ho + y) * wo + x) * co + c] = std::max(std::max(at(2 * y, 2 * x), at(2 * y, 2 * x + 1)), std::max(at(2 * y + 1, 2 * x), at(2 * y + 1, 2 * x + 1)));
      }
  }
  if (!rc && out_pool && fuse_pool && !host_pool) rc = fetch(d_pool, ho, wo, out_pool);
  return rc;
}

int ctpn_profile_enable(ctpn_ctx* c, int on) {
  if (!c) return fail(CTPN_ERR_ARG, "null ctx");
  if (!on) { int rc = prof_drain(c); if (rc) return rc; }
  c->prof = on != 0;
  c->prof_mode = on == 2 ? 2 : 1;
  return CTPN_OK;
}
int ctpn_profile_reset(ctpn_ctx* c) {
  if (!c) return fail(CTPN_ERR_ARG, "null ctx");
  int rc = prof_drain(c);
  if (rc) return rc;
  for (int k = 0; k < CTPN_KIND_COUNT; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; c->prof_work[k] = 0; }
  return CTPN_OK;
}
int ctpn_profile_read(ctpn_ctx* c, int kind, double* ms, long long* launches, double* work) {
  if (!c) return fail(CTPN_ERR_ARG, "null ctx");
  if (kind < 0 || kind >= CTPN_KIND_COUNT) return fail(CTPN_ERR_ARG, "kind out of range");
  int rc = prof_drain(c);
  if (rc) return rc;
  if (ms) *ms = c->prof_ms[kind];
  if (launches) *launches = c->prof_n[kind];
  if (work) *work = c->prof_work[kind];
  return CTPN_OK;
}

}  // extern "C"
