// cv2.imread for PNG files (reference ctpn/demo.py:59 reads whatever the demo directory holds: data/demo has .jpg and .png) -> BGR uint8.
// Host only, and meant to be: a PNG is one DEFLATE stream (inflate is sequential by nature) followed by the row filters, each row depending
// on the one above it -- there is no part of it that wants a GPU, so the library decodes PNG files on host threads, one file per thread, into
// a batch buffer that goes to the device in ONE copy (ctpn_detect_submit's host-image path). What the reference's decoder (libpng behind
// cv2.imread(IMREAD_COLOR)) does with each kind is restated from the PNG specification (ISO/IEC 15948) and libpng's documented transforms:
//   colour type 2 (RGB) / 6 (RGBA), 8 bit     channel order reversed, alpha dropped (png_set_strip_alpha: no compositing)
//   colour type 0 (gray) 1 / 2 / 4 / 8 bit     expanded to 8 bit by replication of the bit pattern (x 255, x 85, x 17), gray -> B = G = R
//   colour type 4 (gray + alpha), 8 bit        gray -> B = G = R, alpha dropped
//   colour type 3 (palette) 1 / 2 / 4 / 8 bit  palette entries, tRNS ignored
//   Adam7 interlacing                          the seven passes written to their pixel positions
// 16-bit samples are CTPN_ERR_UNSUPPORTED and go to the caller's decoder (lib/utils/image.py keeps their high byte, libpng's png_set_strip_16:
// Pillow agrees for colour files and clips 16-bit gray instead, which that module corrects).
// Chunk CRCs of the critical chunks and the zlib Adler-32 are checked (libpng fails on those too). DEFLATE itself is a library's: libdeflate's
// whole-buffer zlib decompressor where the system has libdeflate.so.0 (2 x zlib's speed, and inflate is 3/4 of a PNG decode), else
// zlib's inflate(), the library libpng and Pillow sit on -- same format, same bytes out; both are dlopen'ed (see Deflate / Zlib below). Pinned byte for byte against Pillow's decode
// (tests/test_png.py, both back ends): cv2 is not in this image.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "common.h"

namespace ctpn {

struct PngHead { int h = 0, w = 0, depth = 0, color = 0, interlace = 0; };

// The DEFLATE / CRC-32 code is a system library's, found at run time (dlopen, no link-time dependency: libctpn_hip.so must load on a box that
// has neither): libdeflate.so.0 (whole-buffer zlib decompressor, 2 x zlib's speed; three functions of a stable C API, no header needed) and
// libz.so.1 (the streaming inflate libpng itself sits on; its few types are mirrored below, dlsym gives the functions). With neither, PNG files are
// CTPN_ERR_UNSUPPORTED and the caller's own decoder takes them.
struct Deflate {
  void* (*alloc)() = nullptr;
  int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;      // 0 = success
  void (*free_)(void*) = nullptr;
  uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
  bool ok = false;
};
// zlib's public ABI (z_stream and four constants, unchanged since zlib 1.2.0), declared here: the build needs no zlib development headers,
// as the run needs no link-time libz. The version string inflateInit_ checks against is the LOADED library's own (zlibVersion()).
struct ZStream {
  const unsigned char* next_in; unsigned int avail_in; unsigned long total_in;
  unsigned char* next_out; unsigned int avail_out; unsigned long total_out;
  const char* msg; void* state;
  void* (*zalloc)(void*, unsigned int, unsigned int); void (*zfree)(void*, void*); void* opaque;
  int data_type; unsigned long adler; unsigned long reserved;
};
constexpr int kZ_OK = 0, kZ_STREAM_END = 1, kZ_BUF_ERROR = -5, kZ_NO_FLUSH = 0;
struct Zlib {
  int (*inflate_init)(ZStream*, const char*, int) = nullptr;
  int (*inflate_)(ZStream*, int) = nullptr;
  int (*inflate_end)(ZStream*) = nullptr;
  unsigned long (*crc)(unsigned long, const unsigned char*, unsigned int) = nullptr;
  const char* (*version)() = nullptr;
  bool ok = false;
};
static std::atomic<int> g_png_zlib_only(0);
static const Deflate& deflate_lib() {
  static Deflate d;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    d.alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
    d.zlib_decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_zlib_decompress");
    d.free_ = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
    d.crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
    d.ok = d.alloc && d.zlib_decompress && d.free_ && d.crc;
  });
  return d;
}
static const Zlib& zlib_lib() {
  static Zlib z;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    z.inflate_init = (int (*)(ZStream*, const char*, int))dlsym(h, "inflateInit_");
    z.inflate_ = (int (*)(ZStream*, int))dlsym(h, "inflate");
    z.inflate_end = (int (*)(ZStream*))dlsym(h, "inflateEnd");
    z.crc = (unsigned long (*)(unsigned long, const unsigned char*, unsigned int))dlsym(h, "crc32");
    z.version = (const char* (*)())dlsym(h, "zlibVersion");
    z.ok = z.inflate_init && z.inflate_ && z.inflate_end && z.crc && z.version;
  });
  return z;
}
static inline bool use_libdeflate() { return deflate_lib().ok && !(g_png_zlib_only.load(std::memory_order_relaxed) && zlib_lib().ok); }
static inline bool have_deflate() { return deflate_lib().ok || zlib_lib().ok; }
static inline uint32_t png_crc(const uint8_t* p, size_t n) {
  return use_libdeflate() ? deflate_lib().crc(0, p, n) : (uint32_t)zlib_lib().crc(0, p, (unsigned int)n);
}

static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int png_head(const uint8_t* d, size_t len, PngHead& hd, std::string& why) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (len < 8 + 25 || std::memcmp(d, sig, 8) != 0) { why = "not a PNG file"; return CTPN_ERR_ARG; }
  if (!have_deflate()) { why = "neither libdeflate.so.0 nor libz.so.1 on this system"; return CTPN_ERR_UNSUPPORTED; }
  if (be32(d + 8) != 13 || std::memcmp(d + 12, "IHDR", 4) != 0) { why = "IHDR expected"; return CTPN_ERR_ARG; }
  if (be32(d + 29) != png_crc(d + 12, 17)) { why = "IHDR CRC"; return CTPN_ERR_ARG; }
  const uint32_t w = be32(d + 16), h = be32(d + 20);
  hd.depth = d[24]; hd.color = d[25]; hd.interlace = d[28];
  if (w == 0 || h == 0 || w > 65535 || h > 65535) { why = "bad size"; return w && h ? CTPN_ERR_UNSUPPORTED : CTPN_ERR_ARG; }
  hd.w = (int)w; hd.h = (int)h;
  if (d[26] != 0 || d[27] != 0 || hd.interlace > 1) { why = "bad IHDR"; return CTPN_ERR_ARG; }
  const int c = hd.color, b = hd.depth;
  const bool legal = (c == 0 && (b == 1 || b == 2 || b == 4 || b == 8 || b == 16)) || (c == 3 && (b == 1 || b == 2 || b == 4 || b == 8)) ||
                     ((c == 2 || c == 4 || c == 6) && (b == 8 || b == 16));
  if (!legal) { why = "illegal colour type / bit depth"; return CTPN_ERR_ARG; }
  if (b == 16) { why = "16-bit samples"; return CTPN_ERR_UNSUPPORTED; }
  return CTPN_OK;
}

static inline int png_channels(int color) { return color == 0 || color == 3 ? 1 : (color == 4 ? 2 : (color == 2 ? 3 : 4)); }

// undo one scanline's filter in place; prev = the unfiltered row above (nullptr: all zero), bpp = bytes per complete pixel (>= 1)
static int png_unfilter(int type, uint8_t* row, const uint8_t* prev, size_t n, int bpp) {
  switch (type) {
    case 0: return 0;
    case 1: for (size_t i = bpp; i < n; ++i) row[i] = (uint8_t)(row[i] + row[i - bpp]); return 0;
    case 2: if (prev) for (size_t i = 0; i < n; ++i) row[i] = (uint8_t)(row[i] + prev[i]); return 0;
    case 3:
      for (size_t i = 0; i < n; ++i) {
        const int a = i >= (size_t)bpp ? row[i - bpp] : 0, b = prev ? prev[i] : 0;
        row[i] = (uint8_t)(row[i] + ((a + b) >> 1));
      }
      return 0;
    case 4:
      for (size_t i = 0; i < n; ++i) {
        const int a = i >= (size_t)bpp ? row[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
        const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
        row[i] = (uint8_t)(row[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
      }
      return 0;
    default: return -1;
  }
}

// one unfiltered scanline of `npx` pixels -> BGR at out + k * stride for pixel k
static void png_row_to_bgr(const uint8_t* row, int npx, const PngHead& hd, const uint8_t* plte, int nplte, uint8_t* out, size_t stride) {
  const int c = hd.color, b = hd.depth;
  if (c == 2 || c == 6) {
    const int step = c == 2 ? 3 : 4;
    for (int k = 0; k < npx; ++k) { uint8_t* o = out + k * stride; const uint8_t* s = row + (size_t)k * step; o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; }
    return;
  }
  if (c == 4) { for (int k = 0; k < npx; ++k) { uint8_t* o = out + k * stride; o[0] = o[1] = o[2] = row[2 * k]; } return; }
  // one channel of b bits: gray (scaled to 8 bits) or a palette index
  const int mask = (1 << b) - 1, scale = b == 8 ? 1 : 255 / mask;
  for (int k = 0; k < npx; ++k) {
    int v;
    if (b == 8) v = row[k];
    else { const int per = 8 / b; v = (row[k / per] >> ((per - 1 - k % per) * b)) & mask; }
    uint8_t* o = out + k * stride;
    if (c == 0) o[0] = o[1] = o[2] = (uint8_t)(v * scale);
    else if (v < nplte) { o[0] = plte[3 * v + 2]; o[1] = plte[3 * v + 1]; o[2] = plte[3 * v]; }
    else o[0] = o[1] = o[2] = 0;      // an index past the palette: libpng reports it and leaves black
  }
}

static int png_decode(const uint8_t* d, size_t len, uint8_t* out, size_t cap, int want_h, int want_w, std::string& why) {
  PngHead hd;
  int rc = png_head(d, len, hd, why);
  if (rc) return rc;
  if ((want_h > 0 && hd.h != want_h) || (want_w > 0 && hd.w != want_w)) { why = "the file is " + std::to_string(hd.h) + " x " + std::to_string(hd.w) + ", not the announced size"; return CTPN_ERR_ARG; }
  if ((size_t)hd.h * hd.w * 3 > cap) { why = "output capacity too small"; return CTPN_ERR_CAPACITY; }
  const int bits = png_channels(hd.color) * hd.depth, bpp = std::max(1, bits / 8);
  // the passes: (x0, y0, dx, dy); a non-interlaced image is one pass over everything
  static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const int whole[1][4] = {{0, 0, 1, 1}};
  const int (*pass)[4] = hd.interlace ? adam7 : whole;
  const int npass = hd.interlace ? 7 : 1;
  size_t raw_bytes = 0;
  for (int p = 0; p < npass; ++p) {
    const int pw = (hd.w - pass[p][0] + pass[p][2] - 1) / pass[p][2], ph = (hd.h - pass[p][1] + pass[p][3] - 1) / pass[p][3];
    if (pw > 0 && ph > 0) raw_bytes += (size_t)ph * (1 + ((size_t)pw * bits + 7) / 8);
  }
  if (raw_bytes > ((size_t)1 << 31)) { why = "image too large"; return CTPN_ERR_UNSUPPORTED; }
  thread_local std::vector<uint8_t> raw;             // the inflated scanlines; every byte is written before it is read (got == raw_bytes below)
  try { raw.resize(raw_bytes); } catch (const std::bad_alloc&) { why = "out of memory for the scanlines"; return CTPN_ERR_CAPACITY; }
  // chunks: PLTE, IDAT ... IEND
  uint8_t plte[768];
  int nplte = 0;
  std::vector<std::pair<const uint8_t*, size_t>> idat;
  size_t idat_bytes = 0;
  size_t i = 8 + 25;
  while (i + 12 <= len) {
    const uint32_t L = be32(d + i);
    const uint8_t* type = d + i + 4;
    if ((size_t)L > len - i - 12) { why = "chunk runs past the end of the file"; return CTPN_ERR_ARG; }
    const uint8_t* body = d + i + 8;
    const bool critical = !(type[0] & 0x20);
    if (critical && be32(body + L) != png_crc(type, 4 + (size_t)L)) { why = "chunk CRC"; return CTPN_ERR_ARG; }
    if (std::memcmp(type, "IDAT", 4) == 0) { if (L) { idat.emplace_back(body, (size_t)L); idat_bytes += L; } }
    else if (std::memcmp(type, "PLTE", 4) == 0) {
      if (L % 3 != 0 || L > 768) { why = "bad PLTE"; return CTPN_ERR_ARG; }
      std::memcpy(plte, body, L);
      nplte = (int)(L / 3);
    } else if (std::memcmp(type, "IEND", 4) == 0) break;
    else if (critical && std::memcmp(type, "IHDR", 4) != 0) { why = "unknown critical chunk"; return CTPN_ERR_UNSUPPORTED; }
    i += 12 + (size_t)L;
  }
  // every IDAT chunk is a piece of ONE zlib stream
  size_t got = 0;
  const Deflate& dl = deflate_lib();
  bool done = false;
  if (use_libdeflate()) {
    thread_local std::vector<uint8_t> joined;
    const uint8_t* in = idat.empty() ? d : idat[0].first;
    if (idat.size() > 1) {
      joined.resize(idat_bytes);
      size_t o = 0;
      for (auto& c : idat) { std::memcpy(joined.data() + o, c.first, c.second); o += c.second; }
      in = joined.data();
    }
    void* dec = dl.alloc();
    if (dec) {
      const int zr = dl.zlib_decompress(dec, in, idat_bytes, raw.data(), raw_bytes, &got);
      dl.free_(dec);
      // 0 = all of it; a stream that holds MORE than the image (3 = insufficient space) is cut at the image, like libpng ("too much image
      // data" is a warning there); bad data / a short stream: let zlib's streaming inflate say which, with its message
      if (zr == 0) done = true;
    }
  }
  if (!done && !zlib_lib().ok) { why = "corrupt or short image data"; return CTPN_ERR_ARG; }
  if (!done) {
    const Zlib& zl = zlib_lib();
    ZStream z;
    std::memset(&z, 0, sizeof(z));
    if (zl.inflate_init(&z, zl.version(), (int)sizeof(ZStream)) != kZ_OK) { why = "inflateInit failed"; return CTPN_ERR_STATE; }
    z.next_out = raw.data();
    z.avail_out = (unsigned int)raw_bytes;
    for (auto& c : idat) {
      z.next_in = c.first;
      z.avail_in = (unsigned int)c.second;
      const int zr = zl.inflate_(&z, kZ_NO_FLUSH);
      if (zr == kZ_STREAM_END) break;
      if (zr != kZ_OK && !(zr == kZ_BUF_ERROR && z.avail_out == 0)) { why = std::string("inflate: ") + (z.msg ? z.msg : "error"); rc = CTPN_ERR_ARG; break; }
      if (z.avail_out == 0) break;
    }
    got = raw_bytes - z.avail_out;
    zl.inflate_end(&z);
    if (rc) return rc;
  }
  if (got != raw_bytes) { why = "image data ends early"; return CTPN_ERR_ARG; }
  if (hd.color == 3 && nplte == 0) { why = "palette image without PLTE"; return CTPN_ERR_ARG; }
  // filters, pass by pass, then the pixels to their places
  size_t off = 0;
  for (int p = 0; p < npass; ++p) {
    const int x0 = pass[p][0], y0 = pass[p][1], dx = pass[p][2], dy = pass[p][3];
    const int pw = (hd.w - x0 + dx - 1) / dx, ph = (hd.h - y0 + dy - 1) / dy;
    if (pw <= 0 || ph <= 0) continue;
    const size_t rb = ((size_t)pw * bits + 7) / 8;
    const uint8_t* prev = nullptr;
    for (int r = 0; r < ph; ++r) {
      uint8_t* row = raw.data() + off + 1;
      if (png_unfilter(row[-1], row, prev, rb, bpp)) { why = "bad filter type"; return CTPN_ERR_ARG; }
      png_row_to_bgr(row, pw, hd, plte, nplte, out + ((size_t)(y0 + r * dy) * hd.w + x0) * 3, (size_t)dx * 3);
      prev = row;
      off += 1 + rb;
    }
  }
  if (raw.capacity() > ((size_t)64 << 20)) std::vector<uint8_t>().swap(raw);      // scanlines of one huge image: not kept per worker thread for the run
  return CTPN_OK;
}

static bool png_read_file(const char* path, std::vector<uint8_t>& buf, size_t limit) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  bool ok = false;
  if (limit) {
    buf.resize(limit);
    buf.resize(std::fread(buf.data(), 1, limit, f));
    ok = !buf.empty();
  } else if (std::fseek(f, 0, SEEK_END) == 0) {
    const long sz = std::ftell(f);
    if (sz > 0 && sz <= (1L << 30) && std::fseek(f, 0, SEEK_SET) == 0) { buf.resize((size_t)sz); ok = std::fread(buf.data(), 1, (size_t)sz, f) == (size_t)sz; }
  }
  std::fclose(f);
  return ok;
}

// Decode workers: persistent threads, started on first use and grown on demand, each keeping its scratch buffers (the file's bytes, the
// inflated scanlines: 2.5 MB per 600 x 900 image) in thread_local storage -- a fresh std::thread team per call costs its start-up and,
// worse, 600 page faults per image on buffers the allocator has just returned to the kernel, all contending for one address space (measured:
// 32 threads decoded 32 images in 8.6 ms, one thread one image in 3.9 ms). A leaked singleton: the threads wait on a condition variable and
// end with the process.
class PngPool {
 public:
  static PngPool& get() { static PngPool* p = new PngPool; return *p; }
  // one(i) for i in [0, n) on up to `threads` threads, the caller among them; returns when all are done. Callers are served one at a time.
  void run(int n, int threads, const std::function<void(int)>& one) {
    if (threads <= 0) { const unsigned hw = std::thread::hardware_concurrency(); threads = (int)std::min<unsigned>(32u, hw ? hw : 1u); }
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) { for (int i = 0; i < n; ++i) one(i); return; }
    std::lock_guard<std::mutex> serial(job_mu_);
    {
      std::lock_guard<std::mutex> l(mu_);
      for (; workers_ < threads - 1; ++workers_) std::thread(&PngPool::loop, this).detach();
      fn_ = &one; n_ = n; next_.store(0); seats_ = threads - 1; open_ = true; ++gen_;
    }
    cv_.notify_all();
    for (int i; (i = next_.fetch_add(1)) < n;) one(i);
    std::unique_lock<std::mutex> l(mu_);
    open_ = false;                                     // a worker that wakes from here on stays out
    done_.wait(l, [&] { return running_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop() {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_.wait(l, [&] { return open_ && gen_ != seen && seats_ > 0; });
      seen = gen_;
      --seats_;
      ++running_;
      const std::function<void(int)>* fn = fn_;
      const int n = n_;
      l.unlock();
      for (int i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
      l.lock();
      if (--running_ == 0) done_.notify_all();
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, seats_ = 0, running_ = 0, workers_ = 0;
  bool open_ = false;
  unsigned long long gen_ = 0;
};

}  // namespace ctpn

using namespace ctpn;

extern "C" {

int ctpn_debug_png_backend(int zlib_only) {
  if (zlib_only >= 0) g_png_zlib_only.store(zlib_only ? 1 : 0);
  return use_libdeflate() ? 1 : 0;
}

int ctpn_png_probe(const uint8_t* data, size_t len, int* h, int* w, int* color_type, int* bit_depth) {
  if (!data) return fail(CTPN_ERR_ARG, "ctpn_png_probe: null pointer");
  PngHead hd; std::string why;
  const int rc = png_head(data, len, hd, why);
  if (rc) return fail(rc, "png: " + why);
  if (h) *h = hd.h; if (w) *w = hd.w; if (color_type) *color_type = hd.color; if (bit_depth) *bit_depth = hd.depth;
  return CTPN_OK;
}

int ctpn_png_decode(const uint8_t* data, size_t len, uint8_t* bgr_out, size_t capacity) {
  if (!data || !bgr_out) return fail(CTPN_ERR_ARG, "ctpn_png_decode: null pointer");
  std::string why;
  const int rc = png_decode(data, len, bgr_out, capacity, 0, 0, why);
  return rc ? fail(rc, "png: " + why) : CTPN_OK;
}

int ctpn_png_probe_files(const char* const* paths, int n, int* info4, int threads) {
  if (!paths || !info4 || n < 0) return fail(CTPN_ERR_ARG, "ctpn_png_probe_files: bad arguments");
  for (int i = 0; i < n; ++i) if (!paths[i]) return fail(CTPN_ERR_ARG, "ctpn_png_probe_files: null path");
  PngPool::get().run(n, threads <= 0 ? 16 : threads, [&](int i) {
    int* o = info4 + 4 * (size_t)i;
    o[0] = o[1] = o[2] = o[3] = 0;
    std::vector<uint8_t> buf;
    PngHead hd; std::string why;
    if (png_read_file(paths[i], buf, 64) && png_head(buf.data(), buf.size(), hd, why) == CTPN_OK) { o[0] = hd.h; o[1] = hd.w; o[2] = hd.color; o[3] = hd.depth; }
  });
  return CTPN_OK;
}

int ctpn_decode_png_files(const char* const* paths, int n, int h, int w, uint8_t* bgr_out, int threads) {
  if (!paths || !bgr_out || n < 0 || h <= 0 || w <= 0) return fail(CTPN_ERR_ARG, "ctpn_decode_png_files: bad arguments");
  for (int i = 0; i < n; ++i) if (!paths[i]) return fail(CTPN_ERR_ARG, "ctpn_decode_png_files: null path");
  std::vector<int> st((size_t)n, CTPN_OK);
  std::vector<std::string> msg((size_t)n);
  const size_t per = (size_t)h * w * 3;
  PngPool::get().run(n, threads, [&](int i) {
    thread_local std::vector<uint8_t> buf;
    try {
      if (!png_read_file(paths[i], buf, 0)) { st[i] = CTPN_ERR_ARG; msg[i] = std::string("cannot read ") + paths[i]; return; }
      st[i] = png_decode(buf.data(), buf.size(), bgr_out + per * i, per, h, w, msg[i]);
      if (buf.capacity() > ((size_t)16 << 20)) std::vector<uint8_t>().swap(buf);      // one huge file must not pin its size per worker for the run
    } catch (const std::exception& e) { st[i] = CTPN_ERR_CAPACITY; msg[i] = e.what(); }      // nothing may leave a worker thread
  });
  for (int i = 0; i < n; ++i) if (st[i]) return fail(st[i], "ctpn_decode_png_files: file " + std::to_string(i) + " (" + paths[i] + "): " + msg[i]);
  return CTPN_OK;
}

}  // extern "C"
