// Text-line connector, host C++ (the Python double loops of the reference).
//
// Restates, with the same fp32 / fp64 rounding points (numpy-2 scalar promotion):
//   TextDetector.detect / filter_boxes     reference lib/text_connector/detectors.py:19-49
//   TextProposalGraphBuilder               lib/text_connector/text_proposal_graph_builder.py:6-78
//   Graph.sub_graphs_connected             lib/text_connector/other.py:20-29
//   TextProposalConnector (H)              lib/text_connector/text_proposal_connector.py:21-64
//   TextProposalConnector (O)              lib/text_connector/text_proposal_connector_oriented.py:24-105
//   clip_boxes                             lib/text_connector/other.py:7-13
//   constants                              lib/text_connector/text_connect_cfg.py:1-12
// Tie order where the reference leaves it to numpy's unstable sort: descending score, equal scores by
// ascending input index.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "common.h"

#pragma STDC FP_CONTRACT OFF

namespace ctpn {

namespace {
constexpr float kMinScore = 0.7f;        // TEXT_PROPOSALS_MIN_SCORE
constexpr float kNmsThresh = 0.2f;       // TEXT_PROPOSALS_NMS_THRESH
constexpr int kMaxGap = 50;              // MAX_HORIZONTAL_GAP
constexpr float kMinVOverlaps = 0.7f;    // MIN_V_OVERLAPS
constexpr float kMinSizeSim = 0.7f;      // MIN_SIZE_SIM
constexpr double kMinRatio = 0.5;        // MIN_RATIO
constexpr double kLineMinScore = 0.9;    // LINE_MIN_SCORE
constexpr double kMinWidth = 16.0 * 2;   // TEXT_PROPOSALS_WIDTH * MIN_NUM_PROPOSALS

struct Props {
  std::vector<float> x1, y1, x2, y2, h, s;
  std::vector<std::vector<int>> table;  // boxes_table: proposals bucketed by int(x1)
  int im_w = 0;
  size_t size() const { return x1.size(); }
};

bool meet_v_iou(const Props& p, int a, int b) {
  const float h1 = p.h[a], h2 = p.h[b];
  const float y0 = std::max(p.y1[b], p.y1[a]);
  const float y1 = std::min(p.y2[b], p.y2[a]);
  const float ov = std::max(0.0f, y1 - y0 + 1.0f) / std::min(h1, h2);
  const float sim = std::min(h1, h2) / std::max(h1, h2);
  return ov >= kMinVOverlaps && sim >= kMinSizeSim;
}

// first non-empty column to the right of `index` (x1+1 .. x1+50, inside the image)
void successions(const Props& p, int index, std::vector<int>& out) {
  out.clear();
  const int x = (int)p.x1[index];
  const int hi = std::min(x + kMaxGap + 1, p.im_w);
  for (int left = x + 1; left < hi; ++left) {
    for (int adj : p.table[left])
      if (meet_v_iou(p, adj, index)) out.push_back(adj);
    if (!out.empty()) return;
  }
}

// first non-empty column to the left of `index`
void precursors(const Props& p, int index, std::vector<int>& out) {
  out.clear();
  const int x = (int)p.x1[index];
  const int lo = std::max((int)(p.x1[index] - (float)kMaxGap), 0);
  for (int left = x - 1; left >= lo; --left) {
    for (int adj : p.table[left])
      if (meet_v_iou(p, adj, index)) out.push_back(adj);
    if (!out.empty()) return;
  }
}

// numpy's float32 add.reduce (ndarray.sum / np.mean of a contiguous float32 array; numpy/core/src/umath/loops_utils.h.src
// pairwise_sum): fewer than 8 elements sequentially from 0, up to 128 in eight strided accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) with the remainder added last, larger arrays split in halves (the left one a multiple of 8).
// The reference's line score `scores[idx].sum() / n` (text_proposal_connector.py:39) and mean height (..._oriented.py:59) are
// this sum; a sequential fp32 sum differs from it in the last bit for chains of 8 or more proposals.
float np_sum_f32(const float* a, size_t n) {
  if (n < 8) {
    float r = 0.f;
    for (size_t i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    size_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  size_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_sum_f32(a, n2) + np_sum_f32(a + n2, n - n2);
}

// np.polyfit(X, Y, 1) on fp32 data: double least squares, coefficients rounded to fp32
void polyfit1(const std::vector<float>& X, const std::vector<float>& Y, float& c0, float& c1) {
  const size_t n = X.size();
  double mx = 0, my = 0;
  for (size_t i = 0; i < n; ++i) { mx += X[i]; my += Y[i]; }
  mx /= (double)n; my /= (double)n;
  double sxx = 0, sxy = 0;
  for (size_t i = 0; i < n; ++i) {
    const double dx = X[i] - mx;
    sxx += dx * dx;
    sxy += dx * (Y[i] - my);
  }
  const double slope = sxx > 0 ? sxy / sxx : 0.0;
  c0 = (float)slope;
  c1 = (float)(my - slope * mx);
}

void fit_y(const std::vector<float>& X, const std::vector<float>& Y, float xa, float xb, float& ya, float& yb) {
  bool all_same = true;
  for (float v : X) if (v != X[0]) { all_same = false; break; }
  if (all_same) { ya = Y[0]; yb = Y[0]; return; }
  float c0, c1;
  polyfit1(X, Y, c0, c1);
  ya = c0 * xa + c1;
  yb = c0 * xb + c1;
}

float clampf(float v, float lo, float hi) { return std::max(std::min(v, hi), lo); }
}  // namespace

void nms_host(const float* boxes, int n, int dim, float thresh, std::vector<int>& keep) {
  keep.clear();
  std::vector<char> dead((size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep.push_back(i);
    const float* a = boxes + (size_t)i * dim;
    const float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    for (int j = i + 1; j < n; ++j) {
      if (dead[j]) continue;
      const float* b = boxes + (size_t)j * dim;
      const float left = std::max(a[0], b[0]), right = std::min(a[2], b[2]);
      const float top = std::max(a[1], b[1]), bottom = std::min(a[3], b[3]);
      const float w = std::max(right - left + 1.f, 0.f), h = std::max(bottom - top + 1.f, 0.f);
      const float inter = w * h;
      const float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
      if (inter / (sa + sb - inter) > thresh) dead[j] = 1;
    }
  }
}

int text_lines_host(const float* boxes, const float* scores, int r, int im_h, int im_w, int mode, int device_id,
                    std::vector<double>& recs) {
  recs.clear();
  if (r < 0 || im_h <= 0 || im_w <= 0) return fail(CTPN_ERR_ARG, "text_lines: bad size");
  if (mode != CTPN_MODE_H && mode != CTPN_MODE_O) return fail(CTPN_ERR_ARG, "text_lines: mode must be H(0) or O(1)");

  // detect(): score filter, descending sort, NMS 0.2
  std::vector<int> order;
  for (int i = 0; i < r; ++i)
    if (scores[i] > kMinScore) order.push_back(i);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
  const int n0 = (int)order.size();
  if (n0 == 0) return CTPN_OK;
  std::vector<float> dets((size_t)n0 * 5);
  for (int i = 0; i < n0; ++i) {
    const float* b = boxes + (size_t)order[i] * 4;
    float* d = &dets[(size_t)i * 5];
    d[0] = b[0]; d[1] = b[1]; d[2] = b[2]; d[3] = b[3]; d[4] = scores[order[i]];
  }
  std::vector<int> keep;
  if (device_id >= 0) {
    keep.resize(n0);
    int nk = 0;
    const int rc = ctpn_nms(keep.data(), &nk, dets.data(), n0, 5, kNmsThresh, device_id);
    if (rc) return rc;
    keep.resize(nk);
  } else {
    nms_host(dets.data(), n0, 5, kNmsThresh, keep);
  }

  const int nk = (int)keep.size();
  std::vector<float> kb((size_t)nk * 4), ks(nk);
  for (int i = 0; i < nk; ++i) {
    const float* d = &dets[(size_t)keep[i] * 5];
    kb[4 * i] = d[0]; kb[4 * i + 1] = d[1]; kb[4 * i + 2] = d[2]; kb[4 * i + 3] = d[3]; ks[i] = d[4];
  }
  return connect_lines(kb.data(), ks.data(), nk, im_h, im_w, mode, recs);
}

// graph build + chain extraction + line fit + filter_boxes on proposals that already went through
// detect()'s score filter, sort and NMS (rows in descending score order)
int connect_lines(const float* kept_boxes, const float* kept_scores, int n, int im_h, int im_w, int mode,
                  std::vector<double>& recs) {
  recs.clear();
  if (n < 0 || im_h <= 0 || im_w <= 0) return fail(CTPN_ERR_ARG, "connect_lines: bad size");
  if (mode != CTPN_MODE_H && mode != CTPN_MODE_O) return fail(CTPN_ERR_ARG, "connect_lines: mode must be H(0) or O(1)");
  Props p;
  p.im_w = im_w;
  p.x1.resize(n); p.y1.resize(n); p.x2.resize(n); p.y2.resize(n); p.h.resize(n); p.s.resize(n);
  p.table.assign((size_t)im_w, {});
  for (int i = 0; i < n; ++i) {
    const float* d = kept_boxes + (size_t)i * 4;
    p.x1[i] = d[0]; p.y1[i] = d[1]; p.x2[i] = d[2]; p.y2[i] = d[3]; p.s[i] = kept_scores[i];
    p.h[i] = d[3] - d[1] + 1.0f;
    const int col = (int)d[0];
    if (col < 0 || col >= im_w) return fail(CTPN_ERR_ARG, "text_lines: proposal x1 outside the image (reference raises IndexError)");
    p.table[col].push_back(i);
  }

  // build_graph: at most one out-edge per node
  std::vector<int> succ_of(n, -1);
  std::vector<char> has_in(n, 0);
  std::vector<int> cand, prec;
  for (int i = 0; i < n; ++i) {
    successions(p, i, cand);
    if (cand.empty()) continue;
    int best = cand[0];
    for (int c : cand)
      if (p.s[c] > p.s[best]) best = c;  // np.argmax: first maximum
    precursors(p, best, prec);
    float pmax = -INFINITY;
    for (int c : prec) pmax = std::max(pmax, p.s[c]);
    if (!prec.empty() && p.s[i] >= pmax) { succ_of[i] = best; has_in[best] = 1; }
  }

  // sub_graphs_connected + get_text_lines
  const float wl = (float)(im_w - 1), hl = (float)(im_h - 1);
  std::vector<double> all;  // 9 per line, before filter_boxes
  std::vector<float> X, Y1, Y2, XC, YC, SV, HV;
  for (int i = 0; i < n; ++i) {
    if (has_in[i] || succ_of[i] < 0) continue;
    std::vector<int> chain;
    for (int v = i; v >= 0; v = succ_of[v]) chain.push_back(v);
    X.clear(); Y1.clear(); Y2.clear(); XC.clear(); YC.clear(); SV.clear(); HV.clear();
    float x0 = INFINITY, x1m = -INFINITY;
    for (int v : chain) {
      X.push_back(p.x1[v]); Y1.push_back(p.y1[v]); Y2.push_back(p.y2[v]);
      XC.push_back((p.x1[v] + p.x2[v]) / 2.0f);
      YC.push_back((p.y1[v] + p.y2[v]) / 2.0f);
      x0 = std::min(x0, p.x1[v]); x1m = std::max(x1m, p.x2[v]);
      SV.push_back(p.s[v]);
      HV.push_back(p.y2[v] - p.y1[v]);
    }
    const float ssum = np_sum_f32(SV.data(), SV.size()), hsum = np_sum_f32(HV.data(), HV.size());
    const float offset = (p.x2[chain[0]] - p.x1[chain[0]]) * 0.5f;
    float lt, rt, lb, rb;
    fit_y(X, Y1, x0 + offset, x1m - offset, lt, rt);
    fit_y(X, Y2, x0 + offset, x1m - offset, lb, rb);
    const float score = ssum / (float)chain.size();
    const float top = std::min(lt, rt), bot = std::max(lb, rb);
    double rec[9];
    if (mode == CTPN_MODE_H) {
      const float xmin = clampf(x0, 0.f, wl), xmax = clampf(x1m, 0.f, wl);
      const float ymin = clampf(top, 0.f, hl), ymax = clampf(bot, 0.f, hl);
      const float sc = clampf(score, 0.f, wl);  // other.clip_boxes also clips column 4 (reference quirk, harmless)
      rec[0] = xmin; rec[1] = ymin; rec[2] = xmax; rec[3] = ymin;
      rec[4] = xmin; rec[5] = ymax; rec[6] = xmax; rec[7] = ymax; rec[8] = sc;
    } else {
      float k, b;
      polyfit1(XC, YC, k, b);
      const float height = hsum / (float)chain.size() + 2.5f;
      const float b1 = b - height / 2.0f, b2 = b + height / 2.0f;
      float px1 = x0, py1 = k * x0 + b1;
      float px2 = x1m, py2 = k * x1m + b1;
      float px3 = x0, py3 = k * x0 + b2;
      float px4 = x1m, py4 = k * x1m + b2;
      const float disX = px2 - px1, disY = py2 - py1;
      const float width = std::sqrt(disX * disX + disY * disY);
      const float fTmp0 = py3 - py1;
      const float fTmp1 = fTmp0 * disY / width;
      const float dx = std::fabs(fTmp1 * disX / width);
      const float dy = std::fabs(fTmp1 * disY / width);
      if (k < 0) { px1 -= dx; py1 += dy; px4 += dx; py4 -= dy; }
      else { px2 += dx; py2 += dy; px3 -= dx; py3 -= dy; }
      rec[0] = px1; rec[1] = py1; rec[2] = px2; rec[3] = py2;
      rec[4] = px3; rec[5] = py3; rec[6] = px4; rec[7] = py4; rec[8] = score;
    }
    all.insert(all.end(), rec, rec + 9);
  }

  // filter_boxes (float64 arithmetic on the float64 records)
  for (size_t i = 0; i + 9 <= all.size(); i += 9) {
    const double* b = &all[i];
    const double heights = (std::fabs(b[5] - b[1]) + std::fabs(b[7] - b[3])) / 2.0 + 1;
    const double widths = (std::fabs(b[2] - b[0]) + std::fabs(b[6] - b[4])) / 2.0 + 1;
    if (widths / heights > kMinRatio && b[8] > kLineMinScore && widths > kMinWidth) recs.insert(recs.end(), b, b + 9);
  }
  return CTPN_OK;
}

}  // namespace ctpn

// ---------------------------------------------------------------------------------------------
// draw_boxes (reference ctpn/demo.py:28-52) in C++: the res_<stem>.txt writer and the outline rasteriser.
// ---------------------------------------------------------------------------------------------
namespace {
// demo.py:32 compares SCALARS (np.linalg.norm of box[0] - box[1], box[3] - box[0]): reproduced, SURVEY A.5 iv
inline bool skipped(const double* b) { return std::fabs(b[0] - b[1]) < 5.0 || std::fabs(b[3] - b[0]) < 5.0; }
}  // namespace

// The connector's constants as compiled into this library (TextLineCfg of the reference, lib/text_connector/text_connect_cfg.py:4-12, minus
// SCALE / MAX_SCALE, which the Python side reads): the reference reads them at run time, so a caller that edits its Config expects an
// effect -- lib/text_connector/detectors.py compares its Config with these and fails loudly instead of ignoring the edit.
extern "C" int ctpn_connector_constants(double* out8) {
  if (!out8) return ctpn::fail(CTPN_ERR_ARG, "ctpn_connector_constants: null pointer");
  const double v[8] = {ctpn::kMinWidth, ctpn::kMinRatio, ctpn::kLineMinScore, (double)ctpn::kMaxGap, (double)ctpn::kMinScore, (double)ctpn::kNmsThresh,
                       (double)ctpn::kMinVOverlaps, (double)ctpn::kMinSizeSim};
  std::memcpy(out8, v, sizeof(v));
  return CTPN_OK;
}

extern "C" int ctpn_result_text(const double* recs, int n_lines, double scale, char* out, size_t capacity, size_t* bytes_out, int* lines_out) {
  if ((n_lines > 0 && !recs) || !bytes_out || !(scale > 0.0)) return ctpn::fail(CTPN_ERR_ARG, "ctpn_result_text: bad argument");
  std::string txt;
  int written = 0;
  for (int i = 0; i < n_lines; ++i) {
    const double* b = recs + (size_t)i * 9;
    if (skipped(b)) continue;
    long long xs[4], ys[4];
    for (int k = 0; k < 8; ++k) {                  // Python's int() raises on nan / inf (demo.py:43-46 would stop there); beyond 2^63 the cast is undefined
      const double v = b[k] / scale;
      if (!std::isfinite(v) || std::fabs(v) >= 9.0e18) return ctpn::fail(CTPN_ERR_ARG, "ctpn_result_text: record " + std::to_string(i) + " has a non-finite coordinate");
    }
    for (int k = 0; k < 4; ++k) { xs[k] = (long long)(b[2 * k] / scale); ys[k] = (long long)(b[2 * k + 1] / scale); }   // int(): truncation (demo.py:43-46)
    char line[128];
    const int len = std::snprintf(line, sizeof(line), "%lld,%lld,%lld,%lld\r\n", *std::min_element(xs, xs + 4), *std::min_element(ys, ys + 4),
                                  *std::max_element(xs, xs + 4), *std::max_element(ys, ys + 4));
    txt.append(line, (size_t)len);
    ++written;
  }
  *bytes_out = txt.size();
  if (lines_out) *lines_out = written;
  if (!out) return CTPN_OK;                      // size query
  if (capacity < txt.size()) return ctpn::fail(CTPN_ERR_CAPACITY, "ctpn_result_text: output buffer too small");
  std::memcpy(out, txt.data(), txt.size());
  return CTPN_OK;
}

extern "C" int ctpn_write_result_file(const char* path, const double* recs, int n_lines, double scale, int* lines_out) {
  if (!path) return ctpn::fail(CTPN_ERR_ARG, "ctpn_write_result_file: path is null");
  size_t bytes = 0;
  int rc = ctpn_result_text(recs, n_lines, scale, nullptr, 0, &bytes, lines_out);
  if (rc) return rc;
  std::vector<char> buf(bytes ? bytes : 1);
  if ((rc = ctpn_result_text(recs, n_lines, scale, buf.data(), buf.size(), &bytes, lines_out))) return rc;
  std::FILE* f = std::fopen(path, "wb");         // bytes as they are: "\r\n" stays "\r\n" (the reference opens in text mode on Linux)
  if (!f) return ctpn::fail(CTPN_ERR_ARG, std::string("ctpn_write_result_file: cannot open ") + path);
  const bool ok = bytes == 0 || std::fwrite(buf.data(), 1, bytes, f) == bytes;
  if (std::fclose(f) != 0 || !ok) return ctpn::fail(CTPN_ERR_ARG, std::string("ctpn_write_result_file: write failed: ") + path);
  return CTPN_OK;
}

// Outlines of the kept lines into a BGR uint8 image, as ctpn/demo.py draws them with cv2.line(..., color, 2): green for score >= 0.9,
// blue otherwise; a dense thick-line rasteriser (cv2's exact anti-aliasing-free Bresenham is OpenCV's: parity with it unpinned).
extern "C" int ctpn_draw_boxes(uint8_t* img_bgr, int h, int w, const double* recs, int n_lines) {
  if (!img_bgr || h <= 0 || w <= 0 || (n_lines > 0 && !recs)) return ctpn::fail(CTPN_ERR_ARG, "ctpn_draw_boxes: bad argument");
  auto line = [&](long long x0, long long y0, long long x1, long long y1, const uint8_t (&col)[3]) {
    const long long n = std::max(std::llabs(x1 - x0), std::llabs(y1 - y0)) + 1;
    // only the samples that can touch the image are visited: a line whose ends lie far outside (hostile or degenerate records) must not cost
    // its full length. Sample i sits at a + i * step per axis; the range of i with -2 <= position <= size + 1 is widened by two samples and
    // every pixel is still tested against the image below, so what is drawn does not depend on this.
    long long lo = 0, hi = n - 1;
    auto clip = [&](long long a, long long b, long long size) {
      if (n <= 1 || a == b) { if (a < -2 || a > size + 1) hi = -1; return; }
      const double step = ((double)b - (double)a) / (double)(n - 1);
      double t0 = (-2.0 - (double)a) / step, t1 = ((double)size + 1.0 - (double)a) / step;
      if (t0 > t1) std::swap(t0, t1);
      if (t1 < 0.0 || t0 > (double)(n - 1)) { hi = -1; return; }
      lo = std::max(lo, (long long)std::floor(std::max(t0, 0.0)) - 2);
      hi = std::min(hi, (long long)std::ceil(std::min(t1, (double)(n - 1))) + 2);
    };
    clip(x0, x1, w);
    clip(y0, y1, h);
    lo = std::max(lo, 0LL);
    hi = std::min(hi, n - 1);
    for (long long i = lo; i <= hi; ++i) {
      // np.rint(np.linspace(a, b, n))[i]: a + i * step, the last sample exactly b; round half to even
      const double fx = (n > 1 && i == n - 1) ? (double)x1 : (double)x0 + (double)i * (n > 1 ? ((double)x1 - (double)x0) / (double)(n - 1) : 0.0);
      const double fy = (n > 1 && i == n - 1) ? (double)y1 : (double)y0 + (double)i * (n > 1 ? ((double)y1 - (double)y0) / (double)(n - 1) : 0.0);
      const long long cx = (long long)std::nearbyint(fx), cy = (long long)std::nearbyint(fy);
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const long long yy = cy + dy, xx = cx + dx;
          if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
          uint8_t* p = img_bgr + ((size_t)yy * w + (size_t)xx) * 3;
          p[0] = col[0]; p[1] = col[1]; p[2] = col[2];
        }
    }
  };
  for (int i = 0; i < n_lines; ++i) {
    const double* b = recs + (size_t)i * 9;
    if (skipped(b)) continue;
    const uint8_t green[3] = {0, 255, 0}, blue[3] = {255, 0, 0};
    const uint8_t (&col)[3] = b[8] >= 0.9 ? green : blue;
    bool sane = true;      // NaN, inf or coordinates beyond any image: nothing to draw (and (long long) of them would be undefined)
    for (int k = 0; k < 8; ++k) sane = sane && std::isfinite(b[k]) && std::fabs(b[k]) < 1e12;
    if (!sane) continue;
    const long long px[4] = {(long long)b[0], (long long)b[2], (long long)b[6], (long long)b[4]};
    const long long py[4] = {(long long)b[1], (long long)b[3], (long long)b[7], (long long)b[5]};
    for (int k = 0; k < 4; ++k) line(px[k], py[k], px[(k + 1) & 3], py[(k + 1) & 3], col);
  }
  return CTPN_OK;
}

