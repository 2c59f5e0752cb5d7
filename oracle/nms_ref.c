/* ORACLE (test infrastructure only -- never linked into libctpn_hip.so or imported by the product path).
 *
 * Plain-C restatement of the reference's CPU NMS, lib/utils/cython_nms.pyx:17-68 (`nms`):
 * O(N^2) greedy suppression in descending-score order, "+1" areas, fp32 arithmetic. The suppression test is
 * selectable because the reference's three variants differ only there:
 *     predicate 0: (double)ovr >= thresh_d     cython_nms.pyx:65  (thresh is a Python double)
 *     predicate 1: ovr > (float)thresh         nms_kernel.cu:71 / nms_wrapper.py:45  (canonical for the device path)
 * The reference's generated cython_nms.c (Cython 0.25) does not compile against CPython 3.10 / numpy 2, and
 * nms_kernel.cu needs nvcc, so there is no oracle/_ref build of the original sources.
 * Pinned by tests/test_oracle.py against the keep lists the reference's py_cpu_nms produced (tests/golden).
 * Build: make -C oracle   ->  oracle/_build/liboracle_nms.so
 */
#include <stdlib.h>

static float fmax2(float a, float b) { return a >= b ? a : b; }
static float fmin2(float a, float b) { return a <= b ? a : b; }

/* dets: n x 5 rows [x1,y1,x2,y2,score]; order: n indices, descending score (caller decides the tie order);
 * keep: capacity n; returns the number kept. */
int oracle_nms(const float* dets, const long* order, int n, double thresh, int predicate, long* keep) {
  char* suppressed = (char*)calloc((size_t)(n > 0 ? n : 1), 1);
  float* areas = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  int nk = 0;
  const float thr_f = (float)thresh;
  for (int i = 0; i < n; ++i) {
    const float* d = dets + (size_t)i * 5;
    areas[i] = (d[2] - d[0] + 1) * (d[3] - d[1] + 1);
  }
  for (int _i = 0; _i < n; ++_i) {
    const long i = order[_i];
    if (suppressed[i]) continue;
    keep[nk++] = i;
    const float ix1 = dets[i * 5 + 0], iy1 = dets[i * 5 + 1], ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
    const float iarea = areas[i];
    for (int _j = _i + 1; _j < n; ++_j) {
      const long j = order[_j];
      if (suppressed[j]) continue;
      const float xx1 = fmax2(ix1, dets[j * 5 + 0]);
      const float yy1 = fmax2(iy1, dets[j * 5 + 1]);
      const float xx2 = fmin2(ix2, dets[j * 5 + 2]);
      const float yy2 = fmin2(iy2, dets[j * 5 + 3]);
      const float w = fmax2(0.0f, xx2 - xx1 + 1);
      const float h = fmax2(0.0f, yy2 - yy1 + 1);
      const float inter = w * h;
      const float ovr = inter / (iarea + areas[j] - inter);
      const int hit = predicate == 0 ? ((double)ovr >= thresh) : (ovr > thr_f);
      if (hit) suppressed[j] = 1;
    }
  }
  free(suppressed);
  free(areas);
  return nk;
}
