// utils.cython_bbox.bbox_overlaps (lib/utils/bbox.pyx:15-55) -- a HOST function in the reference (Cython) and here.
// float64, +1 widths, entries stay 0 when the boxes do not intersect.  [n][k] row-major.
#include "mnc_internal.h"

extern "C" int mnc_bbox_overlaps(const double* boxes, int n, const double* query, int k, double* overlaps) {
  MNC_REQUIRE(n >= 0 && k >= 0, "mnc_bbox_overlaps: negative size");
  if (n == 0 || k == 0) { mnc::clear_error(); return MNC_OK; }
  MNC_REQUIRE(boxes && query && overlaps, "mnc_bbox_overlaps: null pointer");
  for (int q = 0; q < k; ++q) {
    const double* qb = query + 4 * (long)q;
    const double qarea = (qb[2] - qb[0] + 1) * (qb[3] - qb[1] + 1);
    for (int i = 0; i < n; ++i) {
      const double* b = boxes + 4 * (long)i;
      double ov = 0.0;
      const double iw = (b[2] < qb[2] ? b[2] : qb[2]) - (b[0] > qb[0] ? b[0] : qb[0]) + 1;
      if (iw > 0) {
        const double ih = (b[3] < qb[3] ? b[3] : qb[3]) - (b[1] > qb[1] ? b[1] : qb[1]) + 1;
        if (ih > 0) {
          const double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + qarea - iw * ih;
          ov = iw * ih / ua;
        }
      }
      overlaps[(long)i * k + q] = ov;
    }
  }
  mnc::clear_error();
  return MNC_OK;
}
