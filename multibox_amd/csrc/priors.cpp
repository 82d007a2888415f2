// HOST: prior box generation, float64, bit-exact with priors.generate_priors
// (reference priors.py:185-314).  Build with -ffp-contract=off.
#include <cstdint>
#include <cmath>
#include <algorithm>
#include "../../include/mbx.h"

namespace {
// One box: priors.py:264-310 (gridded) == 206-258 (1x1 head) with a = 1.
void one_prior(double cx, double cy, double scale, double a, bool restrict_, double* o) {
  const double ra = std::sqrt(a);
  const double w = scale * ra, h = scale / ra;
  double x1 = cx - (w / 2.0), x2 = cx + (w / 2.0);
  double y1 = cy - (h / 2.0), y2 = cy + (h / 2.0);
  if (restrict_) {
    double wt = std::max(std::fabs(std::min(0.0, x1)), std::fabs(std::min(0.0, 1.0 - x2)));
    double ht = std::max(std::fabs(std::min(0.0, y1)), std::fabs(std::min(0.0, 1.0 - y2)));
    const double trim = std::max(wt, ht);
    // priors.py:289-294: the trim is applied along h>w, not along the overshooting axis.
    if (h > w) { wt = trim * a; ht = trim; } else { wt = trim; ht = trim / a; }
    const double xa = x1 + wt, xb = x2 - wt, ya = y1 + ht, yb = y2 - ht;
    x1 = std::min(xa, xb); x2 = std::max(xa, xb);
    y1 = std::min(ya, yb); y2 = std::max(ya, yb);
  }
  o[0] = std::max(x1, 0.0); o[1] = std::max(y1, 0.0);
  o[2] = std::min(x2, 1.0); o[3] = std::min(y2, 1.0);
}
}  // namespace

extern "C" int mbx_priors_count(int k, const int* grids, int n_grids) {
  if (k < 1 || !grids || n_grids < 1) return MBX_ERR_INVALID_ARG;
  long rows = 0;
  for (int g = 0; g < n_grids; ++g) {
    if (grids[g] < 1) return MBX_ERR_INVALID_ARG;
    rows += grids[g] == 1 ? 1 : (long)grids[g] * grids[g] * k;
  }
  return (int)rows;
}

extern "C" int mbx_generate_priors(const double* ars, int k, double min_scale, double max_scale,
                                   int restrict_to_image_bounds, const int* grids, int n_grids,
                                   double* out) {
  if (!ars || !out || mbx_priors_count(k, grids, n_grids) < 0) return MBX_ERR_INVALID_ARG;
  double* o = out;
  for (int gi = 0; gi < n_grids; ++gi) {
    const int g = grids[gi];
    // priors.py:198-200: min + (max-min)*(i-1)/(num_scales-1), i = 1..num_scales
    const double scale = n_grids > 1 ? min_scale + (max_scale - min_scale) * (double)gi / (double)(n_grids - 1)
                                     : min_scale;
    if (g == 1) { one_prior(0.5, 0.5, scale, 1.0, restrict_to_image_bounds != 0, o); o += 4; continue; }
    for (int i = 0; i < g; ++i)
      for (int j = 0; j < g; ++j) {
        const double cy = (i + 0.5) / g, cx = (j + 0.5) / g;
        for (int a = 0; a < k; ++a) { one_prior(cx, cy, scale, ars[a], restrict_to_image_bounds != 0, o); o += 4; }
      }
  }
  return MBX_OK;
}

// CRC-32C (Castagnoli), slice-by-8: the checksum of TFRecord frames and of TensorFlow checkpoint table blocks
// (multibox_amd/tfrecord.py, tf_checkpoint.py).  Host code; `crc` chains calls (0 for the first).
extern "C" uint32_t mbx_crc32c(const void* data, uint64_t n, uint32_t crc) {
  static uint32_t T[8][256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xff];
    init = true;
  }
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {
    const uint32_t lo = c ^ ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
    c = T[7][lo & 0xff] ^ T[6][(lo >> 8) & 0xff] ^ T[5][(lo >> 16) & 0xff] ^ T[4][lo >> 24] ^
        T[3][p[4]] ^ T[2][p[5]] ^ T[1][p[6]] ^ T[0][p[7]];
    p += 8; n -= 8;
  }
  while (n--) c = T[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return ~c;
}

extern "C" int mbx_version(void) { return 100; }

extern "C" const char* mbx_status_string(int status) {
  switch (status) {
    case MBX_OK: return "ok";
    case MBX_ERR_INVALID_ARG: return "invalid argument";
    case MBX_ERR_UNSUPPORTED: return "unsupported size or configuration";
    case MBX_ERR_LAUNCH: return "HIP launch failed";
    case MBX_ERR_WORKSPACE: return "workspace missing or too small";
    default: return "unknown status";
  }
}
