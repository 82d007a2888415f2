// libmbx: the pixel half of the training input augmentation and of the detection input (row F3)
// on the GPU.
//
// The reference's input graph (inputs.py:264-351) crops, resizes with a randomly drawn tf.image.ResizeMethod,
// distorts colours, flips and rescales every image with TF's CPU image ops.  The host side of this build
// (multibox_amd/inputs.py) draws all random decisions -- crop window, method, colour op order and arguments, flip --
// and decodes the JPEG; these kernels do the arithmetic, one thread per output pixel:
//
//   augment_resize_kernel   uint8 crop [h][w][3] -> float32 [S][S][3] in [0,1]  (bilinear / nearest / bicubic / area,
//                           TF 0.11's legacy coordinates: src = dst * in/out, no half-pixel offset)
//   augment_sums_kernel     per-image channel sums of the picture as it stands when adjust_contrast runs (only launched
//                           when some image has a contrast op)
//   augment_color_kernel    the colour ops in float64 (the host restatement's arithmetic), clip, flip, (x - 0.5) * 2
//
// All HBM-trivial (a batch of 64 is 68 MB out, a few tens of MB in).  Built with -ffp-contract=off: the float32
// resize arithmetic keeps the host restatement's rounding sequence, so the resize is bit-identical to it and the colour
// ops differ only through the summation order of the contrast mean.
#include "common.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int kSumBlocks = 64;          // partial sums per image for the contrast mean

__device__ __forceinline__ float src_px(const uint8_t* p, int w, int y, int x, int c) {
  return __fmul_rn((float)p[((size_t)y * w + x) * 3 + c], 1.0f / 255.0f);      // convert_image_dtype(uint8 -> float32)
}

// Keys cubic convolution coefficients, A = -0.75 (tf.image.resize_bicubic), taps at -1, 0, +1, +2
__device__ __forceinline__ void cubic_weights(double t, double w[4]) {
  const double a = -0.75;
  w[0] = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a;
  w[1] = ((a + 2) * t - (a + 3)) * t * t + 1;
  w[2] = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1;
  w[3] = ((a * (2 - t) - 5 * a) * (2 - t) + 8 * a) * (2 - t) - 4 * a;
}

// resize_area's weights along one axis for output index o: taps source pixels from i0 on, weight k = covered length of
// pixel i0+k by [o*scale, (o+1)*scale), normalised, rounded to float32
struct AreaAxis {
  int i0, taps, n_in;
  double lo, hi, inv_sum;
  __device__ void init(int o, int n_in_, int n_out) {
    n_in = n_in_;
    const double scale = (double)n_in / (double)n_out;
    lo = o * scale; hi = (o + 1) * scale;
    i0 = (int)floor(lo);
    taps = (int)ceil(scale) + 1;
    double s = 0.0;
    for (int k = 0; k < taps; ++k) s += raw(k);
    inv_sum = fmax(s, 1e-12);
  }
  __device__ double raw(int k) const {
    const int idx = i0 + k;
    const double w = fmin(hi, idx + 1.0) - fmax(lo, (double)idx);
    return (idx < n_in && w > 0) ? w : 0.0;
  }
  __device__ float weight(int k) const { return (float)(raw(k) / inv_sum); }
  __device__ int index(int k) const { return min(i0 + k, n_in - 1); }
};

__global__ void __launch_bounds__(kThreads)
augment_resize_kernel(const uint8_t* __restrict__ src, const mbx_augment_item* __restrict__ items, int S,
                      float* __restrict__ tmp) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * kThreads + threadIdx.x;
  if (pix >= S * S) return;
  const mbx_augment_item it = items[b];
  const int oy = pix / S, ox = pix - oy * S;
  const int H = it.src_h, W = it.src_w;
  float* out = tmp + ((size_t)b * S * S + pix) * 3;
  const uint8_t* p = src + it.src_offset;
  if (it.method == 4) {                                   // prepared on the host: float32 [S][S][3] in [0,1]
    const float* f = reinterpret_cast<const float*>(p) + (size_t)pix * 3;
    out[0] = f[0]; out[1] = f[1]; out[2] = f[2];
    return;
  }
  const float sy = (float)((double)H / (double)S), sx = (float)((double)W / (double)S);
  if (it.method == 0) {                                   // legacy bilinear
    const float ys = __fmul_rn((float)oy, sy), xs = __fmul_rn((float)ox, sx);
    const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float yl = __fsub_rn(ys, (float)y0), xl = __fsub_rn(xs, (float)x0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tl = src_px(p, W, y0, x0, c), tr = src_px(p, W, y0, x1, c);
      const float bl = src_px(p, W, y1, x0, c), br = src_px(p, W, y1, x1, c);
      const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), xl));
      const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), xl));
      out[c] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), yl));
    }
  } else if (it.method == 1) {                            // nearest neighbour
    const int y = min((int)floorf(__fmul_rn((float)oy, sy)), H - 1);
    const int x = min((int)floorf(__fmul_rn((float)ox, sx)), W - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = src_px(p, W, y, x, c);
  } else if (it.method == 2) {                            // bicubic: rows first, then columns, in float64
    const float ys = __fmul_rn((float)oy, sy), xs = __fmul_rn((float)ox, sx);
    const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
    double wy[4], wx[4];
    cubic_weights((double)ys - (double)y0, wy);
    cubic_weights((double)xs - (double)x0, wx);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = min(max(x0 - 1 + j, 0), W - 1);
        double col = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) col = col + (double)src_px(p, W, min(max(y0 - 1 + k, 0), H - 1), x, c) * wy[k];
        v = v + col * wx[j];
      }
      out[c] = (float)v;
    }
  } else {                                                // area: rows first (float32), then columns (float32)
    AreaAxis ay, ax;
    ay.init(oy, H, S);
    ax.init(ox, W, S);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < ax.taps; ++j) {
      const int x = ax.index(j);
      const float wxj = ax.weight(j);
      float col[3];
      {
        const int y = ay.index(0);
        const float w = ay.weight(0);
#pragma unroll
        for (int c = 0; c < 3; ++c) col[c] = __fmul_rn(src_px(p, W, y, x, c), w);
      }
      for (int k = 1; k < ay.taps; ++k) {
        const int y = ay.index(k);
        const float w = ay.weight(k);
#pragma unroll
        for (int c = 0; c < 3; ++c) col[c] = __fadd_rn(col[c], __fmul_rn(src_px(p, W, y, x, c), w));
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
        acc[c] = j == 0 ? __fmul_rn(col[c], wxj) : __fadd_rn(acc[c], __fmul_rn(col[c], wxj));
    }
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
  }
}

// ---- colour ops, float64, per pixel (tf.image.adjust_* as multibox_amd/inputs.py restates them)
__device__ __forceinline__ double pymod1(double x) {     // numpy's x % 1.0
  double r = fmod(x, 1.0);
  if (r != 0.0 && r < 0.0) r += 1.0;
  return r;
}
__device__ __forceinline__ double clip01(double x) { return fmin(fmax(x, 0.0), 1.0); }

__device__ __forceinline__ void rgb_to_hsv(const double c[3], double hsv[3]) {
  const double r = c[0], g = c[1], b = c[2];
  const double mx = fmax(r, fmax(g, b)), mn = fmin(r, fmin(g, b));
  const double d = mx - mn;
  const double s = mx > 0 ? d / mx : 0.0;
  const double dz = d > 0 ? d : 1.0;
  double h = (mx == r) ? (g - b) / dz : (mx == g) ? 2.0 + (b - r) / dz : 4.0 + (r - g) / dz;
  h = d > 0 ? pymod1(h / 6.0) : 0.0;
  hsv[0] = h; hsv[1] = s; hsv[2] = mx;
}
__device__ __forceinline__ void hsv_to_rgb(const double hsv[3], double c[3]) {
  const double h = hsv[0], s = hsv[1], v = hsv[2];
  const double dh = h * 6.0;
  const double dr = clip01(fabs(dh - 3.0) - 1.0);
  const double dg = clip01(2.0 - fabs(dh - 2.0));
  const double db = clip01(2.0 - fabs(dh - 4.0));
  c[0] = (1 - s + s * dr) * v; c[1] = (1 - s + s * dg) * v; c[2] = (1 - s + s * db) * v;
}
__device__ __forceinline__ void pixel_op(int op, double arg, double c[3]) {
  if (op == 0) {                          // adjust_brightness
    c[0] += arg; c[1] += arg; c[2] += arg;
  } else if (op == 1) {                   // adjust_saturation
    double hsv[3];
    rgb_to_hsv(c, hsv);
    hsv[1] = clip01(hsv[1] * arg);
    hsv_to_rgb(hsv, c);
  } else {                                // adjust_hue
    double hsv[3];
    rgb_to_hsv(c, hsv);
    hsv[0] = pymod1(hsv[0] + arg);
    hsv_to_rgb(hsv, c);
  }
}
__device__ __forceinline__ int contrast_index(const mbx_augment_item& it) {
  for (int k = 0; k < it.n_ops; ++k)
    if (it.op[k] == 3) return k;
  return -1;
}

__global__ void __launch_bounds__(kThreads)
augment_sums_kernel(const mbx_augment_item* __restrict__ items, int S, const float* __restrict__ tmp,
                    double* __restrict__ partial /*[B][kSumBlocks][3]*/) {
  const int b = blockIdx.y;
  const mbx_augment_item& it = items[b];                 // read in place (uniform scalar loads): a by-value copy indexed by a
  const int ci = contrast_index(it);                     // run-time k lives in scratch memory
  if (ci < 0) return;
  double s[3] = {0.0, 0.0, 0.0};
  const float* img = tmp + (size_t)b * S * S * 3;
  for (int pix = blockIdx.x * kThreads + threadIdx.x; pix < S * S; pix += kSumBlocks * kThreads) {
    double c[3] = {(double)img[pix * 3], (double)img[pix * 3 + 1], (double)img[pix * 3 + 2]};
    for (int k = 0; k < ci; ++k) pixel_op(it.op[k], it.arg[k], c);
    s[0] += c[0]; s[1] += c[1]; s[2] += c[2];
  }
  __shared__ double red[kThreads / 64][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) s[c] = wave_sum(s[c]);
  if (mbx_lane() == 0) { red[threadIdx.x >> 6][0] = s[0]; red[threadIdx.x >> 6][1] = s[1]; red[threadIdx.x >> 6][2] = s[2]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += red[w][threadIdx.x];
    partial[((size_t)b * kSumBlocks + blockIdx.x) * 3 + threadIdx.x] = t;
  }
}

__global__ void __launch_bounds__(kThreads)
augment_color_kernel(const mbx_augment_item* __restrict__ items, int S, const float* __restrict__ tmp,
                     const double* __restrict__ partial, float* __restrict__ out) {
  const int b = blockIdx.y;
  const mbx_augment_item& it = items[b];
  const int ci = contrast_index(it);
  __shared__ double mean[3];
  if (ci >= 0) {                                                   // uniform over the block: `it` is per image
    if (threadIdx.x < 3) {
      double t = 0.0;
      for (int k = 0; k < kSumBlocks; ++k) t += partial[((size_t)b * kSumBlocks + k) * 3 + threadIdx.x];
      mean[threadIdx.x] = t / ((double)S * (double)S);
    }
    __syncthreads();
  }
  const int pix = blockIdx.x * kThreads + threadIdx.x;
  if (pix >= S * S) return;
  const int oy = pix / S, ox = pix - oy * S;
  const float* in = tmp + ((size_t)b * S * S + pix) * 3;
  float v[3] = {in[0], in[1], in[2]};
  if (it.n_ops > 0) {
    double c[3] = {(double)v[0], (double)v[1], (double)v[2]};
    for (int k = 0; k < it.n_ops; ++k) {
      if (it.op[k] == 3) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) c[ch] = (c[ch] - mean[ch]) * it.arg[k] + mean[ch];
      } else {
        pixel_op(it.op[k], it.arg[k], c);
      }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = (float)clip01(c[ch]);
  }
  const int dx = it.flip ? S - 1 - ox : ox;                        // image[:, ::-1]
  float* o = out + ((size_t)b * S * S + (size_t)oy * S + dx) * 3;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) o[ch] = __fmul_rn(__fsub_rn(v[ch], 0.5f), 2.0f);   // inputs.py:350-351
}

// Detection input (detect.py:181-281): every patch is a window of the decoded image -- or of its mirror image -- that
// has been scaled to [-1,1] FIRST and is then resized with the legacy bilinear kernel.
__global__ void __launch_bounds__(kThreads)
extract_patches_kernel(const uint8_t* __restrict__ src, const mbx_patch_item* __restrict__ items, int S,
                       float* __restrict__ out) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * kThreads + threadIdx.x;
  if (pix >= S * S) return;
  const mbx_patch_item it = items[b];
  const int oy = pix / S, ox = pix - oy * S;
  const int H = it.win_h, W = it.win_w;
  const uint8_t* p = src + it.src_offset;
  const float sy = (float)((double)H / (double)S), sx = (float)((double)W / (double)S);
  const float ys = __fmul_rn((float)oy, sy), xs = __fmul_rn((float)ox, sx);
  const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float yl = __fsub_rn(ys, (float)y0), xl = __fsub_rn(xs, (float)x0);
  const int gy0 = it.win_y + y0, gy1 = it.win_y + y1;
  int gx0 = it.win_x + x0, gx1 = it.win_x + x1;
  if (it.flip_source) { gx0 = it.img_w - 1 - gx0; gx1 = it.img_w - 1 - gx1; }      // image[:, ::-1]
  float* o = out + ((size_t)b * S * S + pix) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    auto px = [&](int y, int x) {                                                  // (image - 0.5) * 2, detect.py:181-182
      return __fmul_rn(__fsub_rn(src_px(p, it.img_w, y, x, c), 0.5f), 2.0f);
    };
    const float tl = px(gy0, gx0), tr = px(gy0, gx1), bl = px(gy1, gx0), br = px(gy1, gx1);
    const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), xl));
    const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), xl));
    o[c] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), yl));
  }
}

}  // namespace

extern "C" int mbx_extract_patches(const uint8_t* src, const mbx_patch_item* items, int n, int S, float* out,
                                   mbx_stream_t stream) {
  if (n < 0 || S <= 0 || n > 65535) return MBX_ERR_INVALID_ARG;
  if (n == 0) return MBX_OK;
  if (!src || !items || !out) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(extract_patches_kernel, dim3((S * S + kThreads - 1) / kThreads, n), dim3(kThreads), 0, mbx_s(stream),
                     src, items, S, out);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" size_t mbx_augment_workspace_bytes(int B, int S) {
  if (B <= 0 || S <= 0) return 0;
  const size_t tmp = (((size_t)B * S * S * 3 * sizeof(float)) + 255) & ~(size_t)255;
  return tmp + (size_t)B * kSumBlocks * 3 * sizeof(double);
}

extern "C" int mbx_augment_batch(const uint8_t* src, const mbx_augment_item* items, int B, int S, int any_contrast,
                                 float* out, void* workspace, mbx_stream_t stream) {
  if (B < 0 || S <= 0) return MBX_ERR_INVALID_ARG;
  if (B == 0) return MBX_OK;
  if (!src || !items || !out || !workspace) return MBX_ERR_INVALID_ARG;
  if (B > 65535) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  float* tmp = static_cast<float*>(workspace);
  const size_t tmp_bytes = (((size_t)B * S * S * 3 * sizeof(float)) + 255) & ~(size_t)255;
  double* partial = reinterpret_cast<double*>(static_cast<char*>(workspace) + tmp_bytes);
  const dim3 grid((S * S + kThreads - 1) / kThreads, B);
  hipLaunchKernelGGL(augment_resize_kernel, grid, dim3(kThreads), 0, mbx_s(stream), src, items, S, tmp);
  MBX_LAUNCH_CHECK();
  if (any_contrast) {
    hipLaunchKernelGGL(augment_sums_kernel, dim3(kSumBlocks, B), dim3(kThreads), 0, mbx_s(stream), items, S, tmp, partial);
    MBX_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(augment_color_kernel, grid, dim3(kThreads), 0, mbx_s(stream), items, S, tmp, partial, out);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}
