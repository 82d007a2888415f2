// libmbx: prior decode, bipartite matching, loss fwd+bwd, detect post-processing.
// HBM/latency-bound integer+float kernels; one workgroup per image/patch,
// wave64 shuffle reductions.  Built with -ffp-contract=off: the float32 cost arithmetic
// must keep the reference's rounding sequence (loss.py:21-35).
#include "common.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

// ------------------------------------------------------------------ decode + conf
__global__ void __launch_bounds__(kThreads)
decode_conf_kernel(const float4* __restrict__ raw, const float* __restrict__ logits,
                   const float4* __restrict__ priors, int total, int P, float eps_add,
                   float4* __restrict__ decoded, float* __restrict__ conf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (decoded) {
      const float4 r = raw[i], p = priors[i % P];
      decoded[i] = make_float4(__fadd_rn(r.x, p.x), __fadd_rn(r.y, p.y), __fadd_rn(r.z, p.z), __fadd_rn(r.w, p.w));
    }
    if (conf) {
      const float s = 1.0f / (1.0f + expf(-logits[i]));     // model.py:322
      conf[i] = __fadd_rn(s, eps_add);                      // loss.py:74
    }
  }
}

// ----------------------------------------------------------------------- matching
// cost of (prediction, gt) in the reference's float32 operation order, loss.py:35:
//   (alpha/2) * norm(l - g)**2 - log c + log(1-c)          (left to right)
__device__ __forceinline__ double match_cost(const float4 l, const float4 g, float half_alpha,
                                             float lc, float l1c) {
  const float d0 = __fsub_rn(l.x, g.x), d1 = __fsub_rn(l.y, g.y), d2 = __fsub_rn(l.z, g.z), d3 = __fsub_rn(l.w, g.w);
  const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), __fmul_rn(d3, d3));
  const float nrm = __fsqrt_rn(ss);
  float t = __fmul_rn(half_alpha, __fmul_rn(nrm, nrm));
  t = __fsub_rn(t, lc);
  t = __fadd_rn(t, l1c);
  return (double)t;
}

struct Cand {
  double val;
  int j;      // column (prediction); -1 = none
  int free_;  // 1 if the column is unassigned
};

__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {
  if (a.j < 0) return false;
  if (b.j < 0) return true;
  if (a.val < b.val) return true;
  if (a.val > b.val) return false;
  if (a.free_ != b.free_) return a.free_ > b.free_;   // scipy prefers an unassigned column among equals
  return a.j < b.j;
}

__device__ __forceinline__ Cand cand_shfl_xor(const Cand& c, int o) {
  Cand r;
  r.val = __shfl_xor(c.val, o, 64);
  r.j = __shfl_xor(c.j, o, 64);
  r.free_ = __shfl_xor(c.free_, o, 64);
  return r;
}

// One workgroup per image.  Rectangular LSAP on the transposed problem (rows = gt boxes,
// columns = predictions), shortest augmenting path with float64 duals, as scipy's
// linear_sum_assignment (loss.py:40).  LDS: 36 B per prediction + 16 B per gt.  256 threads per image up to 1536 predictions,
// 512 beyond (round 4, tools/match_bench.py on random boxes: P = 3199 / G = 100 812 -> 589 us; P = 646 is fastest at 256:
// 47 us against 84 with one wave -- the column scan, not the barriers, is what an iteration costs; staging the locations in
// LDS changed nothing: they are L1 hits).
__global__ void __launch_bounds__(1024)
match_kernel(const float4* __restrict__ decoded, const float* __restrict__ conf,
             const float4* __restrict__ gt, const int* __restrict__ n_gt, float alpha, int P, int G,
             int* __restrict__ match, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* v = reinterpret_cast<double*>(smem);            // [P] column duals
  double* spc = v + P;                                    // [P] shortest path costs
  double* u = spc + P;                                    // [G] row duals
  float2* lcl = reinterpret_cast<float2*>(u + G);         // [P] {log c, log(1-c)}
  int* path = reinterpret_cast<int*>(lcl + P);            // [P]
  int* row4col = path + P;                                // [P]
  int* SC = row4col + P;                                  // [P]
  int* col4row = SC + P;                                  // [G]
  int* SR = col4row + G;                                  // [G]
  __shared__ Cand wave_best[16];
  __shared__ int sh_i, sh_sink, sh_bad;
  __shared__ double sh_min;

  const int b = blockIdx.x, tid = threadIdx.x;
  const int nt = (int)blockDim.x, nwaves = nt >> 6;       // 256 or 512 threads (mbx_match)
  const float4* loc = decoded + (size_t)b * P;
  const float* cf = conf + (size_t)b * P;
  const float4* g = gt + (size_t)b * G;
  int* mt = match + (size_t)b * P;
  const int n = n_gt[b];
  const float half_alpha = alpha / 2.0f;

  if (tid == 0) sh_bad = 0;
  __syncthreads();
  int bad = 0;
  for (int j = tid; j < P; j += nt) {
    mt[j] = -1;
    const float c = cf[j];
    const float lc = logf(c);                              // loss.py:21
    float w = __fsub_rn(1.0f, c);                          // loss.py:22-24
    if (w > 1.0f) w = 1.0f;
    if (w <= 0.0f) w = 1e-10f;
    const float l1c = logf(w);                             // loss.py:25
    lcl[j] = make_float2(lc, l1c);
    v[j] = 0.0;
    row4col[j] = -1;
    const float4 l = loc[j];
    if (!(isfinite(lc) && isfinite(l1c) && isfinite(l.x) && isfinite(l.y) && isfinite(l.z) && isfinite(l.w))) bad = 1;
  }
  for (int i = tid; i < G; i += nt) {
    u[i] = 0.0;
    col4row[i] = -1;
    if (i < n) {
      const float4 q = g[i];
      if (!(isfinite(q.x) && isfinite(q.y) && isfinite(q.z) && isfinite(q.w))) bad = 1;
    }
  }
  if (bad) sh_bad = 1;
  __syncthreads();
  if (n <= 0) { if (tid == 0) status[b] = 0; return; }
  if (n > P || n > G) { if (tid == 0) status[b] = 1; return; }
  if (sh_bad) { if (tid == 0) status[b] = 2; return; }

  for (int cur = 0; cur < n; ++cur) {
    for (int j = tid; j < P; j += nt) { spc[j] = INFINITY; SC[j] = 0; }
    for (int i = tid; i < n; i += nt) SR[i] = 0;
    if (tid == 0) { sh_i = cur; sh_sink = -1; sh_min = 0.0; }
    __syncthreads();
    while (true) {
      const int i = sh_i;
      const double min_val = sh_min;
      const float4 gi = g[i];
      const double ui = u[i];
      Cand best; best.val = INFINITY; best.j = -1; best.free_ = 0;
      for (int j = tid; j < P; j += nt) {
        if (SC[j]) continue;
        const float2 ll = lcl[j];
        // scipy: r = minVal + cost[i][j] - u[i] - v[j]
        const double r = ((min_val + match_cost(loc[j], gi, half_alpha, ll.x, ll.y)) - ui) - v[j];
        double s = spc[j];
        if (r < s) { path[j] = i; spc[j] = r; s = r; }
        Cand c; c.val = s; c.j = j; c.free_ = row4col[j] < 0;
        if (cand_better(c, best)) best = c;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const Cand other = cand_shfl_xor(best, o);
        if (cand_better(other, best)) best = other;
      }
      if ((tid & 63) == 0) wave_best[tid >> 6] = best;
      __syncthreads();
      if (tid == 0) {
        Cand bb = wave_best[0];
        for (int w = 1; w < nwaves; ++w) if (cand_better(wave_best[w], bb)) bb = wave_best[w];
        if (bb.j < 0 || !(bb.val < INFINITY)) {
          sh_sink = -2;                                    // infeasible
        } else {
          sh_min = bb.val;
          SC[bb.j] = 1;
          SR[i] = 1;
          const int r4c = row4col[bb.j];
          if (r4c < 0) sh_sink = bb.j; else sh_i = r4c;
        }
      }
      __syncthreads();
      if (sh_sink != -1) break;
    }
    const int sink = sh_sink;
    if (sink == -2) { if (tid == 0) status[b] = 2; return; }
    const double min_val = sh_min;
    // dual updates
    for (int i = tid; i < n; i += nt) {
      if (i == cur) u[i] += min_val;
      else if (SR[i]) u[i] += min_val - spc[col4row[i]];
    }
    for (int j = tid; j < P; j += nt)
      if (SC[j]) v[j] -= min_val - spc[j];
    __syncthreads();
    if (tid == 0) {                                        // augment along the path
      int j = sink;
      while (true) {
        const int i = path[j];
        row4col[j] = i;
        const int t = col4row[i];
        col4row[i] = j;
        j = t;
        if (i == cur) break;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) mt[col4row[i]] = i;
  if (tid == 0) status[b] = 0;
}

// ---------------------------------------------------------------------- loss
__global__ void __launch_bounds__(kThreads)
loss_kernel(const float4* __restrict__ decoded, const float* __restrict__ logits, int is_logit,
            const float4* __restrict__ gt, const int* __restrict__ match, float alpha, float grad_scale,
            int P, int G, double* __restrict__ partial /*[B,2]*/, float4* __restrict__ d_locs,
            float* __restrict__ d_logits) {
  const int b = blockIdx.x, tid = threadIdx.x;
  double loc_acc = 0.0, conf_acc = 0.0;
  for (int p = tid; p < P; p += kThreads) {
    const size_t i = (size_t)b * P + p;
    const int m = match[i];
    const float z = logits[i];
    const float s = is_logit ? 1.0f / (1.0f + expf(-z)) : z;
    const float c = __fadd_rn(s, 1e-10f);                  // loss.py:74
    const float ds = is_logit ? s * (1.0f - s) : 1.0f;
    float4 dl = make_float4(0.f, 0.f, 0.f, 0.f);
    float dz;
    if (m >= 0) {
      const float4 l = decoded[i], q = gt[(size_t)b * G + m];
      const float d0 = __fsub_rn(l.x, q.x), d1 = __fsub_rn(l.y, q.y), d2 = __fsub_rn(l.z, q.z), d3 = __fsub_rn(l.w, q.w);
      loc_acc += (double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2 + (double)d3 * d3;   // loss.py:100
      conf_acc -= (double)logf(c);                                                      // loss.py:101
      const float a = alpha * grad_scale;
      dl = make_float4(a * d0, a * d1, a * d2, a * d3);
      dz = -ds / c;
    } else {
      const float w = __fadd_rn(__fsub_rn(1.0f, c), 1e-10f);
      conf_acc -= (double)logf(w);
      dz = ds / w;
    }
    if (d_locs) d_locs[i] = dl;
    if (d_logits) d_logits[i] = dz * grad_scale;
  }
  loc_acc = wave_sum(loc_acc);
  conf_acc = wave_sum(conf_acc);
  __shared__ double red[kWaves][2];
  if ((tid & 63) == 0) { red[tid >> 6][0] = loc_acc; red[tid >> 6][1] = conf_acc; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < kWaves; ++w) { a += red[w][0]; c += red[w][1]; }
    partial[2 * b] = a;
    partial[2 * b + 1] = c;
  }
}

__global__ void loss_final_kernel(const double* __restrict__ partial, int B, float alpha, float* __restrict__ loss2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int b = 0; b < B; ++b) { a += partial[2 * b]; c += partial[2 * b + 1]; }
    loss2[0] = alpha * (float)(a * 0.5);                   // alpha * tf.nn.l2_loss
    loss2[1] = (float)c;
  }
}

// ------------------------------------------------------- decode + filter + top-K
// Key: kept flag (bit 63) | order-preserving image of the confidence's float bits (32 bits) | prediction index
// (31 bits); bitonic sort, descending.  The float image (flip the sign bit of non-negative values, all bits of
// negative ones) orders ANY float like a comparison sort does -- negative scores and scores >= 2 included; a NaN
// sorts above everything, where numpy's argsort(...)[::-1] (detect.py:423) puts it.
__global__ void __launch_bounds__(kThreads)
decode_filter_topk_kernel(const float4* __restrict__ raw, const float* __restrict__ conf,
                          const float4* __restrict__ priors, const mbx_patch_meta* __restrict__ meta,
                          int P, int N /*pow2 >= P*/, int k_max, double* __restrict__ out_boxes,
                          float* __restrict__ out_scores, int* __restrict__ out_index,
                          int* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  __shared__ int kept_waves[kWaves];
  const int b = blockIdx.x, tid = threadIdx.x;
  const mbx_patch_meta m = meta[b];
  const float4* r = raw + (size_t)b * P;
  const float* c = conf + (size_t)b * P;
  int kept = 0;
  for (int j = tid; j < N; j += kThreads) {
    unsigned long long key = 0ull;
    if (j < P) {
      const float4 a = r[j], p = priors[j];
      const float x1 = fminf(fmaxf(__fadd_rn(a.x, p.x), 0.f), 1.f), y1 = fminf(fmaxf(__fadd_rn(a.y, p.y), 0.f), 1.f);
      const float x2 = fminf(fmaxf(__fadd_rn(a.z, p.z), 0.f), 1.f), y2 = fminf(fmaxf(__fadd_rn(a.w, p.w), 0.f), 1.f);
      // detect.py:92-99 (strict)
      const bool drop = (x1 < m.restrictions[0]) || (y1 < m.restrictions[1]) || (x2 > m.restrictions[2]) || (y2 > m.restrictions[3]);
      if (!drop) {
        const float cj = c[j];
        unsigned u = __float_as_uint(cj);
        u = (cj != cj) ? 0xffffffffu : (cj == 0.f) ? 0x80000000u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));   // -0 == +0
        key = (1ull << 63) | ((unsigned long long)u << 31) | (unsigned long long)j;
        ++kept;
      }
    }
    keys[j] = key;
  }
  kept = (int)wave_sum((float)kept);
  if ((tid & 63) == 0) kept_waves[tid >> 6] = kept;
  __syncthreads();
  int total_kept = 0;
  for (int w = 0; w < kWaves; ++w) total_kept += kept_waves[w];

  for (int size = 2; size <= N; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (N >> 1); t += kThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = keys[lo], bb = keys[hi];
        if ((a < bb) == desc) { keys[lo] = bb; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  int count = total_kept < m.max_to_keep ? total_kept : m.max_to_keep;   // detect.py:424
  if (count > k_max) count = k_max;
  if (count < 0) count = 0;
  if (tid == 0) out_count[b] = count;
  // detect.py:119-129
  const double xs = (double)m.patch_w / (double)m.image_w, ys = (double)m.patch_h / (double)m.image_h;
  const double xo = (double)m.offset_x / (double)m.image_w, yo = (double)m.offset_y / (double)m.image_h;
  for (int t = tid; t < k_max; t += kThreads) {
    double* ob = out_boxes + ((size_t)b * k_max + t) * 4;
    if (t < count) {
      const unsigned long long key = keys[t];
      const int j = (int)(key & 0x7fffffffull);
      const float4 a = r[j], p = priors[j];
      const float x1 = fminf(fmaxf(__fadd_rn(a.x, p.x), 0.f), 1.f), y1 = fminf(fmaxf(__fadd_rn(a.y, p.y), 0.f), 1.f);
      const float x2 = fminf(fmaxf(__fadd_rn(a.z, p.z), 0.f), 1.f), y2 = fminf(fmaxf(__fadd_rn(a.w, p.w), 0.f), 1.f);
      double X1 = __dadd_rn(__dmul_rn((double)x1, xs), xo), Y1 = __dadd_rn(__dmul_rn((double)y1, ys), yo);
      double X2 = __dadd_rn(__dmul_rn((double)x2, xs), xo), Y2 = __dadd_rn(__dmul_rn((double)y2, ys), yo);
      if (m.is_flipped) { const double t1 = 1.0 - X2, t2 = 1.0 - X1; X1 = t1; X2 = t2; }
      ob[0] = X1; ob[1] = Y1; ob[2] = X2; ob[3] = Y2;
      out_scores[(size_t)b * k_max + t] = c[j];
      out_index[(size_t)b * k_max + t] = j;
    } else {
      ob[0] = ob[1] = ob[2] = ob[3] = 0.0;
      out_scores[(size_t)b * k_max + t] = 0.f;
      out_index[(size_t)b * k_max + t] = -1;
    }
  }
}

}  // namespace

// =============================================================================== C ABI
extern "C" int mbx_decode_conf(const float* raw_locs, const float* logits, const float* priors, int B,
                               int P, float eps_add, float* decoded, float* conf, mbx_stream_t stream) {
  if (B < 0 || P <= 0) return MBX_ERR_INVALID_ARG;
  if ((decoded && (!raw_locs || !priors)) || (conf && !logits)) return MBX_ERR_INVALID_ARG;
  if (B == 0 || (!decoded && !conf)) return MBX_OK;
  const int total = B * P;
  int blocks = (total + kThreads - 1) / kThreads;
  if (blocks > 2048) blocks = 2048;
  MBX_ENTER();
  hipLaunchKernelGGL(decode_conf_kernel, dim3(blocks), dim3(kThreads), 0, mbx_s(stream),
                     reinterpret_cast<const float4*>(raw_locs), logits, reinterpret_cast<const float4*>(priors),
                     total, P, eps_add, reinterpret_cast<float4*>(decoded), conf);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

static size_t match_lds_bytes(int P, int G) { return (size_t)P * 36 + (size_t)G * 16 + 16; }

extern "C" size_t mbx_match_workspace_bytes(int, int, int) { return 0; }

extern "C" int mbx_match(const float* decoded, const float* conf, const float* gt, const int32_t* n_gt,
                         float alpha, int B, int P, int G, int32_t* match, int32_t* status, void*, size_t,
                         mbx_stream_t stream) {
  if (!decoded || !conf || !gt || !n_gt || !match || !status || B < 0 || P <= 0 || G <= 0) return MBX_ERR_INVALID_ARG;
  if (B == 0) return MBX_OK;
  const size_t lds = match_lds_bytes(P, G);
  if (lds > 150 * 1024) return MBX_ERR_UNSUPPORTED;        // P > ~4200 at G=100
  static const int force_nt = getenv("MBX_MATCH_THREADS") ? atoi(getenv("MBX_MATCH_THREADS")) : 0;      // (tools/match_bench.py)
  const int nthreads = force_nt ? force_nt : (P > 1536 ? 512 : kThreads);
  MBX_ENTER();
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return MBX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(match_kernel, dim3(B), dim3(nthreads), lds, mbx_s(stream),
                     reinterpret_cast<const float4*>(decoded), conf, reinterpret_cast<const float4*>(gt), n_gt,
                     alpha, P, G, match, status);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" size_t mbx_loss_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 1) * 2 * sizeof(double); }

extern "C" int mbx_loss_fwd_bwd(const float* decoded, const float* logits, int conf_is_logit, const float* gt,
                                const int32_t* match, float alpha, float grad_scale, int B, int P, int G, float* loss2,
                                float* d_raw_locs, float* d_logits, void* workspace, size_t workspace_bytes,
                                mbx_stream_t stream) {
  if (!decoded || !logits || !gt || !match || !loss2 || B <= 0 || P <= 0 || G <= 0) return MBX_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < mbx_loss_workspace_bytes(B)) return MBX_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(workspace);
  MBX_ENTER();
  hipLaunchKernelGGL(loss_kernel, dim3(B), dim3(kThreads), 0, mbx_s(stream),
                     reinterpret_cast<const float4*>(decoded), logits, conf_is_logit,
                     reinterpret_cast<const float4*>(gt), match, alpha, grad_scale, P, G, partial, reinterpret_cast<float4*>(d_raw_locs), d_logits);
  MBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, mbx_s(stream), partial, B, alpha, loss2);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

// ------------------------------------------------------------------------------------------ optional NMS (N1)
// The reference has NO non-maximum suppression (detect.py keeps the top max_to_keep boxes of every patch);
// BASELINE's north_star names one, so it exists as an OPTIONAL stage after mbx_decode_filter_topk, off by default.
// Greedy NMS per patch over the score-sorted list: box i is kept iff its IoU with every earlier KEPT box is <= thr.
// One workgroup per patch: the K x K "suppresses" relation as bit rows in LDS (pair (i, j), j > i, computed in
// float64 in the operation order of the numpy restatement: bit-exact decisions), then one wave walks the rows in
// order OR-ing the rows of kept boxes into the removed set (wavefront ballot-free: 64-bit words per lane), then the
// survivors are compacted in place.  K <= 1024.
constexpr int kNmsMaxK = 1024;

__global__ void __launch_bounds__(kThreads)
nms_kernel(double* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ index,
           int32_t* __restrict__ count, int k_max, double thr) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long nms_lds[];
  const int b = blockIdx.x;
  const int K = min(count[b], k_max);
  const int W = (K + 63) >> 6;                               // 64-bit words per row
  unsigned long long* rows = nms_lds;                        // [K][W]
  unsigned long long* removed = nms_lds + (size_t)K * W;     // [W]
  double* bx = boxes + (size_t)b * k_max * 4;
  for (int i = threadIdx.x; i < K * W; i += kThreads) rows[i] = 0ull;
  if (threadIdx.x < W) removed[threadIdx.x] = 0ull;
  __syncthreads();
  // pair (i, j), j > i: thread t takes row i = t / W', word w: 64 columns at a time
  for (int t = threadIdx.x; t < K * W; t += kThreads) {
    const int i = t / W, w = t - i * W;
    const double x1 = bx[i * 4], y1 = bx[i * 4 + 1], x2 = bx[i * 4 + 2], y2 = bx[i * 4 + 3];
    const double ai = (x2 - x1) * (y2 - y1);
    unsigned long long bits = 0ull;
    const int j0 = max(w * 64, i + 1), j1 = min(w * 64 + 64, K);
    for (int j = j0; j < j1; ++j) {
      const double u1 = bx[j * 4], v1 = bx[j * 4 + 1], u2 = bx[j * 4 + 2], v2 = bx[j * 4 + 3];
      const double iw = fmin(x2, u2) - fmax(x1, u1), ih = fmin(y2, v2) - fmax(y1, v1);
      const double inter = (iw > 0.0 && ih > 0.0) ? iw * ih : 0.0;
      const double uni = ai + (u2 - u1) * (v2 - v1) - inter;
      const double iou = uni > 0.0 ? inter / uni : 0.0;
      if (iou > thr) bits |= 1ull << (j & 63);
    }
    rows[t] = bits;
  }
  __syncthreads();
  if (threadIdx.x < 64) {                                    // one wave: lane w owns word w of the removed set (W <= 16)
    const int lane = threadIdx.x;
    unsigned long long rem = 0ull;
    for (int i = 0; i < K; ++i) {
      const unsigned long long word = __shfl(rem, i >> 6, 64);          // is box i still alive?
      if (!((word >> (i & 63)) & 1ull) && lane < W) rem |= rows[i * W + lane];
    }
    if (lane < W) removed[lane] = rem;
  }
  __syncthreads();
  // compaction: survivor i moves to position (number of survivors before i); one thread per box, prefix by popcount
  int dst = -1;
  double kb[4] = {0, 0, 0, 0};
  float ks = 0.f;
  int ki = 0;
  const int i = threadIdx.x;
  for (int base = 0; base < K; base += kThreads) {           // K <= 1024: at most four sweeps, each box read before any write
    const int ii = base + i;
    dst = -1;
    if (ii < K && !((removed[ii >> 6] >> (ii & 63)) & 1ull)) {
      int before = 0;
      for (int w = 0; w < (ii >> 6); ++w) before += __popcll(~removed[w]);
      before += __popcll(~removed[ii >> 6] & ((1ull << (ii & 63)) - 1ull));
      dst = before;
      kb[0] = bx[ii * 4]; kb[1] = bx[ii * 4 + 1]; kb[2] = bx[ii * 4 + 2]; kb[3] = bx[ii * 4 + 3];
      ks = scores[(size_t)b * k_max + ii];
      ki = index[(size_t)b * k_max + ii];
    }
    __syncthreads();                                          // dst <= ii and all sources of this sweep are in registers
    if (dst >= 0) {
      bx[dst * 4] = kb[0]; bx[dst * 4 + 1] = kb[1]; bx[dst * 4 + 2] = kb[2]; bx[dst * 4 + 3] = kb[3];
      scores[(size_t)b * k_max + dst] = ks;
      index[(size_t)b * k_max + dst] = ki;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int kept = 0;
    for (int w = 0; w < W; ++w) kept += __popcll(~removed[w] & (w == W - 1 && (K & 63) ? ((1ull << (K & 63)) - 1ull) : ~0ull));
    count[b] = K == 0 ? 0 : kept;
  }
}

extern "C" int mbx_nms(double* boxes, float* scores, int32_t* index, int32_t* count, int B, int k_max,
                       double iou_threshold, mbx_stream_t stream) {
  if (!boxes || !scores || !index || !count || B < 0 || k_max <= 0) return MBX_ERR_INVALID_ARG;
  if (k_max > kNmsMaxK) return MBX_ERR_UNSUPPORTED;
  if (B == 0) return MBX_OK;
  const int W = (k_max + 63) / 64;
  const size_t lds = ((size_t)k_max * W + W) * sizeof(unsigned long long);
  MBX_ENTER();
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return MBX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(kThreads), lds, mbx_s(stream), boxes, scores, index, count, k_max,
                     iou_threshold);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_decode_filter_topk(const float* raw_locs, const float* conf, const float* priors,
                                      const mbx_patch_meta* meta, int B, int P, int k_max, double* out_boxes,
                                      float* out_scores, int32_t* out_index, int32_t* out_count,
                                      mbx_stream_t stream) {
  if (!raw_locs || !conf || !priors || !meta || !out_boxes || !out_scores || !out_index || !out_count)
    return MBX_ERR_INVALID_ARG;
  if (B < 0 || P <= 0 || k_max <= 0) return MBX_ERR_INVALID_ARG;
  if (B == 0) return MBX_OK;
  int N = 64;
  while (N < P) N <<= 1;
  if (N > 16384) return MBX_ERR_UNSUPPORTED;
  const size_t lds = (size_t)N * sizeof(unsigned long long);
  MBX_ENTER();
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(decode_filter_topk_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MBX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(decode_filter_topk_kernel, dim3(B), dim3(kThreads), lds, mbx_s(stream),
                     reinterpret_cast<const float4*>(raw_locs), conf, reinterpret_cast<const float4*>(priors), meta,
                     P, N, k_max, out_boxes, out_scores, out_index, out_count);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}
