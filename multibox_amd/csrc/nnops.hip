// libmbx: batch norm (fwd/bwd), pooling, relu mask, head gather/scatter, input packing,
// filter preparation and the fused RMSProp+L2+EMA step.  All HBM-bound: 16-byte (8 x bf16)
// accesses per lane along the NHWC channel axis, grid-stride loops capped at 2048 blocks.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

// Workgroups are dealt to the eight XCDs round-robin (b and b + 8 share one): logical id = a contiguous run per XCD, the mapping
// of the convolution kernels' tiles (conv_common.h xcd_remap).  A row-wise kernel that walks its rows by LOGICAL id reads what
// the producing convolution's tiles left in THIS XCD's L2 and leaves its output where the consuming convolution's tiles will
// look for it: an XCD's L2 keeps what a kernel wrote across the kernel boundary (tools/l2_persist.hip: 32 MB read back by the
// XCD that wrote it 6.2 us, by another 8.2 us).  MEASURED LEVEL on the training step (tools/xcd_rows_ab.sh: 14.64-14.66 vs
// 14.61-14.65 ms; the 512 x 512 leg 37.70 vs 37.85): the convolutions' K loops are not paced by where their operands come
// from (LAB_NOTES round 2: loaders that do not wait for their data run the same 0.43 us per step).  Off; MBX_XCD_ROWS=1.
__device__ __forceinline__ int xcd_logical(int bid, int nblk) {
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
inline int xcd_rows_on() { static const int v = [] { const char* e = getenv("MBX_XCD_ROWS"); return e ? atoi(e) : 0; }(); return v; }


typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
constexpr int kT = 256;

__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned f2bf(float f) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f); }
// two floats to a packed bf16 pair in one v_cvt_pk_bf16_f32 (f2bf's rounding; the compiler emits one conversion per value
// plus a shift and an or), and max(x, 0) in one v_max_f32 (fmaxf canonicalises its operand first)
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{lo, hi}, pk_bf16x2));
}
__device__ __forceinline__ float relu_f(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
  f[0] = bf2f(v.x & 0xffffu); f[1] = bf2f(v.x >> 16); f[2] = bf2f(v.y & 0xffffu); f[3] = bf2f(v.y >> 16);
  f[4] = bf2f(v.z & 0xffffu); f[5] = bf2f(v.z >> 16); f[6] = bf2f(v.w & 0xffffu); f[7] = bf2f(v.w >> 16);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}
__device__ __forceinline__ u32x4 ld8(const unsigned short* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st8(unsigned short* p, const u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// Channel map of a batch-norm GROUP (several sibling convolutions normalised by one set of launches): channel c of the
// group's contiguous [M, C] tensors (y, dy) lives at channel c + off[i] of the strided activation / gradient view, where
// i is the last entry with cb[i] <= c (n = 0: the identity).  Lane-constant wherever a lane owns a channel group.
struct ChanMap { int n; int cb[4]; int off[4]; };
__device__ __forceinline__ int chan_off(const ChanMap& m, int c) {
  int o = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) if (i < m.n && c >= m.cb[i]) o = m.off[i];
  return o;
}
inline int chan_map_max(const ChanMap& m) { int o = 0; for (int i = 0; i < m.n; ++i) if (m.off[i] > o) o = m.off[i]; return o; }

inline int grid_for(long long work_items) {
  long long b = (work_items + kT - 1) / kT;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ batch norm: finalize
// one workgroup per channel: the 256 lanes stride over the partial rows (all loads in flight at once),
// wave reduction + LDS; latency of this kernel sits between every conv and its normalisation.
__device__ __forceinline__ void block_sum2(double& s1, double& s2) {
  __shared__ double red[kT / 64][2];
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s1; red[threadIdx.x >> 6][1] = s2; }
  __syncthreads();
  s1 = red[0][0] + red[1][0] + red[2][0] + red[3][0];
  s2 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
}

__global__ void __launch_bounds__(kT)
bn_finalize_kernel(const float* __restrict__ part, int rows, int C, double inv_count, float eps, float decay,
                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ mmean,
                   float* __restrict__ mvar) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += kT) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)r * C + c) * 2);
    s1 += v.x; s2 += v.y;
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    const double m = s1 * inv_count;
    double var = s2 * inv_count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (decay < 0.f) {                                                     // store mode: the update is applied later,
      if (mmean) mmean[c] = (float)m;                                      // gated by the step control word
      if (mvar) mvar[c] = (float)var;                                      // (mbx_bn_moving_update)
    } else {
      if (mmean) mmean[c] -= (1.0f - decay) * (mmean[c] - (float)m);      // assign_moving_average
      if (mvar) mvar[c] -= (1.0f - decay) * (mvar[c] - (float)var);
    }
  }
}

// the same for a GROUP: channel c belongs to part p (cstart[p] <= c < cstart[p + 1]), whose convolution wrote its own
// partial rows [rows[p]][C[p]][2]; mean / rstd / moving statistics are the group's contiguous arrays
struct PartTable { const float* part[4]; int rows[4], C[4], cstart[5]; int n; };
__global__ void __launch_bounds__(kT)
bn_finalize_parts_kernel(const PartTable t, double inv_count, float eps, float decay, float* __restrict__ mean,
                         float* __restrict__ rstd, float* __restrict__ mmean, float* __restrict__ mvar) {
  const int c = blockIdx.x;
  int pi = 0;
  while (pi + 1 < t.n && c >= t.cstart[pi + 1]) ++pi;
  const float* part = t.part[pi];
  const int rows = t.rows[pi], C = t.C[pi], cl = c - t.cstart[pi];
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += kT) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)r * C + cl) * 2);
    s1 += v.x; s2 += v.y;
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    const double m = s1 * inv_count;
    double var = s2 * inv_count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (decay < 0.f) {
      if (mmean) mmean[c] = (float)m;
      if (mvar) mvar[c] = (float)var;
    } else {
      if (mmean) mmean[c] -= (1.0f - decay) * (mmean[c] - (float)m);
      if (mvar) mvar[c] -= (1.0f - decay) * (mvar[c] - (float)var);
    }
  }
}

__global__ void __launch_bounds__(kT)
bn_fold_kernel(const float* mm, const float* mv, const float* beta, float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c < C) {
    const float s = 1.0f / sqrtf(mv[c] + eps);
    scale[c] = s;
    shift[c] = (beta ? beta[c] : 0.f) - mm[c] * s;
  }
}

// ------------------------------------------------------------------ batch norm: apply
// Every lane owns one 8-channel group for the whole launch (mean / rstd / beta stay in registers) and walks the
// rows: no per-element index division, no per-element parameter loads.
__global__ void __launch_bounds__(kT)
bn_apply_kernel(const unsigned short* __restrict__ y, long long M, int C, const float* __restrict__ mean,
                const float* __restrict__ rstd, const float* __restrict__ beta, int relu,
                unsigned short* __restrict__ a, int ld_a, const ChanMap map) {
  const int C8 = C >> 3;
  if (C8 <= kT) {
    const int rpi = kT / C8;                              // rows per sweep of the workgroup
    const int vc = threadIdx.x % C8, rr = threadIdx.x / C8;
    if (rr >= rpi) return;
    const int c = vc << 3;
    a += chan_off(map, c);                                // (lane-constant: the lane owns this channel group)
    float mu[8], rs[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; be[j] = beta[c + j]; }
    const long long step = (long long)gridDim.x * rpi;
    for (long long m = (long long)blockIdx.x * rpi + rr; m < M; m += step) {
      float f[8];
      unpack8(ld8(y + m * C + c), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = (f[j] - mu[j]) * rs[j] + be[j];
        f[j] = relu ? relu_f(v) : v;
      }
      st8(a + m * ld_a + c, pack8(f));
    }
    return;
  }
  const long long total = M * C8;                         // more than 2048 channels: generic walk
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
    const long long m = i / C8;
    const int c = (int)(i - m * C8) << 3;
    float f[8];
    unpack8(ld8(y + m * C + c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = (f[j] - mean[c + j]) * rstd[c + j] + beta[c + j];
      f[j] = relu ? relu_f(v) : v;
    }
    st8(a + m * ld_a + c + chan_off(map, c), pack8(f));
  }
}

// ---------------------------------------------------------------- batch norm: backward
struct BnBwdGeom { int C8, rows_per_iter, rpb, rows; };
inline int bn_bwd_rows_target() { const char* e = getenv("MBX_BN_BWD_ROWS"); const int v = e ? atoi(e) : 2048; return v >= 64 && v <= 8192 ? v : 2048; }
inline BnBwdGeom bn_bwd_geom(long long M, int C) {
  BnBwdGeom g;
  g.C8 = C / 8;
  g.rows_per_iter = kT / g.C8;
  if (g.rows_per_iter < 1) g.rows_per_iter = 1;
  static const int target = bn_bwd_rows_target();         // partial rows (= workgroups of the reduce launch) aimed at
  long long rpb = (M + target - 1) / target;
  if (rpb < 4LL * g.rows_per_iter) rpb = 4LL * g.rows_per_iter;
  rpb = ((rpb + g.rows_per_iter - 1) / g.rows_per_iter) * g.rows_per_iter;
  g.rpb = (int)rpb;
  g.rows = (int)((M + rpb - 1) / rpb);
  return g;
}

// MASK: 0 = no relu, 1 = relu mask from the stored activation (a > 0), 2 = relu mask recomputed from y with
// the forward expression ((y-mean)*rstd+beta > 0), so that `a` is not read at all (4 instead of 6 bytes/element).
template <int MASK>
__global__ void __launch_bounds__(kT)
bn_bwd_reduce_kernel(const unsigned short* __restrict__ da, int ld_da, const unsigned short* __restrict__ a, int ld_a,
                     const unsigned short* __restrict__ y, long long M, int C,
                     const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta,
                     int rpi, int rpb, float* __restrict__ partial, const ChanMap map) {
  extern __shared__ __attribute__((aligned(16))) float sred[];   // [rpi][C8][16]
  const int C8 = C >> 3;
  const int vc = threadIdx.x % C8, rr = threadIdx.x / C8;
  const bool active = rr < rpi;
  const int c = vc << 3;
  da += chan_off(map, c);                                 // (lane-constant)
  float s1[8], s2[8], mu[8], rs[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = 0.f; s2[j] = 0.f; mu[j] = mean[c + j]; rs[j] = rstd[c + j];
    be[j] = MASK == 2 ? beta[c + j] : 0.f;
  }
  const long long r0 = (long long)blockIdx.x * rpb;
  long long r1 = r0 + rpb;
  if (r1 > M) r1 = M;
  if (active)
    for (long long m = r0 + rr; m < r1; m += rpi) {
      float g[8], yy[8], aa[8];
      unpack8(ld8(da + m * ld_da + c), g);
      unpack8(ld8(y + m * C + c), yy);
      if (MASK == 1) unpack8(ld8(a + m * ld_a + c), aa);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (yy[j] - mu[j]) * rs[j];
        const bool pass = MASK == 0 || (MASK == 1 ? aa[j] > 0.f : xh + be[j] > 0.f);
        const float gj = pass ? g[j] : 0.f;
        s1[j] += gj;
        s2[j] += gj * xh;
      }
    }
  if (active) {
    float* o = sred + ((size_t)rr * C8 + vc) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = s1[j]; o[8 + j] = s2[j]; }
  }
  __syncthreads();
  if (threadIdx.x < C8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int r = 0; r < rpi; ++r) {
      const float* o = sred + ((size_t)r * C8 + threadIdx.x) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += o[j]; s2[j] += o[8 + j]; }
    }
    float* p = partial + ((size_t)blockIdx.x * C + (threadIdx.x << 3)) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[2 * j] = s1[j]; p[2 * j + 1] = s2[j]; }
  }
}

__global__ void __launch_bounds__(kT)
bn_bwd_finalize_kernel(const float* __restrict__ part, int rows, int C, double inv_M, float* __restrict__ dbeta,
                       float* __restrict__ m12) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += kT) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)r * C + c) * 2);
    s1 += v.x; s2 += v.y;
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] += (float)s1;
    m12[c] = (float)(s1 * inv_M);
    m12[C + c] = (float)(s2 * inv_M);
  }
}

template <int MASK>
__global__ void __launch_bounds__(kT)
bn_bwd_apply_kernel(const unsigned short* __restrict__ da, int ld_da, const unsigned short* __restrict__ a, int ld_a,
                    const unsigned short* __restrict__ y, long long M, int C,
                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta,
                    const float* __restrict__ m12, unsigned short* __restrict__ dy, const ChanMap map) {
  const int C8 = C >> 3;
  const long long total = M * C8;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
    const long long m = i / C8;
    const int c = (int)(i - m * C8) << 3;
    float g[8], yy[8], aa[8], o[8];
    unpack8(ld8(da + m * ld_da + c + chan_off(map, c)), g);
    unpack8(ld8(y + m * C + c), yy);
    if (MASK == 1) unpack8(ld8(a + m * ld_a + c), aa);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float rs = rstd[c + j];
      const float xh = (yy[j] - mean[c + j]) * rs;
      const bool pass = MASK == 0 || (MASK == 1 ? aa[j] > 0.f : xh + beta[c + j] > 0.f);
      const float gj = pass ? g[j] : 0.f;
      o[j] = rs * (gj - m12[c + j] - xh * m12[C + c + j]);
    }
    st8(dy + m * C + c, pack8(o));
  }
}

// ---------------------------------------------------------- batch norm: fused finalize + apply
// One launch instead of two: every workgroup owns a 64-channel group and a chunk of rows; it first
// re-reduces the conv-epilogue partials of ITS 64 channels (rows <= 1024: a few hundred L2-resident
// loads per lane), keeps mean / rstd in LDS, then normalises its rows.  The chunk-0 workgroup of each
// group also publishes mean / rstd (for the backward pass) and updates the moving statistics.
constexpr int kBnGroup = 64;

__global__ void __launch_bounds__(kT)
bn_apply_fused_kernel(const float* __restrict__ part, int rows, double inv_count, float eps, float decay,
                      const unsigned short* __restrict__ y, long long M, int C, const float* __restrict__ beta, int relu,
                      unsigned short* __restrict__ a, int ld_a, float* __restrict__ mean, float* __restrict__ rstd,
                      float* __restrict__ mmean, float* __restrict__ mvar, int rows_per_chunk, const ChanMap map,
                      float* __restrict__ thr) {
  __shared__ double red[8][kBnGroup][2];
  __shared__ float s_mean[kBnGroup], s_rstd[kBnGroup], s_beta[kBnGroup];
  const int c0 = blockIdx.x * kBnGroup;
  const int cw = min(kBnGroup, C - c0);
  const int V = cw >> 3;                                  // 16-byte vectors per row in this group
  const long long r_begin = (long long)blockIdx.y * rows_per_chunk;
  long long r_end = r_begin + rows_per_chunk;
  if (r_end > M) r_end = M;
  const int total = (int)(r_end - r_begin) * V;
  // The first kPre vectors of every lane are loaded BEFORE the statistics are reduced: the reads of y do not depend on
  // them, and with the partial rows just written by another kernel's atomics (memory side: an L2 miss) the two round
  // trips would otherwise sit back to back in front of the first store -- 9.7 us per launch against 6.5 for the plain
  // apply kernel, which ate what the removed finalize launch had saved.
  constexpr int kPre = 4;
  u32x4 pre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int i = threadIdx.x + u * kT;
    pre[u] = u32x4{0u, 0u, 0u, 0u};
    if (i < total) pre[u] = ld8(y + (r_begin + i / V) * C + c0 + ((i % V) << 3));
  }
  {
    // lane = (channel pair, one of 8 row lanes): 16-byte loads of {sum, sumsq} x 2 channels, SIXTEEN partial rows in
    // flight per lane (one L2 round trip per batch instead of one per row).  The order of the additions is fixed, so
    // every workgroup of a channel group computes bit-identical statistics.  (C % 8 == 0: a pair never straddles C.)
    const int cp = (threadIdx.x & 31) * 2, rl = threadIdx.x >> 5;
    double s1a = 0.0, s2a = 0.0, s1b = 0.0, s2b = 0.0;
    if (cp < cw) {
      const float* src = part + ((size_t)c0 + cp) * 2;
      int r = rl;
      for (; r + 8 * 15 < rows; r += 8 * 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(r + 8 * u) * C * 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) { s1a += v[u].x; s2a += v[u].y; s1b += v[u].z; s2b += v[u].w; }
      }
      if (r + 8 < rows) {                                  // (the 16-row tables of the atomic mode: both loads in flight)
        const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)r * C * 2);
        const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)(r + 8) * C * 2);
        s1a += v0.x; s2a += v0.y; s1b += v0.z; s2b += v0.w;
        s1a += v1.x; s2a += v1.y; s1b += v1.z; s2b += v1.w;
        r += 16;
      }
      for (; r < rows; r += 8) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * C * 2);
        s1a += v.x; s2a += v.y; s1b += v.z; s2b += v.w;
      }
    }
    red[rl][cp][0] = s1a; red[rl][cp][1] = s2a; red[rl][cp + 1][0] = s1b; red[rl][cp + 1][1] = s2b;
    __syncthreads();
    const int ch = threadIdx.x;
    if (ch < cw) {
      double s1 = red[0][ch][0], s2 = red[0][ch][1];
      for (int r = 1; r < 8; ++r) { s1 += red[r][ch][0]; s2 += red[r][ch][1]; }
      const double m = s1 * inv_count;
      double var = s2 * inv_count - m * m;
      if (var < 0.0) var = 0.0;
      const float fm = (float)m, fr = (float)(1.0 / sqrt(var + (double)eps));
      s_mean[ch] = fm; s_rstd[ch] = fr; s_beta[ch] = beta[c0 + ch];
      if (blockIdx.y == 0) {
        mean[c0 + ch] = fm; rstd[c0 + ch] = fr;
        // relu threshold on y for the backward pass's statistics epilogue: (y - mean) rstd + beta > 0  <=>  y > mean - beta / rstd
        if (thr) thr[c0 + ch] = relu ? fm - s_beta[ch] / fr : -__builtin_inff();
        if (decay < 0.f) {                                                 // store mode (bn_finalize_kernel)
          if (mmean) mmean[c0 + ch] = fm;
          if (mvar) mvar[c0 + ch] = (float)var;
        } else {
          if (mmean) mmean[c0 + ch] -= (1.0f - decay) * (mmean[c0 + ch] - fm);
          if (mvar) mvar[c0 + ch] -= (1.0f - decay) * (mvar[c0 + ch] - (float)var);
        }
      }
    }
    __syncthreads();
  }
  auto finish = [&](const int i, const u32x4 raw) {
    const long long m = r_begin + i / V;
    const int vc = (i % V) << 3;
    float f[8];
    unpack8(raw, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (f[j] - s_mean[vc + j]) * s_rstd[vc + j] + s_beta[vc + j];
      f[j] = relu ? relu_f(v) : v;
    }
    st8(a + m * ld_a + c0 + vc + chan_off(map, c0 + vc), pack8(f));
  };
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int i = threadIdx.x + u * kT;
    if (i < total) finish(i, pre[u]);
  }
  for (int i = threadIdx.x + kPre * kT; i < total; i += kT)
    finish(i, ld8(y + (r_begin + i / V) * C + c0 + ((i % V) << 3)));
}

// ------------------------------------------- batch norm: apply straight from FEW statistics rows (round 4)
// The consumer of a convolution that ADDED its tile sums into R <= 16 rows (mbx_conv_desc.stats_rows_mod).  bn_apply_kernel's
// walk -- a lane owns one 8-channel group, a workgroup sweeps whole rows: every wave access is a contiguous run of the
// tensor -- behind a prologue in which the workgroup reduces the R rows of ALL C channels itself (lane t: channels t,
// t + 256, ...; consecutive lanes read consecutive 8-byte {sum, sumsq} pairs) into LDS.  The 64-channel-group kernel above
// re-reduces only its own channels but reads 128-byte pieces of rows C * 2 bytes apart: 9.7-10.3 us per launch on the
// block17 / block35 layers against 6.5 for bn_apply_kernel, more than the finalize launch it replaces.  The first vector
// of y is loaded before the prologue (it does not depend on the statistics).  Workgroup 0 publishes mean / rstd / relu
// threshold and the moving statistics.  C <= 2048.
constexpr int kRowsMaxC = 2048;
__global__ void __launch_bounds__(kT)
bn_apply_rows_kernel(const float* __restrict__ part, int rows, double inv_count, float eps, float decay,
                     const unsigned short* __restrict__ y, long long M, int C, const float* __restrict__ beta, int relu,
                     unsigned short* __restrict__ a, int ld_a, float* __restrict__ mean, float* __restrict__ rstd,
                     float* __restrict__ mmean, float* __restrict__ mvar, const ChanMap map, float* __restrict__ thr, int xcd_rows) {
  extern __shared__ __attribute__((aligned(16))) float s_par[];          // [3][C]: mean, rstd, beta
  const int C8 = C >> 3;
  const int rpi = kT / C8 > 0 ? kT / C8 : 1;              // rows per sweep of the workgroup (C8 <= 256)
  const int vc = threadIdx.x % C8, rr = threadIdx.x / C8;
  const bool active = rr < rpi;
  const int c = vc << 3;
  // xcd_rows: every XCD owns an eighth of the rows (the eighth its convolution tiles cover) and ITS workgroups sweep it with
  // the XCD's stride -- eight compact windows moving through the tensor; else one grid-strided sweep (rows interleaved over
  // all XCDs at rpi-row granularity).  (A contiguous range per WORKGROUP measured 7 % slower on this kernel: 456 scattered
  // streams instead of a window.)
  long long step = (long long)gridDim.x * rpi, Mend = M;
  long long m = (long long)blockIdx.x * rpi + rr;
  if (xcd_rows && gridDim.x >= 8) {
    const int xcd = (int)blockIdx.x & 7, nx = ((int)gridDim.x - xcd + 7) >> 3;
    const long long perx = ((M + 7) / 8 + rpi - 1) / rpi * rpi;
    const long long lo = (long long)xcd * perx;
    step = (long long)nx * rpi; m = lo + (long long)((int)blockIdx.x >> 3) * rpi + rr; Mend = lo + perx < M ? lo + perx : M;
  }
  // The lane's first kRowsAhead rows of y are loaded BEFORE the prologue (they do not depend on the statistics): their way from
  // the Infinity Cache overlaps the rows' round trip, and with <= kRowsAhead rows per lane -- every layer of the 17 x 17 and
  // 8 x 8 stages at BATCH_SIZE 64 -- the sweep behind the prologue is arithmetic and stores (round 6; one row ahead before).
  constexpr int kRowsAhead = 4;
  u32x4 pre[kRowsAhead];
#pragma unroll
  for (int i = 0; i < kRowsAhead; ++i) {
    const long long mi = m + i * step;
    pre[i] = (active && mi < Mend) ? ld8(y + mi * C + c) : u32x4{0u, 0u, 0u, 0u};
  }
  // The rows were ADDED at the memory side (atomics): reading them back misses the L2 -- a round trip of ~1 us each.  With the
  // usual eight rows ALL the loads of a lane's channels (two channels per pass: C <= 512 in one pass) are issued back to back
  // and waited for once; as a rolled loop of four-row batches per channel a 320-channel layer took four round trips in a row,
  // half of the launch (round 6).  Any other row count: the batched loop.
  for (int chb = threadIdx.x; chb < C; chb += 2 * kT) {
   longlong2 w0[8], w1[8];
   const bool two = chb + kT < C;
   const float be0 = beta[chb], be1 = beta[two ? chb + kT : chb];          // (in flight with the rows)
   if (rows == 8) {
     const longlong2* a0 = reinterpret_cast<const longlong2*>(part) + chb;
     const longlong2* a1 = reinterpret_cast<const longlong2*>(part) + (two ? chb + kT : chb);
#pragma unroll
     for (int r = 0; r < 8; ++r) { w0[r] = a0[(size_t)r * C]; w1[r] = a1[(size_t)r * C]; }
   }
   for (int half = 0; half < (two ? 2 : 1); ++half) {
    const int ch = chb + half * kT;
    // the rows are 64-bit fixed-point sums (2^-20 units) added by integer atomics: exact integer sum, then one conversion
    const longlong2* src = reinterpret_cast<const longlong2*>(part) + ch;
    long long i1 = 0, i2 = 0;
    bool bad = false;
    int r = 0;
    if (rows == 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const longlong2 v = half ? w1[q] : w0[q];
        i1 += v.x; i2 += v.y;
        bad |= v.y < 0;
      }
      r = 8;
    }
    for (; r + 3 < rows; r += 4) {                        // four rows' loads in flight
      const longlong2 v0 = src[(size_t)(r + 0) * C], v1 = src[(size_t)(r + 1) * C], v2 = src[(size_t)(r + 2) * C], v3 = src[(size_t)(r + 3) * C];
      i1 += (v0.x + v1.x) + (v2.x + v3.x);
      i2 += (v0.y + v1.y) + (v2.y + v3.y);
      bad |= ((v0.y | v1.y) | (v2.y | v3.y)) < 0;
    }
    for (; r < rows; ++r) {
      const longlong2 v = src[(size_t)r * C];
      i1 += v.x; i2 += v.y;
      bad |= v.y < 0;
    }
    const double s1 = (double)i1 * (1.0 / 1048576.0), s2 = (double)i2 * (1.0 / 1048576.0);
    const double mu = s1 * inv_count;
    double var = s2 * inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    // a negative sum of squares = a tile sum was out of the fixed-point range or not finite (stats_add_fixed, conv_common.h):
    // the channel's statistics are NaN, as the float32 rows would have made them (inf - inf), never finite garbage
    // (the producers bound every channel's grand total by 2^62 in fixed point, so neither a row nor this sum can wrap, and a
    // poisoned row -- INT64_MIN plus less than 2^62 of legitimate adds -- keeps the total negative: stats_add_fixed)
    const bool poisoned = bad || i2 < 0;
    if (poisoned) var = (double)__builtin_nanf("");
    const float fm = poisoned ? __builtin_nanf("") : (float)mu, fr = poisoned ? __builtin_nanf("") : (float)(1.0 / sqrt(var + (double)eps)), be = half ? be1 : be0;
    s_par[ch] = fm; s_par[C + ch] = fr; s_par[2 * C + ch] = be;
    if (blockIdx.x == 0) {
      mean[ch] = fm; rstd[ch] = fr;
      if (thr) thr[ch] = relu ? fm - be / fr : -__builtin_inff();
      if (decay < 0.f) {                                                   // store mode (bn_finalize_kernel)
        if (mmean) mmean[ch] = fm;
        if (mvar) mvar[ch] = (float)var;
      } else {
        if (mmean) mmean[ch] -= (1.0f - decay) * (mmean[ch] - fm);
        if (mvar) mvar[ch] -= (1.0f - decay) * (mvar[ch] - (float)var);
      }
    }
   }
  }
  __syncthreads();
  if (!active) return;
  a += chan_off(map, c);
  float mu[8], rs[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mu[j] = s_par[c + j]; rs[j] = s_par[C + c + j]; be[j] = s_par[2 * C + c + j]; }
  auto finish = [&](const long long mi, const u32x4 v) {
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = (f[j] - mu[j]) * rs[j] + be[j];
      f[j] = relu ? relu_f(t) : t;
    }
    st8(a + mi * ld_a + c, pack8(f));
  };
  u32x4 cur = u32x4{0u, 0u, 0u, 0u};
  const long long mrest = m + kRowsAhead * step;
  if (mrest < Mend) cur = ld8(y + mrest * C + c);           // (the rest of a long sweep: in flight behind the first rows' arithmetic)
#pragma unroll
  for (int i = 0; i < kRowsAhead; ++i) {
    const long long mi = m + i * step;
    if (mi < Mend) finish(mi, pre[i]);
  }
  for (m = mrest; m < Mend; m += step) {
    u32x4 nxt = u32x4{0u, 0u, 0u, 0u};
    if (m + step < Mend) nxt = ld8(y + (m + step) * C + c);   // next row's load in flight behind this row's arithmetic
    finish(m, cur);
    cur = nxt;
  }
}

// ------------------------------------------- batch norm backward from FEW statistics rows (round 4)
// The sums {sum g, sum g y} were added by the data gradient(s) that wrote da (conv EV = 6) into `rows` rows: this launch is
// bn_apply_rows_kernel's shape -- prologue: the workgroup reduces the rows of all C channels into LDS (m1, m2 per channel),
// the first (da, y) pair of every lane in flight meanwhile; then whole-row sweeps, a lane owning one 8-channel group.
__global__ void __launch_bounds__(kT)
bn_bwd_rows_kernel(const float* __restrict__ part, int rows, double inv_M, const unsigned short* __restrict__ da, int ld_da,
                   const unsigned short* __restrict__ y, long long M, int C, const float* __restrict__ mean,
                   const float* __restrict__ rstd, const float* __restrict__ thr, float* __restrict__ dbeta,
                   unsigned short* __restrict__ dy, const ChanMap map) {
  extern __shared__ __attribute__((aligned(16))) float s_par[];          // [5][C]: mean, rstd, thr, m1, m2
  const int C8 = C >> 3;
  const int rpi = kT / C8 > 0 ? kT / C8 : 1;
  const int vc = threadIdx.x % C8, rr = threadIdx.x / C8;
  const bool active = rr < rpi;
  const int c = vc << 3;
  const long long step = (long long)gridDim.x * rpi;
  long long m = (long long)blockIdx.x * rpi + rr;
  const unsigned short* dap = da + c + chan_off(map, c);
  u32x4 g0 = u32x4{0u, 0u, 0u, 0u}, y0 = u32x4{0u, 0u, 0u, 0u};
  if (active && m < M) { g0 = ld8(dap + m * ld_da); y0 = ld8(y + m * C + c); }
  for (int ch = threadIdx.x; ch < C; ch += kT) {
    const float* src = part + (size_t)ch * 2;
    double s1 = 0.0, s2 = 0.0;
    int r = 0;
    for (; r + 3 < rows; r += 4) {
      const float2 v0 = *reinterpret_cast<const float2*>(src + (size_t)(r + 0) * C * 2);
      const float2 v1 = *reinterpret_cast<const float2*>(src + (size_t)(r + 1) * C * 2);
      const float2 v2 = *reinterpret_cast<const float2*>(src + (size_t)(r + 2) * C * 2);
      const float2 v3 = *reinterpret_cast<const float2*>(src + (size_t)(r + 3) * C * 2);
      s1 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      s2 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; r < rows; ++r) {
      const float2 v = *reinterpret_cast<const float2*>(src + (size_t)r * C * 2);
      s1 += v.x; s2 += v.y;
    }
    const float mu = mean[ch], rs = rstd[ch];
    // sum g xhat = rstd (sum g y - mean sum g), in double: the two terms nearly cancel when |mean| >> 1 / rstd
    const double sgx = (double)rs * (s2 - (double)mu * s1);
    s_par[ch] = mu; s_par[C + ch] = rs; s_par[2 * C + ch] = thr[ch];
    s_par[3 * C + ch] = (float)(s1 * inv_M); s_par[4 * C + ch] = (float)(sgx * inv_M);
    if (blockIdx.x == 0 && dbeta) dbeta[ch] += (float)s1;
  }
  __syncthreads();
  if (!active) return;
  float mu[8], rs[8], th[8], m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu[j] = s_par[c + j]; rs[j] = s_par[C + c + j]; th[j] = s_par[2 * C + c + j]; m1[j] = s_par[3 * C + c + j]; m2[j] = s_par[4 * C + c + j];
  }
  u32x4 gc = g0, yc = y0;
  for (; m < M; m += step) {
    u32x4 gn = u32x4{0u, 0u, 0u, 0u}, yn = u32x4{0u, 0u, 0u, 0u};
    if (m + step < M) { gn = ld8(dap + (m + step) * ld_da); yn = ld8(y + (m + step) * C + c); }
    float g[8], yy[8], o[8];
    unpack8(gc, g);
    unpack8(yc, yy);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (yy[j] - mu[j]) * rs[j];
      const float gj = yy[j] > th[j] ? g[j] : 0.f;
      o[j] = rs[j] * (gj - m1[j] - xh * m2[j]);
    }
    st8(dy + m * C + c, pack8(o));
    gc = gn; yc = yn;
  }
}

inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline void bn_fused_grid(long long M, int C, int rows, int& groups, int& chunks, int& rpc) {
  groups = (C + kBnGroup - 1) / kBnGroup;
  // every workgroup re-reduces rows x 64 partials: with many partial rows, fewer (longer) workgroups
  static const int blocks_few = env_int("MBX_BN_FUSED_BLOCKS", 1024), blocks_many = env_int("MBX_BN_FUSED_BLOCKS_MANY", 256);
  long long want = (rows > 16 ? blocks_many : blocks_few) / groups;
  if (want < 1) want = 1;
  long long maxc = (M + 255) / 256;
  if (want > maxc) want = maxc;
  rpc = (int)((M + want - 1) / want);
  chunks = (int)((M + rpc - 1) / rpc);
}

// ------------------------------------------------- batch norm backward: ONE pass over (da, y)
// Both passes of the batch-norm backward need every element of da and y, with the per-channel totals in
// between.  For all but the five stem layers the whole layer (M*C*4 bytes) fits in the register files of one
// resident 512-thread workgroup per CU, so one launch keeps its slice in VGPRs across a grid barrier:
//   load slice -> per-channel partial sums -> atomics into kObSlots accumulator copies -> grid barrier ->
//   totals -> dy from the registers.
// HBM traffic 4 B read + 2 B written per element instead of 8 + 2, and one launch instead of three.
// The grid is never larger than the CU count and one workgroup always fits a CU, so every workgroup is
// resident and the barrier cannot deadlock; the spin is bounded all the same (error flag in the workspace).
#ifndef MBX_OB_SLOTS
#define MBX_OB_SLOTS 8
#endif
#ifndef MBX_OB_SLEEP
#define MBX_OB_SLEEP 8
#endif
constexpr int kObT = 512;
constexpr int kObSlots = MBX_OB_SLOTS;                        // (tools/ab_builds.sh: -DMBX_OB_SLOTS=n / -DMBX_OB_SLEEP=n)
constexpr int kObMaxNV = 20;
#ifndef MBX_OB_SUB
#define MBX_OB_SUB 16
#endif
#ifndef MBX_OB_REL
#define MBX_OB_REL 32
#endif
constexpr int kObSub = MBX_OB_SUB, kObRel = MBX_OB_REL, kObLine = 32;   // barrier: counters / release words, words per line
constexpr int kObCtlWords = kObLine * (2 + kObSub + kObRel);     // line 0: {grid size, timeout flag}

struct ObGeom { int C8, rpi, G, rpb, nv; };
inline ObGeom ob_geom(long long M, int C, int ncu) {
  ObGeom g;
  g.C8 = C / 8;
  g.rpi = g.C8 <= kObT ? kObT / g.C8 : 0;
  if (g.rpi == 0) { g.G = g.rpb = 0; g.nv = 1 << 30; return g; }
  long long G = (M + g.rpi - 1) / g.rpi;
  if (G > ncu) G = ncu;
  g.rpb = (int)((M + G - 1) / G);
  g.G = (int)((M + g.rpb - 1) / g.rpb);                  // no empty workgroups
  g.nv = (g.rpb + g.rpi - 1) / g.rpi;
  return g;
}

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ob_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ob_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);      // past the end -> zeros
}
__device__ __forceinline__ void ob_store16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 0);         // past the end -> dropped
}

template <int NV, bool RELU>
__global__ void __launch_bounds__(kObT)
bn_bwd_onepass_kernel(const unsigned short* __restrict__ da, int ld_da, const unsigned short* __restrict__ y,
                      long long M, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                      const float* __restrict__ beta, float* __restrict__ dbeta, unsigned short* __restrict__ dy,
                      float* __restrict__ ws, int rpi, int rpb, float inv_M, unsigned spin_limit, int fault,
                      float* __restrict__ step_poison, const ChanMap map, int map_max) {
  extern __shared__ __attribute__((aligned(16))) float sred[];   // [rpi][C8][16] partial sums, then [2C] totals
  __shared__ int s_timeout;                                      // this workgroup gave up on the grid barrier
  const int C8 = C >> 3;
  const int vc = threadIdx.x % C8, rr = threadIdx.x / C8;
  const bool active = rr < rpi;
  const int c = vc << 3;
  // (fault bit 1: row slices dealt to the XCDs in contiguous runs, like the tiles of the data gradient that wrote da -- xcd_logical)
  const long long r0 = (long long)((fault & 2) ? xcd_logical((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * rpb;
  long long nr = (r0 + rpb > M ? M : r0 + rpb) - r0;                           // rows of this workgroup's slice
  const int nrows = nr > 0 ? (int)nr : 0;
  // slice-relative buffer descriptors: 32-bit offsets, rows past the slice read zeros / are not stored
  // (an empty slice gets zero-length descriptors: nothing is ever addressed through them)
  const __amdgpu_buffer_rsrc_t gr = ob_rsrc(da + r0 * ld_da, nrows ? (unsigned)(((nrows - 1) * ld_da + C + map_max) * 2) : 0u);
  const int cg = c + chan_off(map, c);                                         // this lane's channels in the gradient view
  const __amdgpu_buffer_rsrc_t yr = ob_rsrc(y + r0 * C, (unsigned)(nrows * C * 2));
  const __amdgpu_buffer_rsrc_t dr = ob_rsrc(dy + r0 * C, (unsigned)(nrows * C * 2));
  constexpr unsigned kPast = 0x80000000u;
  u32x4 vg[NV], vy[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int lr = rr + i * rpi;
    const bool ok = active && lr < nrows;
    vg[i] = ob_load16(gr, ok ? (unsigned)((lr * ld_da + cg) * 2) : kPast);  // zero gradient adds nothing to the sums
    vy[i] = ob_load16(yr, ok ? (unsigned)((lr * C + c) * 2) : kPast);
  }
  float s1[8], s2[8], mu[8], rs[8], be[8];
  {
    const int cs = active ? c : 0;                        // idle lanes read channel 0 (their gradient is zero)
    const float4 a0 = *reinterpret_cast<const float4*>(mean + cs), a1 = *reinterpret_cast<const float4*>(mean + cs + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(rstd + cs), b1 = *reinterpret_cast<const float4*>(rstd + cs + 4);
    mu[0] = a0.x; mu[1] = a0.y; mu[2] = a0.z; mu[3] = a0.w; mu[4] = a1.x; mu[5] = a1.y; mu[6] = a1.z; mu[7] = a1.w;
    rs[0] = b0.x; rs[1] = b0.y; rs[2] = b0.z; rs[3] = b0.w; rs[4] = b1.x; rs[5] = b1.y; rs[6] = b1.z; rs[7] = b1.w;
    if (RELU) {
      const float4 c0 = *reinterpret_cast<const float4*>(beta + cs), c1 = *reinterpret_cast<const float4*>(beta + cs + 4);
      be[0] = c0.x; be[1] = c0.y; be[2] = c0.z; be[3] = c0.w; be[4] = c1.x; be[5] = c1.y; be[6] = c1.z; be[7] = c1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) be[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  }
  // The empty volatile asms pin the order: one vector is expanded to floats at a time, and only the PACKED slice
  // stays live across the barrier (the second phase re-derives xhat / the mask from it).  Without them the
  // compiler expands the whole slice at once (16 floats per vector pair) and spills.
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    asm volatile("" : "+v"(vg[i]), "+v"(vy[i]));
    float g[8], yy[8];
    unpack8(vg[i], g);
    unpack8(vy[i], yy);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (yy[j] - mu[j]) * rs[j];
      const float gj = (!RELU || xh + be[j] > 0.f) ? g[j] : 0.f;
      s1[j] += gj;
      s2[j] += gj * xh;
    }
    asm volatile("" : "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s1[4]), "+v"(s1[5]), "+v"(s1[6]), "+v"(s1[7]));
    asm volatile("" : "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]), "+v"(s2[4]), "+v"(s2[5]), "+v"(s2[6]), "+v"(s2[7]));
  }
  if (active) {
    float* o = sred + ((size_t)rr * C8 + vc) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = s1[j]; o[8 + j] = s2[j]; }
  }
  __syncthreads();
  float* acc = ws + (size_t)(blockIdx.x & (kObSlots - 1)) * 2 * C;
  unsigned* ctl = reinterpret_cast<unsigned*>(ws + (size_t)kObSlots * 2 * C);
  if (threadIdx.x < C8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int r = 0; r < rpi; ++r) {
      const float* o = sred + ((size_t)r * C8 + threadIdx.x) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += o[j]; s2[j] += o[8 + j]; }
    }
    // stage the workgroup's sums as [2][C] behind the partials, so that the atomics below are lane-contiguous
    float* t = sred + (size_t)rpi * C8 * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { t[(threadIdx.x << 3) + j] = s1[j]; t[C + (threadIdx.x << 3) + j] = s2[j]; }
  }
  __syncthreads();
  // one wave instruction = 64 consecutive floats: the memory-side atomic unit serialises per instruction and
  // line (~24 ns each, measured: tools/atomic_bench.hip), so the layout decides the cost, not the lane count
  for (int e = threadIdx.x; e < 2 * C; e += kObT) atomicAdd(acc + e, sred[(size_t)rpi * C8 * 16 + e]);
  // ---- grid barrier.  Everything that is polled or counted sits on its own 128-byte line and is shared by few
  // workgroups (the same 24 ns per access apply): arrivals go to kObSub counters, the last arrival of each to
  // the top counter, the last of those sets kObRel release words; workgroup b polls release word b % kObRel.
  // No __threadfence() here: a release fence writes back the whole L2 of the XCD (measured: ~45 us per launch).
  // The only global writes before the barrier are device-scope atomics, which are performed at the memory
  // side; waiting for their acknowledgement orders them before the arrival.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned G = gridDim.x, b = blockIdx.x;
    const unsigned sub = b % kObSub, n_sub = (G - sub + kObSub - 1) / kObSub, n_top = G < kObSub ? G : kObSub;
    s_timeout = 0;
    // fault injection (tests only, MBX_DEBUG_BARRIER_FAULT=1): workgroup 0 never arrives, everyone else times out
    bool last = ((fault & 1) && b == 0) ? false : atomicAdd(ctl + kObLine * (1 + sub), 1u) == n_sub - 1;
    if (last) last = atomicAdd(ctl + kObLine * (1 + kObSub), 1u) == n_top - 1;
    if (last) {
      for (int r = 0; r < kObRel; ++r)
        __hip_atomic_store(ctl + kObLine * (2 + kObSub + r), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned* rel = ctl + kObLine * (2 + kObSub + b % kObRel);
      unsigned spins = 0;
      while (ld_agent(rel) == 0u) {
        __builtin_amdgcn_s_sleep(MBX_OB_SLEEP);
        // cannot happen when the grid is resident.  If it does (CUs taken by another stream's kernels), the totals
        // below are partial: raise the flag AND poison this workgroup's outputs with NaN, so that the step cannot
        // silently train on a wrong gradient (the host also checks the flag: Trainer.check_health)
        if (++spins > spin_limit) {
          ctl[1] = 1u; s_timeout = 1;
          if (step_poison) atomicAdd(step_poison, 1.0f);    // the optimiser skips a step whose control word is non-zero
          break;
        }
      }
    }
    if (b == 0) ctl[0] = G;
  }
  __syncthreads();
  // device-scope loads (served at the memory side like the atomics): no acquire fence, which would
  // invalidate the XCD's L2
  for (int e = threadIdx.x; e < 2 * C; e += kObT) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < kObSlots; ++sl) t += ld_agent(ws + (size_t)sl * 2 * C + e);
    sred[e] = t;
  }
  __syncthreads();
  float m1[8], m2[8];
  const float poison = s_timeout ? __builtin_nanf("") : 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t1 = (active ? sred[c + j] : 0.f) + poison;
    m1[j] = t1 * inv_M;
    m2[j] = (active ? sred[C + c + j] : 0.f) * inv_M;
    // d(beta) += sum g, by workgroup 0: a fire-and-forget atomic add (one adder per address: the same float as `+=`).  As
    // `dbeta[c + j] += t1` it compiled to eight load -> wait -> add -> store round trips IN A ROW on workgroup 0's first lanes --
    // and a launch ends when its last workgroup does: 4-6 us on the tail of every one of the step's 141 launches (round 6,
    // seen in the ISA: eight `s_waitcnt vmcnt(0)` behind the barrier).
    if (blockIdx.x == 0 && rr == 0 && dbeta) atomicAdd(dbeta + c + j, t1);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int lr = rr + i * rpi;
    const bool ok = active && lr < nrows;
    asm volatile("" : "+v"(vg[i]), "+v"(vy[i]));          // re-derive from the packed registers HERE, not earlier
    float g[8], yy[8], o[8];
    unpack8(vg[i], g);
    unpack8(vy[i], yy);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (yy[j] - mu[j]) * rs[j];
      const float gj = (!RELU || xh + be[j] > 0.f) ? g[j] : 0.f;
      o[j] = rs[j] * (gj - m1[j] - xh * m2[j]);
    }
    ob_store16(dr, ok ? (unsigned)((lr * C + c) * 2) : kPast, pack8(o));
    asm volatile("" ::: "memory");
  }
}

// ------------------------------------------------------------------------------- pooling
// KK / SS: window and stride fixed at compile time (3 / 2: every max pool of the network) so that the nine loads of
// a window are independent and in flight together; 0 = run-time values.
template <int KK, int SS>
__global__ void __launch_bounds__(kT)
maxpool_fwd_kernel(const unsigned short* __restrict__ x, long long xs, int ldx, int N, int H, int W, int C, int k_,
                   int stride_, unsigned short* __restrict__ y, long long ys, int ldy, int Ho, int Wo,
                   unsigned char* __restrict__ argmax) {
  const int k = KK ? KK : k_, stride = SS ? SS : stride_;
  const int C8 = C >> 3;
  const unsigned total = (unsigned)N * Ho * Wo * C8;            // < 2^31 (checked on the host): 32-bit index math
  for (unsigned i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    unsigned t = i / (unsigned)C8;
    const int c = (int)(i - t * C8) << 3;
    const unsigned t2 = t / (unsigned)Wo;
    const int ow = (int)(t - t2 * Wo);
    const int n = (int)(t2 / (unsigned)Ho);
    const int oh = (int)(t2 - (unsigned)n * Ho);
    float best[8];
    unsigned arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
    if constexpr (KK > 0) {
      u32x4 v[KK * KK];
#pragma unroll
      for (int r = 0; r < KK; ++r)
#pragma unroll
        for (int s = 0; s < KK; ++s) {
          const int h = oh * stride + r, w = ow * stride + s;
          v[r * KK + s] = (h < H && w < W) ? ld8(x + n * xs + ((long long)h * W + w) * ldx + c) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
      for (int r = 0; r < KK; ++r)
#pragma unroll
        for (int s = 0; s < KK; ++s) {
          const int h = oh * stride + r, w = ow * stride + s;
          if (h >= H || w >= W) continue;
          float f[8];
          unpack8(v[r * KK + s], f);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (f[j] > best[j]) { best[j] = f[j]; arg[j] = r * KK + s; }
        }
    } else {
      for (int r = 0; r < k; ++r)
        for (int s = 0; s < k; ++s) {
          const int h = oh * stride + r, w = ow * stride + s;
          if (h >= H || w >= W) continue;
          float f[8];
          unpack8(ld8(x + n * xs + ((long long)h * W + w) * ldx + c), f);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (f[j] > best[j]) { best[j] = f[j]; arg[j] = r * k + s; }
        }
    }
    st8(y + n * ys + ((long long)oh * Wo + ow) * ldy + c, pack8(best));
    if (argmax) {
      u32x2 av;
      av.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
      av.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
      *reinterpret_cast<u32x2*>(argmax + (((long long)n * Ho + oh) * Wo + ow) * C + c) = av;
    }
  }
}

// bn_apply_kernel + maxpool_fwd_kernel<3, 2> in ONE pass for a layer whose activation feeds ONLY a 3x3 / 2 VALID max-pool
// (the two stem pools): every window tap is normalised from y (relu((y - mean) rstd + beta), rounded to bf16 as bn_apply
// stores it) and reduced straight away -- first maximum in scan order, as maxpool_fwd_kernel -- so the activation (177 MB
// and 124 MB at BATCH_SIZE 64) is neither written nor read back: the pooled tensor and the argmax bytes are the only outputs.
// Bit-identical to the two launches it replaces.
__global__ void __launch_bounds__(kT)
bn_apply_maxpool3s2_kernel(const unsigned short* __restrict__ y, int N, int H, int W, int C, const float* __restrict__ mean,
                           const float* __restrict__ rstd, const float* __restrict__ beta, int relu,
                           unsigned short* __restrict__ p, long long ps, int ldp, int Ho, int Wo,
                           unsigned char* __restrict__ argmax) {
  const int C8 = C >> 3;
  const unsigned total = (unsigned)N * Ho * Wo * C8;
  for (unsigned i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    unsigned t = i / (unsigned)C8;
    const int c = (int)(i - t * C8) << 3;
    const unsigned t2 = t / (unsigned)Wo;
    const int ow = (int)(t - t2 * Wo);
    const int n = (int)(t2 / (unsigned)Ho);
    const int oh = (int)(t2 - (unsigned)n * Ho);
    u32x4 v[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q)                       // VALID: every tap lies inside the image
        v[r * 3 + q] = ld8(y + (((long long)n * H + (oh * 2 + r)) * W + (ow * 2 + q)) * C + c);
    float mu[8], rs[8], be[8], best[8];
    unsigned arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; be[j] = beta[c + j]; best[j] = -INFINITY; arg[j] = 0; }
#pragma unroll
    for (int tq = 0; tq < 9; ++tq) {
      float f[8];
      unpack8(v[tq], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = (f[j] - mu[j]) * rs[j] + be[j];
        a = relu ? relu_f(a) : a;
        a = bf2f(f2bf(a));                              // (the value bn_apply_kernel stores and maxpool_fwd_kernel reads)
        if (a > best[j]) { best[j] = a; arg[j] = tq; }
      }
    }
    st8(p + n * ps + ((long long)oh * Wo + ow) * ldp + c, pack8(best));
    if (argmax) {
      u32x2 av;
      av.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
      av.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
      *reinterpret_cast<u32x2*>(argmax + (((long long)n * Ho + oh) * Wo + ow) * C + c) = av;
    }
  }
}

// Gradient of a 3x3 / stride-2 VALID max-pool's INPUT for the 2 x 2 pixel block (2a + dh, 2b + dw), channels c .. c+7,
// gathered from the pool's output gradient and its argmax bytes -- for the batch-norm backward kernels that read it on the
// fly instead of from a stored tensor (bn_bwd_*_pool_kernel).  The block's four pixels are covered by the same four windows
// (oh in {a-1, a}, ow in {b-1, b}): 4 x 24 bytes of loads serve 4 pixels (the per-pixel gather of maxpool_bwd_kernel moves
// 96 bytes per pixel through L2 and was what a first, per-pixel version of these kernels spent their time on).  Same
// additions in the same order as maxpool_bwd_kernel<3, 2>, rounded to bf16 as it stores them: the same values.
struct PoolSrc { const unsigned short* gy; long long gys; int ld_gy; const unsigned char* argmax; int H, W, Ho, Wo, C, Hb, Wb; };
__device__ __forceinline__ void pool3s2_grad_block(const PoolSrc& p, int n, int a, int b, int c, float (&g)[4][8]) {
  u32x4 gv[4];
  u32x2 av[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {                          // windows (a-1,b-1), (a-1,b), (a,b-1), (a,b)
    const int oh = a - 1 + (i >> 1), ow = b - 1 + (i & 1);
    const bool v = oh >= 0 && oh < p.Ho && ow >= 0 && ow < p.Wo;
    ok[i] = v;
    const long long o = ((long long)n * p.Ho + (v ? oh : 0)) * p.Wo + (v ? ow : 0);
    av[i] = v ? *reinterpret_cast<const u32x2*>(p.argmax + o * p.C + c) : u32x2{0xffffffffu, 0xffffffffu};   // (no tap is 255)
    gv[i] = v ? ld8(p.gy + n * p.gys + ((long long)oh * p.Wo + ow) * p.ld_gy + c) : u32x4{0u, 0u, 0u, 0u};
  }
  float w[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack8(gv[i], w[i]);
  (void)ok;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = ((j < 4 ? av[i].x : av[i].y) >> (8 * (j & 3))) & 0xffu;
    // pixel (dh, dw) <- window i with tap (2a + dh - 2 oh) * 3 + (2b + dw - 2 ow), windows in ascending (oh, ow) order
    float p00 = 0.f, p01 = 0.f, p10 = 0.f, p11 = 0.f;
    if (t[0] == 8u) p00 += w[0][j];
    if (t[1] == 6u) p00 += w[1][j];
    if (t[2] == 2u) p00 += w[2][j];
    if (t[3] == 0u) p00 += w[3][j];
    if (t[1] == 7u) p01 += w[1][j];
    if (t[3] == 1u) p01 += w[3][j];
    if (t[2] == 5u) p10 += w[2][j];
    if (t[3] == 3u) p10 += w[3][j];
    if (t[3] == 4u) p11 += w[3][j];
    g[0][j] = bf2f(f2bf(p00)); g[1][j] = bf2f(f2bf(p01)); g[2][j] = bf2f(f2bf(p10)); g[3][j] = bf2f(f2bf(p11));
  }
}

// bn_bwd_reduce_kernel<2 / 0> and bn_bwd_apply_kernel<2 / 0> for a layer whose output feeds ONLY a 3x3 / 2 max-pool (the two
// stem pools, model.py:103,115): the activation gradient is gathered from the pool's output gradient on the fly, the
// max-pool backward launch and its 2 + 2 x 2 bytes per element of write / re-read disappear (177 MB and 124 MB tensors).
// A "row" of these kernels is a 2 x 2 pixel block (n, a, b); Mb = N Hb Wb of them, Hb = ceil(H / 2), Wb = ceil(W / 2).
template <bool RELU>
__global__ void __launch_bounds__(kT)
bn_bwd_reduce_pool_kernel(const PoolSrc ps, const unsigned short* __restrict__ y, long long Mb, int C,
                          const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta,
                          int rpi, int rpb, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sred[];   // [rpi][C8][16]
  const int C8 = C >> 3;
  const int vc = threadIdx.x % C8, rr = threadIdx.x / C8;
  const bool active = rr < rpi;
  const int c = vc << 3;
  float s1[8], s2[8], mu[8], rs[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = 0.f; s2[j] = 0.f; mu[j] = mean[c + j]; rs[j] = rstd[c + j];
    be[j] = RELU ? beta[c + j] : 0.f;
  }
  const long long r0 = (long long)blockIdx.x * rpb;
  long long r1 = r0 + rpb;
  if (r1 > Mb) r1 = Mb;
  const unsigned HWb = (unsigned)(ps.Hb * ps.Wb);
  if (active)
    for (long long m = r0 + rr; m < r1; m += rpi) {
      const unsigned mu_ = (unsigned)m, n = mu_ / HWb, rem = mu_ - n * HWb, a = rem / (unsigned)ps.Wb, b = rem - a * (unsigned)ps.Wb;
      // the block's four rows of y FIRST, branch-free (clamped addresses: a pixel past the map's edge re-reads a valid one and is
      // masked below) -- under a `continue` per pixel the four loads were four round trips in a row behind the gather's
      // (round 6: 67 us at 3.1 TB/s)
      u32x4 yv[4];
      bool in[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int h = 2 * (int)a + (q >> 1), w = 2 * (int)b + (q & 1);
        in[q] = h < ps.H && w < ps.W;
        yv[q] = ld8(y + (((long long)n * ps.H + (h < ps.H ? h : ps.H - 1)) * ps.W + (w < ps.W ? w : ps.W - 1)) * C + c);
      }
      float g[4][8];
      pool3s2_grad_block(ps, (int)n, (int)a, (int)b, c, g);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float yy[8];
        unpack8(yv[q], yy);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (yy[j] - mu[j]) * rs[j];
          const float gj = (in[q] && (!RELU || xh + be[j] > 0.f)) ? g[q][j] : 0.f;
          s1[j] += gj;
          s2[j] += gj * xh;
        }
      }
    }
  if (active) {
    float* o = sred + ((size_t)rr * C8 + vc) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = s1[j]; o[8 + j] = s2[j]; }
  }
  __syncthreads();
  if (threadIdx.x < C8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int r = 0; r < rpi; ++r) {
      const float* o = sred + ((size_t)r * C8 + threadIdx.x) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += o[j]; s2[j] += o[8 + j]; }
    }
    float* p = partial + ((size_t)blockIdx.x * C + (threadIdx.x << 3)) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[2 * j] = s1[j]; p[2 * j + 1] = s2[j]; }
  }
}

template <bool RELU>
__global__ void __launch_bounds__(kT)
bn_bwd_apply_pool_kernel(const PoolSrc ps, const unsigned short* __restrict__ y, long long Mb, int C, long long M,
                         const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta,
                         const float* __restrict__ m12, unsigned short* __restrict__ dy) {
  const int C8 = C >> 3;
  const long long total = Mb * C8;
  const unsigned HWb = (unsigned)(ps.Hb * ps.Wb);
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
    const long long m = i / C8;
    const int c = (int)(i - m * C8) << 3;
    const unsigned mu_ = (unsigned)m, n = mu_ / HWb, rem = mu_ - n * HWb, a = rem / (unsigned)ps.Wb, b = rem - a * (unsigned)ps.Wb;
    float g[4][8], mu[8], rs[8], be[8], q1[8], q2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mean[c + j]; rs[j] = rstd[c + j]; be[j] = RELU ? beta[c + j] : 0.f; q1[j] = m12[c + j]; q2[j] = m12[C + c + j];
    }
    u32x4 yv[4];                                         // (the four rows of y first, branch-free: bn_bwd_reduce_pool_kernel)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = 2 * (int)a + (q >> 1), w = 2 * (int)b + (q & 1);
      yv[q] = ld8(y + (((long long)n * ps.H + (h < ps.H ? h : ps.H - 1)) * ps.W + (w < ps.W ? w : ps.W - 1)) * C + c);
    }
    pool3s2_grad_block(ps, (int)n, (int)a, (int)b, c, g);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = 2 * (int)a + (q >> 1), w = 2 * (int)b + (q & 1);
      if (h >= ps.H || w >= ps.W) continue;
      const long long row = ((long long)n * ps.H + h) * ps.W + w;
      float yy[8], o[8];
      unpack8(yv[q], yy);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (yy[j] - mu[j]) * rs[j];
        const float gj = (!RELU || xh + be[j] > 0.f) ? g[q][j] : 0.f;
        o[j] = rs[j] * (gj - q1[j] - xh * q2[j]);
      }
      st8(dy + row * C + c, pack8(o));
    }
  }
}

template <int KK, int SS>
__global__ void __launch_bounds__(kT)
maxpool_bwd_kernel(const unsigned short* __restrict__ dy, long long dys, int ld_dy,
                   const unsigned char* __restrict__ argmax, int N, int H, int W, int C, int k_, int stride_, int Ho,
                   int Wo, unsigned short* __restrict__ dx, long long dxs, int ld_dx, int accumulate) {
  const int k = KK ? KK : k_, stride = SS ? SS : stride_;
  const int C8 = C >> 3;
  const unsigned total = (unsigned)N * H * W * C8;              // < 2^31 (checked on the host): 32-bit index math
  for (unsigned i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    unsigned t = i / (unsigned)C8;
    const int c = (int)(i - t * C8) << 3;
    const unsigned t2 = t / (unsigned)W;
    const int w = (int)(t - t2 * W);
    const int n = (int)(t2 / (unsigned)H);
    const int h = (int)(t2 - (unsigned)n * H);
    float acc[8];
    unsigned short* dst = dx + n * dxs + ((long long)h * W + w) * ld_dx + c;
    if (accumulate) unpack8(ld8(dst), acc);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
    if constexpr (KK > 0) {
      // the (at most NW x NW) windows covering (h, w), ascending like the run-time loops: all loads first
      constexpr int NW = (KK + SS - 1) / SS;
      const int ohb = h / SS, owb = w / SS;
      u32x4 gv[NW * NW];
      u32x2 av[NW * NW];
      bool ok[NW * NW];
#pragma unroll
      for (int a = 0; a < NW; ++a)
#pragma unroll
        for (int b = 0; b < NW; ++b) {
          const int oh = ohb - (NW - 1 - a), ow = owb - (NW - 1 - b);
          const bool v = oh >= 0 && oh < Ho && ow >= 0 && ow < Wo && h - oh * SS < KK && w - ow * SS < KK;
          ok[a * NW + b] = v;
          const long long o = ((long long)n * Ho + (v ? oh : 0)) * Wo + (v ? ow : 0);
          av[a * NW + b] = v ? *reinterpret_cast<const u32x2*>(argmax + o * C + c) : u32x2{0u, 0u};
          gv[a * NW + b] = v ? ld8(dy + n * dys + ((long long)oh * Wo + ow) * ld_dy + c) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
      for (int a = 0; a < NW; ++a)
#pragma unroll
        for (int b = 0; b < NW; ++b) {
          if (!ok[a * NW + b]) continue;
          const int oh = ohb - (NW - 1 - a), ow = owb - (NW - 1 - b);
          const unsigned tap = (h - oh * SS) * KK + (w - ow * SS);
          float g[8];
          unpack8(gv[a * NW + b], g);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned aj = ((j < 4 ? av[a * NW + b].x : av[a * NW + b].y) >> (8 * (j & 3))) & 0xffu;
            if (aj == tap) acc[j] += g[j];
          }
        }
    } else {
      int oh0 = (h - k + stride) / stride;          // ceil((h-k+1)/stride) for h-k+1 > -stride
      if (h - k + 1 <= 0) oh0 = 0;
      int ow0 = (w - k + stride) / stride;
      if (w - k + 1 <= 0) ow0 = 0;
      const int oh1 = min(h / stride, Ho - 1), ow1 = min(w / stride, Wo - 1);
      for (int oh = oh0; oh <= oh1; ++oh)
        for (int ow = ow0; ow <= ow1; ++ow) {
          const unsigned tap = (h - oh * stride) * k + (w - ow * stride);
          const long long o = ((long long)n * Ho + oh) * Wo + ow;
          const u32x2 av = *reinterpret_cast<const u32x2*>(argmax + o * C + c);
          float g[8];
          unpack8(ld8(dy + n * dys + ((long long)oh * Wo + ow) * ld_dy + c), g);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned aj = ((j < 4 ? av.x : av.y) >> (8 * (j & 3))) & 0xffu;
            if (aj == tap) acc[j] += g[j];
          }
        }
    }
    st8(dst, pack8(acc));
  }
}

__global__ void __launch_bounds__(kT)
avgpool_fwd_kernel(const unsigned short* __restrict__ x, long long xs, int ldx, int N, int H, int W, int C, int k,
                   int pad, unsigned short* __restrict__ y, long long ys, int ldy, int Ho, int Wo) {
  const int C8 = C >> 3;
  const unsigned total = (unsigned)N * Ho * Wo * C8;            // < 2^31 (checked on the host): 32-bit index math
  for (unsigned i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    unsigned t = i / (unsigned)C8;
    const int c = (int)(i - t * C8) << 3;
    const unsigned t2 = t / (unsigned)Wo;
    const int ow = (int)(t - t2 * Wo);
    const int n = (int)(t2 / (unsigned)Ho);
    const int oh = (int)(t2 - (unsigned)n * Ho);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int cnt = 0;
    if (k == 3) {
      // the 3x3 pool of Mixed_5b (model.py:134): ALL nine taps' loads in flight, added in tap order (as the loop below adds
      // them: same values).  Row by row the launch was three dependent round trips per output: 19 us for 15 MB.
      u32x4 v[9];
      bool ok[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int h = oh - pad + r, w = ow - pad + q;
          ok[r * 3 + q] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
          v[r * 3 + q] = ok[r * 3 + q] ? ld8(x + n * xs + ((long long)h * W + w) * ldx + c) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
      for (int tq = 0; tq < 9; ++tq) {
        if (!ok[tq]) continue;
        float f[8];
        unpack8(v[tq], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
        ++cnt;
      }
    } else
    // eight taps' loads in flight at a time, added in tap order (the 8x8 head pool read its 64 taps one dependent load after
    // the other: 24 us for 12 MB)
    for (int r = 0; r < k; ++r) {
      const int h = oh - pad + r;
      if ((unsigned)h >= (unsigned)H) continue;
      const unsigned short* row = x + n * xs + (long long)h * W * ldx + c;
      for (int s0 = 0; s0 < k; s0 += 8) {                 // (eight: a whole row of the 8x8 head pool, model.py:285)
        u32x4 v[8];
        bool ok[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int w = ow - pad + s0 + q;
          ok[q] = s0 + q < k && (unsigned)w < (unsigned)W;
          v[q] = ok[q] ? ld8(row + (long long)w * ldx) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (!ok[q]) continue;
          float f[8];
          unpack8(v[q], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += f[j];
          ++cnt;
        }
      }
    }
    const float inv = 1.0f / (float)cnt;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv;
    st8(y + n * ys + ((long long)oh * Wo + ow) * ldy + c, pack8(acc));
  }
}

__global__ void __launch_bounds__(kT)
avgpool_bwd_kernel(const unsigned short* __restrict__ dy, long long dys, int ld_dy, int N, int H, int W, int C, int k,
                   int pad, int Ho, int Wo, unsigned short* __restrict__ dx, long long dxs, int ld_dx,
                   int accumulate) {
  const int C8 = C >> 3;
  const unsigned total = (unsigned)N * H * W * C8;              // < 2^31 (checked on the host): 32-bit index math
  for (unsigned i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    unsigned t = i / (unsigned)C8;
    const int c = (int)(i - t * C8) << 3;
    const unsigned t2 = t / (unsigned)W;
    const int w = (int)(t - t2 * W);
    const int n = (int)(t2 / (unsigned)H);
    const int h = (int)(t2 - (unsigned)n * H);
    float acc[8];
    unsigned short* dst = dx + n * dxs + ((long long)h * W + w) * ld_dx + c;
    if (accumulate) unpack8(ld8(dst), acc);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
    const int oh0 = max(h + pad - k + 1, 0), oh1 = min(h + pad, Ho - 1);
    const int ow0 = max(w + pad - k + 1, 0), ow1 = min(w + pad, Wo - 1);
    if (k == 3) {
      // all (at most nine) windows' loads in flight, added in the order of the loops below (same values): as dependent
      // loads the launch took ten round trips per element (35 us for 45 MB)
      u32x4 v[9];
      float inv[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int oh = oh0 + a, ow = ow0 + b;
          const bool ok = oh <= oh1 && ow <= ow1;
          const int nh = min(oh - pad + 3, H) - max(oh - pad, 0), nw = min(ow - pad + 3, W) - max(ow - pad, 0);
          inv[a * 3 + b] = ok ? 1.0f / (float)(nh * nw) : 0.f;
          v[a * 3 + b] = ok ? ld8(dy + n * dys + ((long long)oh * Wo + ow) * ld_dy + c) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
      for (int tq = 0; tq < 9; ++tq) {
        if (inv[tq] == 0.f) continue;
        float g[8];
        unpack8(v[tq], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += g[j] * inv[tq];
      }
    } else
    for (int oh = oh0; oh <= oh1; ++oh) {
      const int nh = min(oh - pad + k, H) - max(oh - pad, 0);
      for (int ow = ow0; ow <= ow1; ++ow) {
        const int nw = min(ow - pad + k, W) - max(ow - pad, 0);
        const float inv = 1.0f / (float)(nh * nw);
        float g[8];
        unpack8(ld8(dy + n * dys + ((long long)oh * Wo + ow) * ld_dy + c), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += g[j] * inv;
      }
    }
    st8(dst, pack8(acc));
  }
}

// ------------------------------------------------------------------------- glue kernels
__global__ void __launch_bounds__(kT)
relu_mask_kernel(unsigned short* __restrict__ g, int ld_g, const unsigned short* __restrict__ a, int ld_a,
                 long long M, int C) {
  const int C8 = C >> 3;
  const long long total = M * C8;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
    const long long m = i / C8;
    const int c = (int)(i - m * C8) << 3;
    float gg[8], aa[8];
    unpack8(ld8(g + m * ld_g + c), gg);
    unpack8(ld8(a + m * ld_a + c), aa);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!(aa[j] > 0.f)) gg[j] = 0.f;
    st8(g + m * ld_g + c, pack8(gg));
  }
}

__global__ void __launch_bounds__(kT)
pack_input_kernel(const float* __restrict__ img, long long pixels, unsigned short* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < pixels; i += (long long)gridDim.x * kT) {
    float f[8] = {img[3 * i], img[3 * i + 1], img[3 * i + 2], 0.f, 0.f, 0.f, 0.f, 0.f};
    st8(out + 8 * i, pack8(f));
  }
}

__global__ void __launch_bounds__(kT)
head_gather_kernel(const float* __restrict__ h, int ld_h, int N, int cells, int k, int P, int off,
                   float* __restrict__ locs, float* __restrict__ logits) {
  const int total = N * cells * k;
  for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    const int a = i % k, t = i / k, cell = t % cells, n = t / cells;
    const float* row = h + (size_t)(n * cells + cell) * ld_h;
    const size_t p = (size_t)n * P + off + cell * k + a;
    *reinterpret_cast<float4*>(locs + p * 4) = make_float4(row[a * 4], row[a * 4 + 1], row[a * 4 + 2], row[a * 4 + 3]);
    logits[p] = row[4 * k + a];
  }
}

__global__ void __launch_bounds__(kT)
head_scatter_kernel(const float* __restrict__ d_locs, const float* __restrict__ d_logits, int N, int cells, int k,
                    int P, int off, unsigned short* __restrict__ g, int ld_g) {
  const int total = N * cells * ld_g;
  for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    const int ch = i % ld_g, t = i / ld_g, cell = t % cells, n = t / cells;
    const size_t p = (size_t)n * P + off + cell * k;
    float v = 0.f;
    if (ch < 4 * k) v = d_locs[(p + ch / 4) * 4 + (ch & 3)];
    else if (ch < 5 * k) v = d_logits[p + (ch - 4 * k)];
    g[i] = (unsigned short)f2bf(v);
  }
}

// every detection head of the network in ONE launch (a kernel costs >= 4.4 us in the replayed step whatever it does: six
// gathers behind six tiny head convolutions were 26 us of launches for 160 KB of data)
constexpr int kMaxHeads = 8;
struct HeadTable {
  const float* h[kMaxHeads]; unsigned short* g[kMaxHeads];
  int ld_h[kMaxHeads], ld_g[kMaxHeads], cells[kMaxHeads], k[kMaxHeads], off[kMaxHeads];
  int start[kMaxHeads + 1];                  // prefix sums of the per-head element counts
  int n;
};

__global__ void __launch_bounds__(kT)
head_gather_all_kernel(const HeadTable t, int P, float* __restrict__ locs, float* __restrict__ logits) {
  const int total = t.start[t.n];
  for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    int j = 0;
    while (j + 1 < t.n && i >= t.start[j + 1]) ++j;
    const int e = i - t.start[j], k = t.k[j], cells = t.cells[j];
    const int a = e % k, q = e / k, cell = q % cells, n = q / cells;
    const float* row = t.h[j] + (size_t)(n * cells + cell) * t.ld_h[j];
    const size_t p = (size_t)n * P + t.off[j] + cell * k + a;
    *reinterpret_cast<float4*>(locs + p * 4) = make_float4(row[a * 4], row[a * 4 + 1], row[a * 4 + 2], row[a * 4 + 3]);
    logits[p] = row[4 * k + a];
  }
}

__global__ void __launch_bounds__(kT)
head_scatter_all_kernel(const HeadTable t, int P, const float* __restrict__ d_locs, const float* __restrict__ d_logits) {
  const int total = t.start[t.n];
  for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
    int j = 0;
    while (j + 1 < t.n && i >= t.start[j + 1]) ++j;
    const int e = i - t.start[j], k = t.k[j], cells = t.cells[j], ld = t.ld_g[j];
    const int ch = e % ld, q = e / ld, cell = q % cells, n = q / cells;
    const size_t p = (size_t)n * P + t.off[j] + cell * k;
    float v = 0.f;
    if (ch < 4 * k) v = d_locs[(p + ch / 4) * 4 + (ch & 3)];
    else if (ch < 5 * k) v = d_logits[p + (ch - 4 * k)];
    t.g[j][e] = (unsigned short)f2bf(v);
  }
}

// start of a step's backward pass in ONE launch: the gradient buffer and the BN-backward workspace cleared (two fills),
// and the grid-barrier time-outs of the PREVIOUS step -- word `ctl` of the gradient buffer, which this launch clears --
// added to a running total first (was: five small torch kernels over the per-layer flags)
__global__ void __launch_bounds__(kT)
step_begin_kernel(float* __restrict__ G, long long nG, float* __restrict__ ws, long long nws, long long ctl,
                  unsigned long long* __restrict__ timeouts_total, float* __restrict__ scalar) {
  if (scalar && blockIdx.x == 0 && threadIdx.x == 0) *scalar = 0.f;      // (the regularisation-loss accumulator of the optimiser)
  const long long n4 = nG >> 2, w4 = nws >> 2;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n4 + w4; i += (long long)gridDim.x * kT) {
    if (i < n4) {
      if (timeouts_total && i == (ctl >> 2)) {         // this lane is the one that clears the control word: read it first
        const float t = G[ctl];
        if (t > 0.f) *timeouts_total += (unsigned long long)t;
      }
      reinterpret_cast<float4*>(G)[i] = z;
    } else {
      reinterpret_cast<float4*>(ws)[i - n4] = z;
    }
  }
}

// ---------------------------------------------------------------------- filter prepare
// dst[c][r'][s'][kq] = src[kq][R-1-r'][S-1-s'][c]: a [K] x [C] transpose per tap.  Each workgroup moves a
// 64(k) x 32(c) patch of one tap through LDS so that both the reads (contiguous in c) and the writes
// (contiguous in k) are coalesced (the first version gathered 2-byte elements: 3.5 GB of fetches for 120 MB).
__global__ void __launch_bounds__(kT)
filter_prepare_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wd,
                      const mbx_filter_entry* __restrict__ table, int n_entries) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][40];     // 80-byte rows: 16-byte aligned chunks
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const mbx_filter_entry e = table[lo];
  const int kt = (e.Kpad + 63) / 64, ct = (e.C + 31) / 32;
  int b = blockIdx.x - e.first_block;
  const int ck = b % kt; b /= kt;
  const int cc = b % ct; b /= ct;
  const int s = b % e.S, r = b / e.S;                       // destination tap (r', s')
  const unsigned short* src = w + e.src_off;
  unsigned short* dst = wd + e.dst_off;
  const int k0 = ck * 64, c0 = cc * 32;
  const long long tap_src = ((long long)(e.R - 1 - r) * e.S + (e.S - 1 - s)) * e.C;
  if (((e.C | e.Kpad) & 7) == 0 && ((e.src_off | e.dst_off) & 7) == 0) {
    // 16 bytes per lane both ways: one load of 8 channels of one filter, one store of 8 filters of one channel
    static_assert(kT == 256, "one 16-byte chunk per thread");
    {
      const int kk = threadIdx.x >> 2, c = (threadIdx.x & 3) << 3;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (k0 + kk < e.K && c0 + c < e.C)
        v = *reinterpret_cast<const uint4*>(src + (long long)(k0 + kk) * e.R * e.S * e.C + tap_src + c0 + c);
      *reinterpret_cast<uint4*>(&tile[kk][c]) = v;
    }
    __syncthreads();
    {
      const int c = threadIdx.x >> 3, kk = (threadIdx.x & 7) << 3;
      if (k0 + kk < e.Kpad && c0 + c < e.C) {
        unsigned q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = (unsigned)tile[kk + 2 * j][c] | ((unsigned)tile[kk + 2 * j + 1][c] << 16);
        *reinterpret_cast<uint4*>(dst + (((long long)(c0 + c) * e.R + r) * e.S + s) * e.Kpad + k0 + kk) =
            make_uint4(q[0], q[1], q[2], q[3]);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < 64 * 32; i += kT) {         // read: c fastest
    const int kk = i >> 5, c = i & 31;
    unsigned short v = 0;
    if (k0 + kk < e.K && c0 + c < e.C) v = src[(long long)(k0 + kk) * e.R * e.S * e.C + tap_src + c0 + c];
    tile[kk][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 32; i += kT) {         // write: k fastest
    const int c = i >> 6, kk = i & 63;
    if (k0 + kk < e.Kpad && c0 + c < e.C)
      dst[(((long long)(c0 + c) * e.R + r) * e.S + s) * e.Kpad + k0 + kk] = tile[kk][c];
  }
}

// --------------------------------------------------------------------------- optimizer
__global__ void __launch_bounds__(kT)
rmsprop_ema_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ ms, float* __restrict__ mom,
                   float* __restrict__ ema, unsigned short* __restrict__ wb, long long n, float lr, float decay,
                   float momentum, float eps, float wd, float ema_decay, int trainable, float* __restrict__ reg_loss,
                   const float* __restrict__ skip_ctl) {
  // step control block (two floats, summed over ranks with the beta gradients): [0] grid-barrier timeouts of this
  // step's one-launch BN backward (its gradients are poisoned), [1] ranks that asked to stop (input exhausted).
  // Either non-zero: the step is NOT applied -- no update, no EMA, nothing written (uniform branch).
  if (skip_ctl && (skip_ctl[0] != 0.f || skip_ctl[1] != 0.f)) return;
  float sq = 0.f;
  auto one = [&](float wi, const float gi_, float& msi, float& momi, float& emai) -> float {     // one element; returns the new weight
    sq += wi * wi;
    if (ema) emai = emai - (1.0f - ema_decay) * (emai - wi);
    if (trainable) {
      const float gi = gi_ + wd * wi;
      const float m = decay * msi + (1.0f - decay) * gi * gi;
      msi = m;
      float step = lr * gi / sqrtf(m + eps);
      if (mom) { step = momentum * momi + step; momi = step; }
      wi -= step;
    }
    return wi;
  };
  // 16-byte accesses where every stream is 16-byte aligned (the flat parameter buffers are; round 6: the 4-byte form streamed
  // 1.8 GB per step at 5.5 TB/s), element by element otherwise
  const bool vec = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(ms) |
                     reinterpret_cast<uintptr_t>(mom) | reinterpret_cast<uintptr_t>(ema)) & 15) == 0 && (reinterpret_cast<uintptr_t>(wb) & 7) == 0;
  const long long n4 = vec ? n / 4 : 0;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n4; i += (long long)gridDim.x * kT) {
    float4 wv = reinterpret_cast<const float4*>(w)[i];
    float4 ev = ema ? reinterpret_cast<const float4*>(ema)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), mv = gv, ov = gv;
    if (trainable) { gv = reinterpret_cast<const float4*>(g)[i]; mv = reinterpret_cast<const float4*>(ms)[i]; if (mom) ov = reinterpret_cast<const float4*>(mom)[i]; }
    wv.x = one(wv.x, gv.x, mv.x, ov.x, ev.x); wv.y = one(wv.y, gv.y, mv.y, ov.y, ev.y);
    wv.z = one(wv.z, gv.z, mv.z, ov.z, ev.z); wv.w = one(wv.w, gv.w, mv.w, ov.w, ev.w);
    if (ema) reinterpret_cast<float4*>(ema)[i] = ev;
    if (trainable) { reinterpret_cast<float4*>(ms)[i] = mv; if (mom) reinterpret_cast<float4*>(mom)[i] = ov; reinterpret_cast<float4*>(w)[i] = wv; }
    if (wb) reinterpret_cast<uint2*>(wb)[i] = make_uint2(pack2bf(wv.x, wv.y), pack2bf(wv.z, wv.w));
  }
  for (long long i = 4 * n4 + (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
    float msi = trainable ? ms[i] : 0.f, momi = (trainable && mom) ? mom[i] : 0.f, emai = ema ? ema[i] : 0.f;
    const float wi = one(w[i], trainable ? g[i] : 0.f, msi, momi, emai);
    if (ema) ema[i] = emai;
    if (trainable) { ms[i] = msi; if (mom) mom[i] = momi; w[i] = wi; }
    if (wb) wb[i] = (unsigned short)f2bf(wi);
  }
  if (reg_loss && wd != 0.f) {
    sq = wave_sum(sq);
    __shared__ float red[kT / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < kT / 64; ++i) t += red[i];
      atomicAdd(reg_loss, 0.5f * wd * t);
    }
  }
}

__global__ void __launch_bounds__(kT)
ema_update_kernel(float* __restrict__ ema, const float* __restrict__ v, long long n, float d,
                  const float* __restrict__ skip_ctl) {
  if (skip_ctl && (skip_ctl[0] != 0.f || skip_ctl[1] != 0.f)) return;       // as rmsprop_ema_kernel
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
    const float e = ema[i];
    ema[i] = e - (1.0f - d) * (e - v[i]);
  }
}

// moving statistics of every batch-norm layer of a step in one launch, gated like the optimiser (a skipped step leaves
// them untouched); lane 0 counts the skipped steps
__global__ void __launch_bounds__(kT)
bn_moving_update_kernel(float* __restrict__ mm, float* __restrict__ mv, const float* __restrict__ bmean,
                        const float* __restrict__ bvar, long long n, float decay, const float* __restrict__ skip_ctl,
                        unsigned long long* __restrict__ skipped, float* __restrict__ ema_m, float* __restrict__ ema_v,
                        float ema_d) {
  if (skip_ctl && (skip_ctl[0] != 0.f || skip_ctl[1] != 0.f)) {
    if (skipped && blockIdx.x == 0 && threadIdx.x == 0) *skipped += 1ull;  // (one writer: launches are stream-ordered)
    return;
  }
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
    const float m = mm[i] - (1.0f - decay) * (mm[i] - bmean[i]);           // bn_finalize_kernel's expressions
    const float v = mv[i] - (1.0f - decay) * (mv[i] - bvar[i]);
    mm[i] = m;
    mv[i] = v;
    if (ema_m) { const float e = ema_m[i]; ema_m[i] = e - (1.0f - ema_d) * (e - m); }   // ema_update_kernel's, on the new values
    if (ema_v) { const float e = ema_v[i]; ema_v[i] = e - (1.0f - ema_d) * (e - v); }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
typedef const unsigned short* cus;
typedef unsigned short* us;

}  // namespace

// ================================================================================== C ABI
extern "C" int mbx_bn_finalize(const float* part, int rows, int C, int64_t count, float eps, float decay, float* mean,
                               float* rstd, float* mmean, float* mvar, mbx_stream_t stream) {
  if (!part || !mean || !rstd || rows <= 0 || C <= 0 || count <= 0) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(kT), 0, mbx_s(stream), part, rows, C,
                     1.0 / (double)count, eps, decay, mean, rstd, mmean, mvar);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_fold(const float* mm, const float* mv, const float* beta, float eps, int C, float* scale,
                           float* shift, mbx_stream_t stream) {
  if (!mm || !mv || !scale || !shift || C <= 0) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + kT - 1) / kT), dim3(kT), 0, mbx_s(stream), mm, mv, beta, eps, C, scale, shift);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

// mbx_chan_map (HOST) -> the kernels' by-value ChanMap; NULL = identity.  Entries: ascending c_begin starting at 0,
// multiples of 8, offsets >= 0 and multiples of 8 (16-byte accesses).
static int to_chan_map(const mbx_chan_map* m, int C, ChanMap& out) {
  memset(&out, 0, sizeof(out));
  if (!m) return MBX_OK;
  if (m->n < 0 || m->n > 4) return MBX_ERR_INVALID_ARG;
  for (int i = 0; i < m->n; ++i) {
    if (m->c_begin[i] % 8 || m->offset[i] % 8 || m->offset[i] < 0 || m->c_begin[i] >= C ||
        (i == 0 ? m->c_begin[0] != 0 : m->c_begin[i] <= m->c_begin[i - 1]))
      return MBX_ERR_INVALID_ARG;
    out.cb[i] = m->c_begin[i]; out.off[i] = m->offset[i];
  }
  out.n = m->n;
  return MBX_OK;
}

extern "C" int mbx_bn_apply_mapped(const void* y, int64_t M, int C, const float* mean, const float* rstd, const float* beta,
                                   int relu, void* a, int ld_a, const mbx_chan_map* map, mbx_stream_t stream) {
  if (!y || !a || !mean || !rstd || !beta || M <= 0 || C <= 0 || C % 8 || ld_a % 8 || !al16(y) || !al16(a))
    return MBX_ERR_INVALID_ARG;
  ChanMap cm;
  if (to_chan_map(map, C, cm) != MBX_OK) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * (C / 8))), dim3(kT), 0, mbx_s(stream), (cus)y, (long long)M, C,
                     mean, rstd, beta, relu, (us)a, ld_a, cm);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_apply(const void* y, int64_t M, int C, const float* mean, const float* rstd, const float* beta,
                            int relu, void* a, int ld_a, mbx_stream_t stream) {
  return mbx_bn_apply_mapped(y, M, C, mean, rstd, beta, relu, a, ld_a, nullptr, stream);
}

extern "C" int mbx_bn_finalize_parts(const float* const* parts, const int32_t* rows, const int32_t* Cs, int n_parts,
                                     int64_t count, float eps, float decay, float* mean, float* rstd, float* mmean,
                                     float* mvar, mbx_stream_t stream) {
  if (!parts || !rows || !Cs || n_parts <= 0 || n_parts > 4 || !mean || !rstd || count <= 0) return MBX_ERR_INVALID_ARG;
  PartTable t;
  memset(&t, 0, sizeof(t));
  int c = 0;
  for (int i = 0; i < n_parts; ++i) {
    if (!parts[i] || rows[i] <= 0 || Cs[i] <= 0) return MBX_ERR_INVALID_ARG;
    t.part[i] = parts[i]; t.rows[i] = rows[i]; t.C[i] = Cs[i]; t.cstart[i] = c;
    c += Cs[i];
  }
  for (int i = n_parts; i <= 4; ++i) t.cstart[i] = c;
  t.n = n_parts;
  MBX_ENTER();
  hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3(c), dim3(kT), 0, mbx_s(stream), t, 1.0 / (double)count, eps, decay, mean,
                     rstd, mmean, mvar);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_bwd_rows(int64_t M, int C) {
  if (M <= 0 || C <= 0 || C % 8 || C > 2048) return MBX_ERR_INVALID_ARG;
  return bn_bwd_geom(M, C).rows;
}

// relu with a == NULL: the mask is recomputed from y (needs beta) instead of read from the activation.
static int bn_bwd_args_ok(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y, int64_t M, int C,
                          const float* mean, const float* rstd, const float* beta) {
  if (!da || !y || !mean || !rstd || (relu && !a && !beta) || M <= 0 || C <= 0 || C % 8 || ld_da % 8 || (a && ld_a % 8))
    return 0;
  return al16(da) && al16(y) && (!a || al16(a));
}

extern "C" int mbx_bn_bwd_reduce(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y, int64_t M,
                                 int C, const float* mean, const float* rstd, const float* beta, float* partial,
                                 mbx_stream_t stream) {
  return mbx_bn_bwd_reduce_mapped(da, ld_da, a, ld_a, relu, y, M, C, mean, rstd, beta, partial, nullptr, stream);
}

extern "C" int mbx_bn_bwd_reduce_mapped(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y, int64_t M,
                                        int C, const float* mean, const float* rstd, const float* beta, float* partial,
                                        const mbx_chan_map* da_map, mbx_stream_t stream) {
  if (!partial || !bn_bwd_args_ok(da, ld_da, a, ld_a, relu, y, M, C, mean, rstd, beta)) return MBX_ERR_INVALID_ARG;
  if (C > 2048) return MBX_ERR_UNSUPPORTED;
  ChanMap cm;
  if (to_chan_map(da_map, C, cm) != MBX_OK || (da_map && da_map->n && a)) return MBX_ERR_INVALID_ARG;
  const BnBwdGeom g = bn_bwd_geom(M, C);
  MBX_ENTER();
  const size_t lds = (size_t)g.rows_per_iter * g.C8 * 16 * sizeof(float);
#define MBX_BN_RED(MASK)                                                                                              \
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<MASK>, dim3(g.rows), dim3(kT), lds, mbx_s(stream), (cus)da, ld_da, (cus)a, \
                     ld_a, (cus)y, (long long)M, C, mean, rstd, beta, g.rows_per_iter, g.rpb, partial, cm)
  if (!relu) MBX_BN_RED(0); else if (a) MBX_BN_RED(1); else MBX_BN_RED(2);
#undef MBX_BN_RED
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_bwd_finalize(const float* partial, int rows, int C, int64_t M, float* dbeta, float* m12,
                                   mbx_stream_t stream) {
  if (!partial || !m12 || rows <= 0 || C <= 0 || M <= 0) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(kT), 0, mbx_s(stream), partial, rows, C,
                     1.0 / (double)M, dbeta, m12);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_bwd_apply(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y, int64_t M,
                                int C, const float* mean, const float* rstd, const float* beta, const float* m12,
                                void* dy, mbx_stream_t stream) {
  return mbx_bn_bwd_apply_mapped(da, ld_da, a, ld_a, relu, y, M, C, mean, rstd, beta, m12, dy, nullptr, stream);
}

extern "C" int mbx_bn_bwd_apply_mapped(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y, int64_t M,
                                       int C, const float* mean, const float* rstd, const float* beta, const float* m12,
                                       void* dy, const mbx_chan_map* da_map, mbx_stream_t stream) {
  if (!m12 || !dy || !al16(dy) || !bn_bwd_args_ok(da, ld_da, a, ld_a, relu, y, M, C, mean, rstd, beta))
    return MBX_ERR_INVALID_ARG;
  ChanMap cm;
  if (to_chan_map(da_map, C, cm) != MBX_OK || (da_map && da_map->n && a)) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
#define MBX_BN_APP(MASK)                                                                                            \
  hipLaunchKernelGGL(bn_bwd_apply_kernel<MASK>, dim3(grid_for(M * (C / 8))), dim3(kT), 0, mbx_s(stream), (cus)da,   \
                     ld_da, (cus)a, ld_a, (cus)y, (long long)M, C, mean, rstd, beta, m12, (us)dy, cm)
  if (!relu) MBX_BN_APP(0); else if (a) MBX_BN_APP(1); else MBX_BN_APP(2);
#undef MBX_BN_APP
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

static int ob_cus() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0)
      return 0;
    ncu = n;
  }
  return ncu;
}

extern "C" int mbx_bn_bwd_apply_rows(const float* stats, int rows, const void* da, int ld_da, const void* y, int64_t M, int C,
                                     const float* mean, const float* rstd, const float* relu_thr, float* dbeta, void* dy,
                                     const mbx_chan_map* da_map, mbx_stream_t stream) {
  if (!stats || rows <= 0 || rows > 16 || !da || !y || !mean || !rstd || !relu_thr || !dy || M <= 0 || C <= 0 || C % 8 ||
      ld_da % 8 || !al16(da) || !al16(y) || !al16(dy) || !al16(stats))
    return MBX_ERR_INVALID_ARG;
  if (C > kRowsMaxC) return MBX_ERR_UNSUPPORTED;
  ChanMap cm;
  if (to_chan_map(da_map, C, cm) != MBX_OK) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  static const int rows_blocks = env_int("MBX_BN_ROWS_BLOCKS", 1024);
  const int C8 = C / 8, rpi = kT / C8 > 0 ? kT / C8 : 1;
  long long g = (M + rpi - 1) / rpi;
  if (g > rows_blocks) g = rows_blocks;
  hipLaunchKernelGGL(bn_bwd_rows_kernel, dim3((unsigned)g), dim3(kT), (size_t)5 * C * sizeof(float), mbx_s(stream), stats, rows,
                     1.0 / (double)M, (cus)da, ld_da, (cus)y, (long long)M, C, mean, rstd, relu_thr, dbeta, (us)dy, cm);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" size_t mbx_bn_bwd_onepass_workspace_bytes(int C) {
  return C > 0 ? ((size_t)kObSlots * 2 * C + kObCtlWords) * sizeof(float) : 0;
}

// workgroups the grid may use: all CUs, or fewer when the caller keeps some free for a concurrent stream
static int ob_groups(int max_workgroups) {
  const int ncu = ob_cus();
  return (max_workgroups > 0 && max_workgroups < ncu) ? max_workgroups : ncu;
}

extern "C" int mbx_bn_bwd_onepass_supported(int64_t M, int C, int max_workgroups) {
  if (M <= 0 || C <= 0 || C % 8) return 0;
  const int ncu = ob_groups(max_workgroups);
  if (ncu <= 0) return 0;
  return ob_geom(M, C, ncu).nv <= kObMaxNV;
}

extern "C" int mbx_bn_bwd_onepass(const void* da, int ld_da, int relu, const void* y, int64_t M, int C, const float* mean,
                                  const float* rstd, const float* beta, float* dbeta, void* dy, void* ws,
                                  int max_workgroups, float* step_poison, mbx_stream_t stream) {
  return mbx_bn_bwd_onepass_mapped(da, ld_da, relu, y, M, C, mean, rstd, beta, dbeta, dy, ws, max_workgroups, step_poison,
                                   nullptr, stream);
}

extern "C" int mbx_bn_bwd_onepass_mapped(const void* da, int ld_da, int relu, const void* y, int64_t M, int C,
                                         const float* mean, const float* rstd, const float* beta, float* dbeta, void* dy,
                                         void* ws, int max_workgroups, float* step_poison, const mbx_chan_map* da_map,
                                         mbx_stream_t stream) {
  ChanMap cm;
  if (C <= 0 || to_chan_map(da_map, C, cm) != MBX_OK) return MBX_ERR_INVALID_ARG;
  const int map_max = chan_map_max(cm);
  if (!da || !y || !mean || !rstd || (relu && !beta) || !dy || !ws || M <= 0 || C <= 0 || C % 8 || ld_da % 8 || !al16(da) ||
      !al16(y) || !al16(dy) || !al16(ws) || !al16(mean) || !al16(rstd) || (relu && !al16(beta)))
    return MBX_ERR_INVALID_ARG;
  const int ncu = ob_groups(max_workgroups);
  if (ncu <= 0) return MBX_ERR_LAUNCH;
  const ObGeom g = ob_geom(M, C, ncu);
  if (g.nv > kObMaxNV) return MBX_ERR_UNSUPPORTED;
  MBX_ENTER();
  const size_t lds = ((size_t)g.rpi * g.C8 * 16 + (size_t)2 * C) * sizeof(float);
  static int fault = -1;
  if (fault < 0) { const char* e = getenv("MBX_DEBUG_BARRIER_FAULT"); fault = (e && e[0] == '1') ? 1 : 0; }
  const unsigned spin_limit = fault ? (1u << 10) : (1u << 22);
#define MBX_OB(NV, RELU)                                                                                               \
  hipLaunchKernelGGL((bn_bwd_onepass_kernel<NV, RELU>), dim3(g.G), dim3(kObT), lds, mbx_s(stream), (cus)da, ld_da,      \
                     (cus)y, (long long)M, C, mean, rstd, beta, dbeta, (us)dy, (float*)ws, g.rpi, g.rpb,               \
                     (float)(1.0 / (double)M), spin_limit, fault | (xcd_rows_on() << 1), step_poison, cm, map_max)
#define MBX_OB_NV(NV) do { if (relu) MBX_OB(NV, true); else MBX_OB(NV, false); } while (0)
  if (g.nv <= 2) MBX_OB_NV(2);
  else if (g.nv <= 4) MBX_OB_NV(4);
  else if (g.nv <= 8) MBX_OB_NV(8);
  else if (g.nv <= 12) MBX_OB_NV(12);
  else if (g.nv <= 16) MBX_OB_NV(16);
  else MBX_OB_NV(20);
#undef MBX_OB_NV
#undef MBX_OB
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

static int pool_args_ok(const void* x, int ldx, const void* y, int ldy, int N, int H, int W, int C, int k, int Ho, int Wo) {
  return x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && k > 0 && k < 16 &&
         Ho > 0 && Wo > 0 && al16(x) && al16(y) && (long long)N * H * W * (C / 8) < (1LL << 31) &&
         (long long)N * Ho * Wo * (C / 8) < (1LL << 31);
}

extern "C" int mbx_maxpool_fwd(const void* x, int64_t xs, int ldx, int N, int H, int W, int C, int k, int stride, void* y,
                               int64_t ys, int ldy, int Ho, int Wo, uint8_t* argmax, mbx_stream_t stream) {
  if (!pool_args_ok(x, ldx, y, ldy, N, H, W, C, k, Ho, Wo) || stride < 1) return MBX_ERR_INVALID_ARG;
  if ((Ho - 1) * stride + k > H || (Wo - 1) * stride + k > W) return MBX_ERR_INVALID_ARG;   // VALID only
  MBX_ENTER();
  if (k == 3 && stride == 2)
    hipLaunchKernelGGL((maxpool_fwd_kernel<3, 2>), dim3(grid_for((long long)N * Ho * Wo * (C / 8))), dim3(kT), 0, mbx_s(stream),
                       (cus)x, (long long)xs, ldx, N, H, W, C, k, stride, (us)y, (long long)ys, ldy, Ho, Wo, argmax);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<0, 0>), dim3(grid_for((long long)N * Ho * Wo * (C / 8))), dim3(kT), 0, mbx_s(stream),
                       (cus)x, (long long)xs, ldx, N, H, W, C, k, stride, (us)y, (long long)ys, ldy, Ho, Wo, argmax);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_maxpool_bwd(const void* dy, int64_t dys, int ld_dy, const uint8_t* argmax, int N, int H, int W, int C,
                               int k, int stride, int Ho, int Wo, void* dx, int64_t dxs, int ld_dx, int accumulate,
                               mbx_stream_t stream) {
  if (!pool_args_ok(dy, ld_dy, dx, ld_dx, N, H, W, C, k, Ho, Wo) || !argmax || stride < 1) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  if (k == 3 && stride == 2)
    hipLaunchKernelGGL((maxpool_bwd_kernel<3, 2>), dim3(grid_for((long long)N * H * W * (C / 8))), dim3(kT), 0, mbx_s(stream),
                       (cus)dy, (long long)dys, ld_dy, argmax, N, H, W, C, k, stride, Ho, Wo, (us)dx, (long long)dxs, ld_dx,
                       accumulate);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<0, 0>), dim3(grid_for((long long)N * H * W * (C / 8))), dim3(kT), 0, mbx_s(stream),
                       (cus)dy, (long long)dys, ld_dy, argmax, N, H, W, C, k, stride, Ho, Wo, (us)dx, (long long)dxs, ld_dx,
                       accumulate);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

static int pool_src_ok(const void* gy, int ld_gy, const uint8_t* argmax, int N, int H, int W, int Ho, int Wo, int C, PoolSrc& ps,
                       int64_t gys) {
  if (!gy || !argmax || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || ld_gy % 8 || !al16(gy) || (reinterpret_cast<uintptr_t>(argmax) & 7))
    return MBX_ERR_INVALID_ARG;
  if (Ho != (H - 3) / 2 + 1 || Wo != (W - 3) / 2 + 1) return MBX_ERR_INVALID_ARG;     // 3x3 / stride 2, VALID
  if ((long long)N * H * W >= (1LL << 31)) return MBX_ERR_UNSUPPORTED;
  ps.gy = (cus)gy; ps.gys = gys; ps.ld_gy = ld_gy; ps.argmax = argmax; ps.H = H; ps.W = W; ps.Ho = Ho; ps.Wo = Wo; ps.C = C;
  ps.Hb = (H + 1) / 2; ps.Wb = (W + 1) / 2;
  return MBX_OK;
}

extern "C" int mbx_bn_apply_maxpool(const void* y, int N, int H, int W, int C, const float* mean, const float* rstd,
                                    const float* beta, int relu, void* p, int64_t p_img_stride, int ld_p, int Ho, int Wo,
                                    uint8_t* argmax, mbx_stream_t stream) {
  if (!y || !p || !mean || !rstd || !beta || N <= 0 || H < 3 || W < 3 || C <= 0 || C % 8 || ld_p % 8 || !al16(y) || !al16(p) ||
      (argmax && (reinterpret_cast<uintptr_t>(argmax) & 7)))
    return MBX_ERR_INVALID_ARG;
  if (Ho != (H - 3) / 2 + 1 || Wo != (W - 3) / 2 + 1) return MBX_ERR_INVALID_ARG;     // 3x3 / stride 2, VALID
  if ((long long)N * Ho * Wo * (C / 8) >= (1LL << 31) || (long long)N * H * W * C >= (1LL << 40)) return MBX_ERR_UNSUPPORTED;
  MBX_ENTER();
  hipLaunchKernelGGL(bn_apply_maxpool3s2_kernel, dim3(grid_for((long long)N * Ho * Wo * (C / 8))), dim3(kT), 0, mbx_s(stream),
                     (cus)y, N, H, W, C, mean, rstd, beta, relu, (us)p, (long long)p_img_stride, ld_p, Ho, Wo, argmax);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_bwd_rows_pooled(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C > 2048) return MBX_ERR_INVALID_ARG;
  return bn_bwd_geom((long long)N * ((H + 1) / 2) * ((W + 1) / 2), C).rows;
}

extern "C" int mbx_bn_bwd_reduce_pooled(const void* gy, int64_t gy_img_stride, int ld_gy, const uint8_t* argmax, int N, int H,
                                        int W, int Ho, int Wo, int relu, const void* y, int C, const float* mean,
                                        const float* rstd, const float* beta, float* partial, mbx_stream_t stream) {
  PoolSrc ps;
  int st = pool_src_ok(gy, ld_gy, argmax, N, H, W, Ho, Wo, C, ps, gy_img_stride);
  if (st != MBX_OK) return st;
  if (!y || !al16(y) || !mean || !rstd || (relu && !beta) || !partial || C > 2048) return MBX_ERR_INVALID_ARG;
  const long long Mb = (long long)N * ps.Hb * ps.Wb;              // 2 x 2 pixel blocks: the "rows" of this launch
  const BnBwdGeom g = bn_bwd_geom(Mb, C);
  MBX_ENTER();
  const size_t lds = (size_t)g.rows_per_iter * g.C8 * 16 * sizeof(float);
  if (relu)
    hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<true>, dim3(g.rows), dim3(kT), lds, mbx_s(stream), ps, (cus)y, Mb, C, mean, rstd,
                       beta, g.rows_per_iter, g.rpb, partial);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<false>, dim3(g.rows), dim3(kT), lds, mbx_s(stream), ps, (cus)y, Mb, C, mean, rstd,
                       beta, g.rows_per_iter, g.rpb, partial);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_bwd_apply_pooled(const void* gy, int64_t gy_img_stride, int ld_gy, const uint8_t* argmax, int N, int H,
                                       int W, int Ho, int Wo, int relu, const void* y, int C, const float* mean,
                                       const float* rstd, const float* beta, const float* m12, void* dy, mbx_stream_t stream) {
  PoolSrc ps;
  int st = pool_src_ok(gy, ld_gy, argmax, N, H, W, Ho, Wo, C, ps, gy_img_stride);
  if (st != MBX_OK) return st;
  if (!y || !al16(y) || !mean || !rstd || (relu && !beta) || !m12 || !dy || !al16(dy)) return MBX_ERR_INVALID_ARG;
  const long long M = (long long)N * H * W, Mb = (long long)N * ps.Hb * ps.Wb;
  MBX_ENTER();
  if (relu)
    hipLaunchKernelGGL(bn_bwd_apply_pool_kernel<true>, dim3(grid_for(Mb * (C / 8))), dim3(kT), 0, mbx_s(stream), ps, (cus)y, Mb, C, M,
                       mean, rstd, beta, m12, (us)dy);
  else
    hipLaunchKernelGGL(bn_bwd_apply_pool_kernel<false>, dim3(grid_for(Mb * (C / 8))), dim3(kT), 0, mbx_s(stream), ps, (cus)y, Mb, C, M,
                       mean, rstd, beta, m12, (us)dy);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_avgpool_fwd(const void* x, int64_t xs, int ldx, int N, int H, int W, int C, int k, int pad, void* y,
                               int64_t ys, int ldy, int Ho, int Wo, mbx_stream_t stream) {
  if (!pool_args_ok(x, ldx, y, ldy, N, H, W, C, k, Ho, Wo) || pad < 0 || pad >= k) return MBX_ERR_INVALID_ARG;
  if (Ho != H + 2 * pad - k + 1 || Wo != W + 2 * pad - k + 1) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long long)N * Ho * Wo * (C / 8))), dim3(kT), 0, mbx_s(stream),
                     (cus)x, (long long)xs, ldx, N, H, W, C, k, pad, (us)y, (long long)ys, ldy, Ho, Wo);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_avgpool_bwd(const void* dy, int64_t dys, int ld_dy, int N, int H, int W, int C, int k, int pad, int Ho,
                               int Wo, void* dx, int64_t dxs, int ld_dx, int accumulate, mbx_stream_t stream) {
  if (!pool_args_ok(dy, ld_dy, dx, ld_dx, N, H, W, C, k, Ho, Wo) || pad < 0 || pad >= k) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long long)N * H * W * (C / 8))), dim3(kT), 0, mbx_s(stream),
                     (cus)dy, (long long)dys, ld_dy, N, H, W, C, k, pad, Ho, Wo, (us)dx, (long long)dxs, ld_dx, accumulate);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_relu_mask(void* g, int ld_g, const void* a, int ld_a, int64_t M, int C, mbx_stream_t stream) {
  if (!g || !a || M <= 0 || C <= 0 || C % 8 || ld_g % 8 || ld_a % 8 || !al16(g) || !al16(a)) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(M * (C / 8))), dim3(kT), 0, mbx_s(stream), (us)g, ld_g, (cus)a, ld_a,
                     (long long)M, C);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_pack_input(const float* img, int64_t pixels, void* out, mbx_stream_t stream) {
  if (!img || !out || pixels <= 0 || !al16(out)) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(pixels)), dim3(kT), 0, mbx_s(stream), img, (long long)pixels, (us)out);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_head_gather(const float* h, int ld_h, int N, int cells, int k, int P, int off, float* locs,
                               float* logits, mbx_stream_t stream) {
  if (!h || !locs || !logits || N <= 0 || cells <= 0 || k <= 0 || ld_h < 5 * k || off < 0 || off + cells * k > P)
    return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(head_gather_kernel, dim3(grid_for((long long)N * cells * k)), dim3(kT), 0, mbx_s(stream), h, ld_h, N,
                     cells, k, P, off, locs, logits);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_head_scatter(const float* d_locs, const float* d_logits, int N, int cells, int k, int P, int off, void* g,
                                int ld_g, mbx_stream_t stream) {
  if (!d_locs || !d_logits || !g || N <= 0 || cells <= 0 || k <= 0 || ld_g < 5 * k || off < 0 || off + cells * k > P)
    return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(head_scatter_kernel, dim3(grid_for((long long)N * cells * ld_g)), dim3(kT), 0, mbx_s(stream), d_locs,
                     d_logits, N, cells, k, P, off, (us)g, ld_g);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

static int fill_head_table(const mbx_head* heads, int n_heads, int N, int P, bool scatter, HeadTable& t) {
  if (!heads || n_heads <= 0 || n_heads > kMaxHeads || N <= 0 || P <= 0) return MBX_ERR_INVALID_ARG;
  long long acc = 0;
  t.n = n_heads;
  for (int j = 0; j < n_heads; ++j) {
    const mbx_head& H = heads[j];
    if (H.cells <= 0 || H.k <= 0 || H.off < 0 || H.off + H.cells * H.k > P) return MBX_ERR_INVALID_ARG;
    if (scatter ? (!H.g || H.ld_g < 5 * H.k) : (!H.h || H.ld_h < 5 * H.k)) return MBX_ERR_INVALID_ARG;
    t.h[j] = H.h; t.g[j] = reinterpret_cast<unsigned short*>(H.g);
    t.ld_h[j] = H.ld_h; t.ld_g[j] = H.ld_g; t.cells[j] = H.cells; t.k[j] = H.k; t.off[j] = H.off;
    t.start[j] = (int)acc;
    acc += (long long)N * H.cells * (scatter ? H.ld_g : H.k);
    if (acc >= (1LL << 31)) return MBX_ERR_UNSUPPORTED;
  }
  for (int j = n_heads; j <= kMaxHeads; ++j) t.start[j] = (int)acc;
  return MBX_OK;
}

extern "C" int mbx_head_gather_all(const mbx_head* heads, int n_heads, int N, int P, float* locs, float* logits,
                                   mbx_stream_t stream) {
  if (!locs || !logits) return MBX_ERR_INVALID_ARG;
  HeadTable t;
  const int st = fill_head_table(heads, n_heads, N, P, false, t);
  if (st != MBX_OK) return st;
  MBX_ENTER();
  hipLaunchKernelGGL(head_gather_all_kernel, dim3(grid_for(t.start[t.n])), dim3(kT), 0, mbx_s(stream), t, P, locs, logits);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_head_scatter_all(const float* d_locs, const float* d_logits, const mbx_head* heads, int n_heads, int N,
                                    int P, mbx_stream_t stream) {
  if (!d_locs || !d_logits) return MBX_ERR_INVALID_ARG;
  HeadTable t;
  const int st = fill_head_table(heads, n_heads, N, P, true, t);
  if (st != MBX_OK) return st;
  MBX_ENTER();
  hipLaunchKernelGGL(head_scatter_all_kernel, dim3(grid_for(t.start[t.n])), dim3(kT), 0, mbx_s(stream), t, P, d_locs, d_logits);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_step_begin(float* grads, int64_t n_grads, float* bn_ws, int64_t n_ws, int64_t ctl_index,
                              uint64_t* timeouts_total, float* zero_scalar, mbx_stream_t stream) {
  if (!grads || n_grads <= 0 || n_grads % 4 || n_ws < 0 || n_ws % 4 || (n_ws && !bn_ws) || !al16(grads) || (bn_ws && !al16(bn_ws)) ||
      (timeouts_total && (ctl_index < 0 || ctl_index >= n_grads)))
    return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(step_begin_kernel, dim3(grid_for((n_grads + n_ws) / 4)), dim3(kT), 0, mbx_s(stream), grads,
                     (long long)n_grads, bn_ws, (long long)n_ws, (long long)ctl_index,
                     reinterpret_cast<unsigned long long*>(timeouts_total), zero_scalar);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_filter_prepare(const void* w_bf16, void* w_dgrad, const mbx_filter_entry* table, int n_entries,
                                  int total_blocks, mbx_stream_t stream) {
  if (!w_bf16 || !w_dgrad || !table || n_entries <= 0 || total_blocks <= 0) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(filter_prepare_kernel, dim3(total_blocks), dim3(kT), 0, mbx_s(stream), (cus)w_bf16, (us)w_dgrad, table,
                     n_entries);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_rmsprop_ema_step(float* w, const float* g, float* ms, float* mom, float* ema, void* w_bf16, int64_t n,
                                    float lr, float decay, float momentum, float eps, float wd, float ema_decay,
                                    int trainable, float* reg_loss, const float* skip_ctl, mbx_stream_t stream) {
  if (!w || n <= 0 || (trainable && (!g || !ms))) return MBX_ERR_INVALID_ARG;
  if (trainable && momentum != 0.f && !mom) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(rmsprop_ema_kernel, dim3(grid_for(n)), dim3(kT), 0, mbx_s(stream), w, g, ms, momentum != 0.f ? mom : nullptr,
                     ema, (us)w_bf16, (long long)n, lr, decay, momentum, eps, wd, ema_decay, trainable, reg_loss, skip_ctl);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_moving_update(float* moving_mean, float* moving_var, const float* batch_mean, const float* batch_var,
                                    int64_t n, float decay, const float* skip_ctl, uint64_t* skipped_steps,
                                    float* ema_mean, float* ema_var, float ema_decay, mbx_stream_t stream) {
  if (!moving_mean || !moving_var || !batch_mean || !batch_var || n <= 0 || decay < 0.f) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(bn_moving_update_kernel, dim3(grid_for(n)), dim3(kT), 0, mbx_s(stream), moving_mean, moving_var,
                     batch_mean, batch_var, (long long)n, decay, skip_ctl, reinterpret_cast<unsigned long long*>(skipped_steps),
                     ema_mean, ema_var, ema_decay);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_ema_update(float* ema, const float* value, int64_t n, float ema_decay, const float* skip_ctl,
                              mbx_stream_t stream) {
  if (!ema || !value || n <= 0) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  hipLaunchKernelGGL(ema_update_kernel, dim3(grid_for(n)), dim3(kT), 0, mbx_s(stream), ema, value, (long long)n, ema_decay, skip_ctl);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_apply_fused_mapped(const void* stats_partial, int rows, int64_t count, float eps, float decay,
                                         const void* y, int64_t M, int C, const float* beta, int relu, void* a, int ld_a,
                                         const mbx_chan_map* a_map, float* mean, float* rstd, float* mmean, float* mvar,
                                         float* relu_thr, mbx_stream_t stream) {
  if (!stats_partial || !y || !a || !beta || !mean || !rstd || rows <= 0 || M <= 0 || C <= 0 || C % 8 || ld_a % 8 ||
      count <= 0 || !al16(y) || !al16(a) || !al16(stats_partial))
    return MBX_ERR_INVALID_ARG;
  if (rows > 16 || C > kRowsMaxC) return MBX_ERR_UNSUPPORTED;
  ChanMap cm;
  if (to_chan_map(a_map, C, cm) != MBX_OK) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  static const int rows_blocks = env_int("MBX_BN_ROWS_BLOCKS", 1024);
  // whole-row sweeps; every workgroup reduces the few fixed-point rows of all C channels itself (bn_apply_rows_kernel)
  const int C8 = C / 8, rpi = kT / C8 > 0 ? kT / C8 : 1;
  long long g = (M + rpi - 1) / rpi;
  if (g > rows_blocks) g = rows_blocks;
  hipLaunchKernelGGL(bn_apply_rows_kernel, dim3((unsigned)g), dim3(kT), (size_t)3 * C * sizeof(float), mbx_s(stream),
                     reinterpret_cast<const float*>(stats_partial), rows, 1.0 / (double)count, eps, decay, (cus)y, (long long)M, C,
                     beta, relu, (us)a, ld_a, mean, rstd, mmean, mvar, cm, relu_thr, xcd_rows_on());
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_bn_apply_fused(const float* stats_partial, int rows, int64_t count, float eps, float decay, const void* y,
                                  int64_t M, int C, const float* beta, int relu, void* a, int ld_a, float* mean,
                                  float* rstd, float* mmean, float* mvar, mbx_stream_t stream) {
  if (!stats_partial || !y || !a || !beta || !mean || !rstd || rows <= 0 || M <= 0 || C <= 0 || C % 8 || ld_a % 8 ||
      count <= 0 || !al16(y) || !al16(a))
    return MBX_ERR_INVALID_ARG;
  // Fused (one launch: every workgroup re-reduces the partial rows of its 64 channels) only for few partial rows.  In
  // isolation (tools/bnf_bench.py) the fused launch wins up to ~1000 rows on small tensors (8 vs 13 us); inside the
  // captured step it LOSES 0.28 ms (round 2, same-box A/B, MBX_BN_FUSED_ROWS=1024): a graph-internal kernel boundary
  // costs ~1.5 us, the 5 us finalize overlaps the conv's tail, and 256 long workgroups stream a cold tensor badly.
  static const int fused_rows = env_int("MBX_BN_FUSED_ROWS", 16);
  static const long long fused_elems = env_int("MBX_BN_FUSED_KELEMS", 6200) * 1000LL;
  if (rows > 16 && (rows > fused_rows || M * C > fused_elems)) {
    int st = mbx_bn_finalize(stats_partial, rows, C, count, eps, decay, mean, rstd, mmean, mvar, stream);
    if (st != MBX_OK) return st;
    return mbx_bn_apply(y, M, C, mean, rstd, beta, relu, a, ld_a, stream);
  }
  MBX_ENTER();
  int groups, chunks, rpc;
  bn_fused_grid(M, C, rows, groups, chunks, rpc);
  hipLaunchKernelGGL(bn_apply_fused_kernel, dim3(groups, chunks), dim3(kT), 0, mbx_s(stream), stats_partial, rows,
                     1.0 / (double)count, eps, decay, (cus)y, (long long)M, C, beta, relu, (us)a, ld_a, mean, rstd, mmean,
                     mvar, rpc, ChanMap{}, nullptr);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}
