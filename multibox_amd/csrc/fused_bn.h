// BATCH-NORM PHASES FUSED INTO THE CONVOLUTION LAUNCHES behind a one-shot grid barrier (round 6).
//
// Forward (FusedApply): a training-mode convolution ADDS its tile sums into the layer's fixed-point statistics rows (integer
// atomics, conv_common.h stats_write); the normalise + beta + relu pass that followed as its own launch (bn_apply_rows_kernel:
// 144 launches of 5-10 us per step, each re-reading y from the Infinity Cache behind a dispatch and a rows prologue) runs as
// the TAIL of the convolution launch instead: every workgroup waits for its atomics, meets the others at the grid barrier,
// reduces the rows of all channels itself (device-scope loads), and then sweeps THE TILES IT WROTE ITSELF -- y comes back
// from this CU's own L2 (written microseconds ago by the same workgroup: coherent without any fence, and never from HBM) and
// the activation goes out.  Same expressions as bn_apply_rows_kernel: bit-identical mean / rstd / threshold / activations.
//
// A tile OWNER sweep, not a row sweep: what a workgroup reads back is only ever what it stored itself, so nothing but the
// statistics (atomics) crosses workgroups inside the launch and no L2 write-back / invalidate is needed (grid_barrier.h).
#pragma once
#include "grid_barrier.h"

namespace {

// Statistics of ALL C channels of the launch -> s_par[3][C] (mean, rstd, beta) in LDS, from the `rows` fixed-point rows
// [rows][ld][2] int64 the launch's tiles added into.  Workgroup 0 publishes.  The expressions are bn_apply_rows_kernel's.
template <int NT>
__device__ __forceinline__ void fused_stats_to_lds(const FusedApply& f, const float* stats, const int rows, const int ld, const int C,
                                                   float* s_par, const bool timed_out) {
  for (int ch = threadIdx.x; ch < C; ch += NT) {
    const long long* src = reinterpret_cast<const long long*>(stats) + (size_t)ch * 2;
    long long i1 = 0, i2 = 0;
    bool bad = false;
    for (int r = 0; r < rows; ++r) {
      const long long v1 = gb_ld(src + (size_t)r * ld * 2), v2 = gb_ld(src + (size_t)r * ld * 2 + 1);
      i1 += v1; i2 += v2;
      bad |= v2 < 0;
    }
    const double s1 = (double)i1 * (1.0 / 1048576.0), s2 = (double)i2 * (1.0 / 1048576.0);
    const double mu = s1 * f.inv_count;
    double var = s2 * f.inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    const bool poisoned = bad || i2 < 0 || timed_out;          // (a workgroup that gave up on the barrier holds partial sums)
    if (poisoned) var = (double)__builtin_nanf("");
    const float fm = poisoned ? __builtin_nanf("") : (float)mu, fr = poisoned ? __builtin_nanf("") : (float)(1.0 / sqrt(var + (double)f.eps));
    const float be = f.beta[ch];
    s_par[ch] = fm; s_par[C + ch] = fr; s_par[2 * C + ch] = be;
    if (blockIdx.x == 0) {
      f.mean[ch] = fm; f.rstd[ch] = fr;
      if (f.thr) f.thr[ch] = f.relu ? fm - be / fr : -__builtin_inff();
      if (f.decay < 0.f) {                                                   // store mode (bn_finalize_kernel)
        if (f.mmean) f.mmean[ch] = fm;
        if (f.mvar) f.mvar[ch] = (float)var;
      } else {
        if (f.mmean) f.mmean[ch] -= (1.0f - f.decay) * (f.mmean[ch] - fm);
        if (f.mvar) f.mvar[ch] -= (1.0f - f.decay) * (f.mvar[ch] - (float)var);
      }
    }
  }
}

// One rectangular region the workgroup stored itself: rows [m0, m0 + nrows) x channels [c0, c0 + nch) of y = [M][ldy]
// (nch a multiple of 8) -> a.  A lane owns one 8-channel group; the workgroup sweeps NT / (nch / 8) rows per pass.
template <int NT>
__device__ __forceinline__ void fused_apply_region(const FusedApply& f, const unsigned short* y, const int ldy, const int M, const int C,
                                                   const float* s_par, const int m0, const int nrows, const int c0, const int nch) {
  const int oct = nch >> 3;
  const int rpp = NT / oct;
  const int vc = threadIdx.x % oct, rr = threadIdx.x / oct;
  const int c = c0 + (vc << 3);
  if (rr >= rpp || c >= C) return;
  float mu[8], rs[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mu[j] = s_par[c + j]; rs[j] = s_par[C + c + j]; be[j] = s_par[2 * C + c + j]; }
  const int mend = (m0 + nrows < M) ? m0 + nrows : M;
  for (int m = m0 + rr; m < mend; m += rpp) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(y + (size_t)m * ldy + c);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = (bf_lo(w[j]) - mu[2 * j]) * rs[2 * j] + be[2 * j];
      const float hi = (bf_hi(w[j]) - mu[2 * j + 1]) * rs[2 * j + 1] + be[2 * j + 1];
      q[j] = f.relu ? pack2bf(relu_f(lo), relu_f(hi)) : pack2bf(lo, hi);
    }
    *reinterpret_cast<u32x4*>(f.a + (size_t)m * f.ld_a + c) = u32x4{q[0], q[1], q[2], q[3]};
  }
}

// The meeting point: all of the workgroup's threads call it when the workgroup's last tile has been stored and its last
// statistics atomics issued.  Returns (to every thread) whether this workgroup gave up on the barrier.
template <int NT>
__device__ __forceinline__ bool fused_grid_meet(const FusedApply& f, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // this wave's stores and atomics are acknowledged
  __syncthreads();
  if (threadIdx.x == 0)
    *s_flag = grid_barrier_arrive_wait(f.bar, gridDim.x, blockIdx.x, f.spin_limit, f.fault, f.step_poison) ? 1 : 0;
  __syncthreads();
  return *s_flag != 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// Backward (FusedBwd): the data gradient that WRITES the activation gradient da of batch-norm layers runs their backward as
// its tail.  slim.batch_norm + relu on the way back (train.py:94-99, 263) needs, per channel, sum g and sum g xhat over all
// pixels (g = da where the activation was positive) before it can write one element of dy = rstd (g - mean g - xhat mean g xhat)
// -- a launch of its own with a grid barrier inside (bn_bwd_onepass_kernel: 141 launches of 9-25 us per step, da and y read
// from the Infinity Cache / HBM behind a dispatch).  Here every workgroup of the data gradient, when its last tile is stored,
// sweeps THE TILES IT WROTE (da back from its own L2, y from memory) for its share of the two sums, adds them to the layers'
// accumulators (float atomics, as the one-launch backward does), meets the others at the barrier, reads the totals and sweeps
// its tiles once more to write dy.  The batch-norm backward launch of those layers is gone; da still goes through memory
// (it is re-read by its writer only), dy is written where the layer's own data gradient and weight gradient expect it.
// Same expressions as bn_bwd_onepass_kernel; like it, not run-to-run reproducible in the last bits (atomic order).

// The segment table lives in the KERNEL ARGUMENTS (ConvK::fb: arrays of four), and a lane needs the entry of ITS channels: a
// run-time index.  Indexing the by-value argument struct dynamically makes the compiler keep a private copy of ALL of ConvK
// (968 bytes of scratch stores at the start of EVERY launch of the kernel, fused or not -- seen in the ISA); so the tail first
// copies the table from the kernarg segment (ConvK is the first argument: offset 0) into LDS, word by word, and indexes that.
struct FbShared { FusedBwd f; int flag; int pad[3]; };

__device__ __forceinline__ FbShared* fused_bwd_stage(void* lds_area) {
  typedef const __attribute__((address_space(4))) unsigned* kernarg_words;
  const kernarg_words ka = (kernarg_words)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvK, fb) / 4;
  unsigned* dst = reinterpret_cast<unsigned*>(lds_area);
  for (int i = threadIdx.x; i < (int)(sizeof(FusedBwd) / 4); i += blockDim.x) dst[i] = ka[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // (also: the workgroup's own stores of da are complete before its tail reads them back)
  __syncthreads();
  return reinterpret_cast<FbShared*>(lds_area);
}

__device__ __forceinline__ int fb_seg(const FusedBwd& f, const int c) { return (c >= f.cb[1]) + (c >= f.cb[2]) + (c >= f.cb[3]); }

struct FbLane { float mu[8], rs[8], be[8]; const unsigned short* y; long long ldy; int sg, cr, relu; };

__device__ __forceinline__ void fb_lane_setup(const FusedBwd& f, const int c, const int C, FbLane& L) {
  const int cc = c < C ? c : 0;
  const int sg = fb_seg(f, cc), cr = cc - f.cb[sg];
  L.sg = sg; L.cr = cr; L.relu = f.relu[sg];
  L.y = f.y[sg] + cr;
  L.ldy = f.ldy[sg];
  const float* mean = f.mean[sg];
  const float* rstd = f.rstd[sg];
  const float* beta = f.beta[sg];
#pragma unroll
  for (int j = 0; j < 8; ++j) { L.mu[j] = mean[cr + j]; L.rs[j] = rstd[cr + j]; L.be[j] = L.relu ? beta[cr + j] : 0.f; }
}

// Phase 1 of one region the workgroup stored (rows [m0, m0 + nrows) x channels [c0, c0 + nch) of da = [M][ldg]): the two sums
// per channel -> LDS partials [row lane][octet][16] -> column sums -> atomics.  s_red: NT * 16 floats of LDS.
template <int NT>
__device__ __forceinline__ void fused_bwd_sums_region(const FusedBwd& f, const unsigned short* da, const int ldg, const int M, const int C,
                                                      float* s_red, const int m0, const int nrows, const int c0, const int nch) {
  const int oct = nch >> 3;
  const int rpp = NT / oct;
  const int vc = threadIdx.x % oct, rr = threadIdx.x / oct;
  const int c = c0 + (vc << 3);
  const bool active = rr < rpp && c < C;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (active) {
    FbLane L;
    fb_lane_setup(f, c, C, L);
    const int mend = (m0 + nrows < M) ? m0 + nrows : M;
    for (int m = m0 + rr; m < mend; m += rpp) {
      const u32x4 vg = *reinterpret_cast<const u32x4*>(da + (size_t)m * ldg + c);
      const u32x4 vy = *reinterpret_cast<const u32x4*>(L.y + (size_t)m * L.ldy);
      const unsigned wg[4] = {vg.x, vg.y, vg.z, vg.w}, wy[4] = {vy.x, vy.y, vy.z, vy.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = (j & 1) ? bf_hi(wg[j >> 1]) : bf_lo(wg[j >> 1]);
        const float yy = (j & 1) ? bf_hi(wy[j >> 1]) : bf_lo(wy[j >> 1]);
        const float xh = (yy - L.mu[j]) * L.rs[j];
        const float gj = (!L.relu || xh + L.be[j] > 0.f) ? g : 0.f;
        s1[j] += gj;
        s2[j] += gj * xh;
      }
    }
  }
  if (rr < rpp) {
    float* o = s_red + ((size_t)rr * oct + vc) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = s1[j]; o[8 + j] = s2[j]; }
  }
  __syncthreads();
  // column sums: thread e < 16 oct owns value (octet e / 16, j = e % 16: s1 of channel j, or s2 of channel j - 8)
  for (int e = threadIdx.x; e < 16 * oct; e += NT) {
    float t = 0.f;
    for (int r = 0; r < rpp; ++r) t += s_red[(size_t)r * oct * 16 + e];
    const int ch = c0 + ((e >> 4) << 3) + (e & 7);
    if (ch < C) {
      const int sg = fb_seg(f, ch);
      unsafeAtomicAdd(f.acc[sg] + ((size_t)(blockIdx.x & (kFbSlots - 1)) * 2 + ((e >> 3) & 1)) * f.acc_ld[sg] + (ch - f.cb[sg]), t);   // (hardware float add, fire and forget)
    }
  }
  __syncthreads();
}

// Totals of all C channels of the launch -> s_tot[2][C] (m1 = mean g, m2 = mean g xhat); workgroup 0 adds sum g to dbeta.
template <int NT>
__device__ __forceinline__ void fused_bwd_totals_to_lds(const FusedBwd& f, const int C, float* s_tot, const bool timed_out) {
  const float poison = timed_out ? __builtin_nanf("") : 0.f;
  for (int ch = threadIdx.x; ch < C; ch += NT) {
    const int sg = fb_seg(f, ch), cr = ch - f.cb[sg];
    const float* acc = f.acc[sg];
    const int ld = f.acc_ld[sg];
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int sl = 0; sl < kFbSlots; ++sl) { t1 += gb_ld(acc + ((size_t)sl * 2) * ld + cr); t2 += gb_ld(acc + ((size_t)sl * 2 + 1) * ld + cr); }
    t1 += poison;
    s_tot[ch] = t1 * f.inv_M;
    s_tot[C + ch] = t2 * f.inv_M;
    if (blockIdx.x == 0 && f.dbeta[sg]) f.dbeta[sg][cr] += t1;
  }
}

// Phase 2 of one region: dy = rstd (g - m1 - xhat m2) from da and y re-read (da: this CU's own L2).
template <int NT>
__device__ __forceinline__ void fused_bwd_apply_region(const FusedBwd& f, const unsigned short* da, const int ldg, const int M, const int C,
                                                       const float* s_tot, const int m0, const int nrows, const int c0, const int nch) {
  const int oct = nch >> 3;
  const int rpp = NT / oct;
  const int vc = threadIdx.x % oct, rr = threadIdx.x / oct;
  const int c = c0 + (vc << 3);
  if (rr < rpp && c < C) {
    FbLane L;
    fb_lane_setup(f, c, C, L);
    unsigned short* dy = f.dy[L.sg] + L.cr;
    const long long lddy = f.lddy[L.sg];
    float m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m1[j] = s_tot[c + j]; m2[j] = s_tot[C + c + j]; }
    const int mend = (m0 + nrows < M) ? m0 + nrows : M;
    for (int m = m0 + rr; m < mend; m += rpp) {
      const u32x4 vg = *reinterpret_cast<const u32x4*>(da + (size_t)m * ldg + c);
      const u32x4 vy = *reinterpret_cast<const u32x4*>(L.y + (size_t)m * L.ldy);
      const unsigned wg[4] = {vg.x, vg.y, vg.z, vg.w}, wy[4] = {vy.x, vy.y, vy.z, vy.w};
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = (j & 1) ? bf_hi(wg[j >> 1]) : bf_lo(wg[j >> 1]);
        const float yy = (j & 1) ? bf_hi(wy[j >> 1]) : bf_lo(wy[j >> 1]);
        const float xh = (yy - L.mu[j]) * L.rs[j];
        const float gj = (!L.relu || xh + L.be[j] > 0.f) ? g : 0.f;
        o[j] = L.rs[j] * (gj - m1[j] - xh * m2[j]);
      }
      *reinterpret_cast<u32x4*>(dy + (size_t)m * lddy) = u32x4{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
  }
}

// The meeting point of the backward tail: the workgroup's atomics are acknowledged, then the grid barrier.
template <int NT>
__device__ __forceinline__ bool fused_bwd_meet(FbShared* sh) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    sh->flag = grid_barrier_arrive_wait(sh->f.bar, gridDim.x, blockIdx.x, sh->f.spin_limit, sh->f.fault, sh->f.step_poison) ? 1 : 0;
  __syncthreads();
  return sh->flag != 0;
}

// The whole tail for a workgroup whose stored regions `for_each_region(fn)` enumerates (fn(m0, nrows, c0, nch)).
// lds: NT * 64 + sizeof(FbShared) bytes.
template <int NT, class Regions>
__device__ __forceinline__ void fused_bwd_tail(const ConvK& p, void* lds, const Regions& for_each_region) {
  float* s_red = reinterpret_cast<float*>(lds);
  FbShared* sh = fused_bwd_stage(s_red + NT * 16);
  const FusedBwd& f = sh->f;
  const unsigned short* da = reinterpret_cast<const unsigned short*>(p.y);
  const int ldg = p.ldy, M = p.M, C = p.C_out;
  for_each_region([&](const int m0, const int nrows, const int c0, const int nch) {
    fused_bwd_sums_region<NT>(f, da, ldg, M, C, s_red, m0, nrows, c0, nch);
  });
  const bool timed_out = fused_bwd_meet<NT>(sh);
  fused_bwd_totals_to_lds<NT>(f, C, s_red, timed_out);
  __syncthreads();
  for_each_region([&](const int m0, const int nrows, const int c0, const int nch) {
    fused_bwd_apply_region<NT>(f, da, ldg, M, C, s_red, m0, nrows, c0, nch);
  });
}

}  // namespace
