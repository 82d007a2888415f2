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

// 16-byte load that is COHERENT AT DEVICE SCOPE (sc1: served at the memory side like the atomics that produced the data) but an
// ordinary load to the compiler: a run of them is issued back to back and waited for once.  (`__hip_atomic_load` compiles to
// the same instruction, but as an atomic it is never batched: a rolled loop of them waited for every single round trip --
// sixteen in a row cost the first version of this tail ~10 us.)
constexpr int kAuxSc1 = 16;                              // cache-policy bit of buffer instructions on gfx940+: sc1
__device__ __forceinline__ u32x4 ld16_agent(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, kAuxSc1);
}
__device__ __forceinline__ float ld4_agent(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, kAuxSc1));
}
// debug probes (MBX_FUSED_PROBE, wrong results): bits of FusedApply::fault / FusedBwd::fault
constexpr int kProbeNoWait = 16, kProbeNoTotals = 32, kProbeNoSweep = 64, kProbeNoSums = 128;

// Statistics of ALL C channels of the launch -> s_par[3][C] (mean, rstd, beta) in LDS, from the `rows` (<= 16) fixed-point rows
// [rows][ld][2] int64 the launch's tiles added into: every row's load of a channel in flight together.  Workgroup 0 publishes.
// The expressions are bn_apply_rows_kernel's.
template <int NT>
__device__ __forceinline__ void fused_stats_to_lds(const FusedApply& f, const float* stats, const int rows, const int ld, const int C,
                                                   float* s_par, const bool timed_out) {
  const __amdgpu_buffer_rsrc_t sr = make_rsrc(stats, (f.fault & kProbeNoTotals) ? 0u : (unsigned)(rows * ld * 16));
  for (int ch = threadIdx.x; ch < C; ch += NT) {
    u32x4 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = ld16_agent(sr, r < rows ? (unsigned)((r * ld + ch) * 16) : kOOB);      // (past the table: zeros)
    const float be = f.beta[ch];
    long long i1 = 0, i2 = 0;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long v1 = (long long)(((unsigned long long)v[r].y << 32) | v[r].x), v2 = (long long)(((unsigned long long)v[r].w << 32) | v[r].z);
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

// A lane's share of one rectangular region the workgroup stored itself -- rows [m0, m0 + nrows) x channels [c0, c0 + nch) of
// y = [M][ldy], nch a multiple of 8: a lane owns one 8-channel group, the workgroup sweeps NT / (nch / 8) rows per pass -- in
// chunks of kFaChunk rows whose loads are in flight together.
constexpr int kFaChunk = 5;
struct FaLane { int c, rpp, mfirst, mend; bool active; };
template <int NT>
__device__ __forceinline__ FaLane fa_lane(const int M, const int C, const int m0, const int nrows, const int c0, const int nch) {
  FaLane L;
  const int oct = nch >> 3;
  L.rpp = NT / oct;
  const int vc = threadIdx.x % oct, rr = threadIdx.x / oct;
  L.c = c0 + (vc << 3);
  L.active = rr < L.rpp && L.c < C;
  L.mend = (m0 + nrows < M) ? m0 + nrows : M;
  L.mfirst = m0 + rr;
  return L;
}
__device__ __forceinline__ void fa_load_chunk(const FaLane& L, const unsigned short* y, const int ldy, const int mbase, u32x4 (&v)[kFaChunk]) {
#pragma unroll
  for (int i = 0; i < kFaChunk; ++i) {
    const int m = mbase + i * L.rpp;
    v[i] = (L.active && m < L.mend) ? *reinterpret_cast<const u32x4*>(y + (size_t)m * ldy + L.c) : u32x4{0u, 0u, 0u, 0u};
  }
}
__device__ __forceinline__ void fa_finish_chunk(const FusedApply& f, const FaLane& L, const int C, const float* s_par, const int mbase,
                                                const u32x4 (&v)[kFaChunk]) {
  if (!L.active) return;
  float mu[8], rs[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mu[j] = s_par[L.c + j]; rs[j] = s_par[C + L.c + j]; be[j] = s_par[2 * C + L.c + j]; }
#pragma unroll
  for (int i = 0; i < kFaChunk; ++i) {
    const int m = mbase + i * L.rpp;
    if (m < L.mend) {
      const unsigned w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      unsigned q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = (bf_lo(w[j]) - mu[2 * j]) * rs[2 * j] + be[2 * j];
        const float hi = (bf_hi(w[j]) - mu[2 * j + 1]) * rs[2 * j + 1] + be[2 * j + 1];
        q[j] = f.relu ? pack2bf(relu_f(lo), relu_f(hi)) : pack2bf(lo, hi);
      }
      *reinterpret_cast<u32x4*>(f.a + (size_t)m * f.ld_a + L.c) = u32x4{q[0], q[1], q[2], q[3]};
    }
  }
}
template <int NT>
__device__ __forceinline__ void fused_apply_region(const FusedApply& f, const unsigned short* y, const int ldy, const int M, const int C,
                                                   const float* s_par, const int m0, const int nrows, const int c0, const int nch,
                                                   const int skip_chunks = 0) {
  const FaLane L = fa_lane<NT>(M, C, m0, nrows, c0, nch);
  for (int mb = L.mfirst + skip_chunks * kFaChunk * L.rpp; mb < L.mend; mb += kFaChunk * L.rpp) {
    u32x4 v[kFaChunk];
    fa_load_chunk(L, y, ldy, mb, v);
    fa_finish_chunk(f, L, C, s_par, mb, v);
  }
}

// The meeting point: all of the workgroup's threads call it when the workgroup's last tile has been stored and its last
// statistics atomics issued.  Returns (to every thread) whether this workgroup gave up on the barrier.
template <int NT>
__device__ __forceinline__ bool fused_grid_meet(const FusedApply& f, int* s_flag) {
  if (threadIdx.x == 0)
    *s_flag = (f.fault & kProbeNoWait) ? 0 : (grid_barrier_arrive_wait(f.bar, gridDim.x, blockIdx.x, f.spin_limit, f.fault & 1, f.step_poison) ? 1 : 0);
  __syncthreads();
  return *s_flag != 0;
}

// The whole forward tail of a workgroup whose stored regions `for_each_region(fn)` enumerates (fn(m0, nrows, c0, nch)):
//   own stores + atomics acknowledged -> the FIRST region's first chunk of y back from L2 into registers (in flight across the
//   barrier: they are this workgroup's own stores) -> grid barrier -> statistics rows -> the sweeps.
// lds: 3 C floats + 16 bytes.
template <int NT, class Regions>
__device__ __forceinline__ void fused_apply_tail(const ConvK& p, void* lds, const Regions& for_each_region) {
  const FusedApply& f = p.fa;
  float* s_par = reinterpret_cast<float*>(lds);
  int* s_flag = reinterpret_cast<int*>(s_par + 3 * p.C_out);
  const unsigned short* y = reinterpret_cast<const unsigned short*>(p.y);
  const int ldy = p.ldy, M = p.M, C = p.C_out;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // this wave's stores and atomics are acknowledged
  __syncthreads();
  u32x4 v0[kFaChunk];
  FaLane L0;
  L0.active = false; L0.c = 0; L0.rpp = 1; L0.mfirst = 0; L0.mend = 0;
  int idx = 0;
  for_each_region([&](const int m0, const int nrows, const int c0, const int nch) {
    if (idx++ == 0) { L0 = fa_lane<NT>(M, C, m0, nrows, c0, nch); fa_load_chunk(L0, y, ldy, L0.mfirst, v0); }
  });
  const bool timed_out = fused_grid_meet<NT>(f, s_flag);
  fused_stats_to_lds<NT>(f, p.stats, p.stats_mod, p.stats_ld, C, s_par, timed_out);
  __syncthreads();
  if (f.fault & kProbeNoSweep) return;
  idx = 0;
  for_each_region([&](const int m0, const int nrows, const int c0, const int nch) {
    if (idx++ == 0) {
      fa_finish_chunk(f, L0, C, s_par, L0.mfirst, v0);
      fused_apply_region<NT>(f, y, ldy, M, C, s_par, m0, nrows, c0, nch, 1);
    } else {
      fused_apply_region<NT>(f, y, ldy, M, C, s_par, m0, nrows, c0, nch);
    }
  });
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

// Pointers that come out of the LDS copy of the table are of unknown address space to the compiler: a dereference would be a
// FLAT access, which counts on the LDS counter as well -- every LDS read behind it then waits for global memory (the first
// version of this tail took eight serial round trips for a lane's 24 parameters: 13 us per phase).  So: every field a lane needs
// is read from LDS FIRST, the pointers are cast to the global address space, and the parameters come as 16-byte vectors.
typedef const __attribute__((address_space(1))) u32x4* g_u32x4_cptr;
typedef __attribute__((address_space(1))) u32x4* g_u32x4_ptr;
typedef const __attribute__((address_space(1))) f32x4u* g_f32x4_cptr;
typedef __attribute__((address_space(1))) float* g_f32_ptr;
template <class T> __device__ __forceinline__ unsigned long long gaddr(T* p) { return reinterpret_cast<unsigned long long>(p); }

struct FbLane {
  float mu[8], rs[8], be[8];
  unsigned long long y, dy, mean, rstd, beta;            // global addresses AT THE LANE'S FIRST CHANNEL
  int ldy, lddy, relu;
  int c, rpp, mfirst, mend;
  bool active;
};
constexpr int kFbChunk = 4;                              // (da, y) row pairs of a lane in flight together

// the lane's share of a region (rows [m0, m0 + nrows) x channels [c0, c0 + nch) of da = [M][ldg]) -- geometry and addresses (LDS reads only)
template <int NT>
__device__ __forceinline__ void fb_lane_geom(const FusedBwd& f, const int M, const int C, const int m0, const int nrows, const int c0, const int nch, FbLane& L) {
  const int oct = nch >> 3;
  L.rpp = NT / oct;
  const int vc = threadIdx.x % oct, rr = threadIdx.x / oct;
  L.c = c0 + (vc << 3);
  L.active = rr < L.rpp && L.c < C;
  L.mend = (m0 + nrows < M) ? m0 + nrows : M;
  L.mfirst = m0 + rr;
  const int cc = L.c < C ? L.c : 0;
  const int sg = fb_seg(f, cc), cr = cc - f.cb[sg];
  L.relu = f.relu[sg];
  L.ldy = f.ldy[sg]; L.lddy = f.lddy[sg];
  L.y = gaddr(f.y[sg] + cr); L.dy = gaddr(f.dy[sg] + cr);
  L.mean = gaddr(f.mean[sg] + cr); L.rstd = gaddr(f.rstd[sg] + cr);
  L.beta = L.relu ? gaddr(f.beta[sg] + cr) : L.mean;     // (no relu: beta may be NULL -- a valid address, values unused)
}
// ... and the per-channel parameters: six 16-byte loads (issued BEHIND the lane's first data loads: all in flight together)
__device__ __forceinline__ void fb_lane_params(FbLane& L) {
  const f32x4u a0 = *(g_f32x4_cptr)L.mean, a1 = *((g_f32x4_cptr)L.mean + 1);
  const f32x4u b0 = *(g_f32x4_cptr)L.rstd, b1 = *((g_f32x4_cptr)L.rstd + 1);
  const f32x4u c0 = *(g_f32x4_cptr)L.beta, c1 = *((g_f32x4_cptr)L.beta + 1);
  L.mu[0] = a0.x; L.mu[1] = a0.y; L.mu[2] = a0.z; L.mu[3] = a0.w; L.mu[4] = a1.x; L.mu[5] = a1.y; L.mu[6] = a1.z; L.mu[7] = a1.w;
  L.rs[0] = b0.x; L.rs[1] = b0.y; L.rs[2] = b0.z; L.rs[3] = b0.w; L.rs[4] = b1.x; L.rs[5] = b1.y; L.rs[6] = b1.z; L.rs[7] = b1.w;
  const float k = L.relu ? 1.f : 0.f;                     // (x * 0 for the unused values: they are finite means)
  L.be[0] = c0.x * k; L.be[1] = c0.y * k; L.be[2] = c0.z * k; L.be[3] = c0.w * k; L.be[4] = c1.x * k; L.be[5] = c1.y * k; L.be[6] = c1.z * k; L.be[7] = c1.w * k;
}
__device__ __forceinline__ void fb_load_chunk(const FbLane& L, const unsigned short* da, const int ldg, const int mbase,
                                              u32x4 (&vg)[kFbChunk], u32x4 (&vy)[kFbChunk]) {
  const unsigned long long ga = gaddr(da) + 2ull * (unsigned)L.c;
#pragma unroll
  for (int i = 0; i < kFbChunk; ++i) {
    const int m = mbase + i * L.rpp;
    const bool ok = L.active && m < L.mend;
    const unsigned long long mm = ok ? (unsigned long long)m : 0ull;       // (row 0 is always valid: no branch around the loads)
    const u32x4 g = *(g_u32x4_cptr)(ga + 2ull * mm * (unsigned)ldg);
    const u32x4 y = *(g_u32x4_cptr)(L.y + 2ull * mm * (unsigned)L.ldy);
    vg[i] = ok ? g : u32x4{0u, 0u, 0u, 0u};                                 // (a zero gradient adds nothing)
    vy[i] = y;
  }
}
__device__ __forceinline__ void fb_sums_chunk(const FbLane& L, const u32x4 (&vg)[kFbChunk], const u32x4 (&vy)[kFbChunk], float (&s1)[8], float (&s2)[8]) {
#pragma unroll
  for (int i = 0; i < kFbChunk; ++i) {
    const unsigned wg[4] = {vg[i].x, vg[i].y, vg[i].z, vg[i].w}, wy[4] = {vy[i].x, vy[i].y, vy[i].z, vy[i].w};
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
__device__ __forceinline__ void fb_apply_chunk(const FusedBwd& f, const FbLane& L, const int C, const float* s_tot, const int mbase,
                                               const u32x4 (&vg)[kFbChunk], const u32x4 (&vy)[kFbChunk]) {
  if (!L.active) return;
  float m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m1[j] = s_tot[L.c + j]; m2[j] = s_tot[C + L.c + j]; }
#pragma unroll
  for (int i = 0; i < kFbChunk; ++i) {
    const int m = mbase + i * L.rpp;
    if (m < L.mend) {
      const unsigned wg[4] = {vg[i].x, vg[i].y, vg[i].z, vg[i].w}, wy[4] = {vy[i].x, vy[i].y, vy[i].z, vy[i].w};
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = (j & 1) ? bf_hi(wg[j >> 1]) : bf_lo(wg[j >> 1]);
        const float yy = (j & 1) ? bf_hi(wy[j >> 1]) : bf_lo(wy[j >> 1]);
        const float xh = (yy - L.mu[j]) * L.rs[j];
        const float gj = (!L.relu || xh + L.be[j] > 0.f) ? g : 0.f;
        o[j] = L.rs[j] * (gj - m1[j] - xh * m2[j]);
      }
      *(g_u32x4_ptr)(L.dy + 2ull * (unsigned long long)m * (unsigned)L.lddy) = u32x4{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
  }
}

// a lane's sums of one region -> LDS partials [row lane][octet][16] -> column sums in two stages (every thread takes a column
// and a slice of the row lanes) -> atomics.  s_red: NT * 16 floats, s_red2: NT floats.
template <int NT>
__device__ __forceinline__ void fb_reduce_region(const FusedBwd& f, const FbLane& L, const int C, const int c0, const int nch,
                                                 const float (&s1)[8], const float (&s2)[8], float* s_red, float* s_red2) {
  const int oct = nch >> 3, rpp = L.rpp, ncol = 16 * oct;
  const int vc = threadIdx.x % oct, rr = threadIdx.x / oct;
  if (rr < rpp) {
    float* o = s_red + ((size_t)rr * oct + vc) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = s1[j]; o[8 + j] = s2[j]; }
  }
  __syncthreads();
  const int nsl = NT / ncol > 0 ? NT / ncol : 1;               // (ncol <= NT: nch <= 8 NT / 16)
  const int rps = (rpp + nsl - 1) / nsl;
  for (int e = threadIdx.x; e < nsl * ncol; e += NT) {
    const int col = e % ncol, sl = e / ncol;
    float t = 0.f;
    const int r1 = min(rpp, (sl + 1) * rps);
    for (int r = sl * rps; r < r1; ++r) t += s_red[(size_t)r * ncol + col];
    s_red2[e] = t;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ncol; e += NT) {
    float t = 0.f;
    for (int sl = 0; sl < nsl; ++sl) t += s_red2[sl * ncol + e];
    const int ch = c0 + ((e >> 4) << 3) + (e & 7);
    if (ch < C) {
      const int sg = fb_seg(f, ch);
      float* ap = f.acc[sg] + ((size_t)(blockIdx.x & (kFbSlots - 1)) * 2 + ((e >> 3) & 1)) * f.acc_ld[sg] + (ch - f.cb[sg]);
      __builtin_amdgcn_global_atomic_fadd_f32((g_f32_ptr)gaddr(ap), t);      // (hardware float add at the memory side, fire and forget)
    }
  }
  // (s_red is rewritten only behind the next region's first barrier; s_red2 behind its second)
}

// Totals of all C channels of the launch -> s_tot[2][C] (m1 = mean g, m2 = mean g xhat); workgroup 0 adds sum g to dbeta.
// Segment by segment (uniform): the sixteen device-scope loads of a channel (8 accumulator copies x 2 sums) in flight together.
template <int NT>
__device__ __forceinline__ void fused_bwd_totals_to_lds(const FusedBwd& f, const int C, float* s_tot, const bool timed_out) {
  const float poison = timed_out ? __builtin_nanf("") : 0.f;
  const int n = f.n;
  for (int sg = 0; sg < n; ++sg) {
    const int c_lo = f.cb[sg], c_hi = (sg + 1 < n) ? f.cb[sg + 1] : C;
    const int ld = f.acc_ld[sg];
    const unsigned long long ap = reinterpret_cast<unsigned long long>(f.acc[sg]);
    const float* acc = reinterpret_cast<const float*>(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(ap >> 32)) << 32) |
                                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(ap & 0xffffffffu)));
    const __amdgpu_buffer_rsrc_t ar = make_rsrc(acc, (f.fault & kProbeNoTotals) ? 0u : (unsigned)(kFbSlots * 2 * ld * 4));
    float* db = f.dbeta[sg];
    for (int ch = c_lo + (int)threadIdx.x; ch < c_hi; ch += NT) {
      const int cr = ch - c_lo;
      float v1[kFbSlots], v2[kFbSlots];
#pragma unroll
      for (int sl = 0; sl < kFbSlots; ++sl) { v1[sl] = ld4_agent(ar, (unsigned)(((sl * 2) * ld + cr) * 4)); v2[sl] = ld4_agent(ar, (unsigned)(((sl * 2 + 1) * ld + cr) * 4)); }
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int sl = 0; sl < kFbSlots; ++sl) { t1 += v1[sl]; t2 += v2[sl]; }
      t1 += poison;
      s_tot[ch] = t1 * f.inv_M;
      s_tot[C + ch] = t2 * f.inv_M;
      if (blockIdx.x == 0 && db) { g_f32_ptr dbp = (g_f32_ptr)gaddr(db + cr); *dbp = *dbp + t1; }
    }
  }
}

// The meeting point of the backward tail: the workgroup's atomics are acknowledged, then the grid barrier.
template <int NT>
__device__ __forceinline__ bool fused_bwd_meet(FbShared* sh) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    sh->flag = (sh->f.fault & kProbeNoWait) ? 0 : (grid_barrier_arrive_wait(sh->f.bar, gridDim.x, blockIdx.x, sh->f.spin_limit, sh->f.fault & 1, sh->f.step_poison) ? 1 : 0);
  __syncthreads();
  return sh->flag != 0;
}

// The whole tail for a workgroup whose stored regions `for_each_region(fn)` enumerates (fn(m0, nrows, c0, nch)).  The FIRST
// region's first chunk of (da, y) pairs stays in registers across the barrier (a workgroup with one tile -- most launches at
// BATCH_SIZE 64 -- reads da and y once).  ONE_CHUNK: a lane's share of any region is at most kFbChunk rows (the igemm5 tiles:
// 128 registers per lane there, no room for a second chunk in flight).  lds: NT * 68 + sizeof(FbShared) bytes.
template <int NT, bool ONE_CHUNK, class Regions>
__device__ __forceinline__ void fused_bwd_tail(const ConvK& p, void* lds, const Regions& for_each_region) {
  float* s_red = reinterpret_cast<float*>(lds);
  float* s_red2 = s_red + NT * 16;
  FbShared* sh = fused_bwd_stage(s_red2 + NT);
  const FusedBwd& f = sh->f;
  const unsigned short* da = reinterpret_cast<const unsigned short*>(p.y);
  const int ldg = p.ldy, M = p.M, C = p.C_out;
  u32x4 g0[kFbChunk], y0[kFbChunk];
  int idx = 0;
  for_each_region([&](const int m0, const int nrows, const int c0, const int nch) {
    FbLane L;
    fb_lane_geom<NT>(f, M, C, m0, nrows, c0, nch, L);
    u32x4 vg[kFbChunk], vy[kFbChunk];
    fb_load_chunk(L, da, ldg, L.mfirst, vg, vy);
    fb_lane_params(L);
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (!(f.fault & kProbeNoSums)) {
      fb_sums_chunk(L, vg, vy, s1, s2);
      if constexpr (!ONE_CHUNK) {
        for (int mb = L.mfirst + kFbChunk * L.rpp; mb < L.mend; mb += kFbChunk * L.rpp) {
          u32x4 wg[kFbChunk], wy[kFbChunk];
          fb_load_chunk(L, da, ldg, mb, wg, wy);
          fb_sums_chunk(L, wg, wy, s1, s2);
        }
      }
    }
    if (idx++ == 0) {
#pragma unroll
      for (int i = 0; i < kFbChunk; ++i) { g0[i] = vg[i]; y0[i] = vy[i]; }
    }
    fb_reduce_region<NT>(f, L, C, c0, nch, s1, s2, s_red, s_red2);
  });
  const bool timed_out = fused_bwd_meet<NT>(sh);
  fused_bwd_totals_to_lds<NT>(f, C, s_red, timed_out);
  __syncthreads();
  if (f.fault & kProbeNoSweep) return;
  idx = 0;
  for_each_region([&](const int m0, const int nrows, const int c0, const int nch) {
    FbLane L;
    fb_lane_geom<NT>(f, M, C, m0, nrows, c0, nch, L);
    fb_lane_params(L);
    if (idx++ == 0) {
      fb_apply_chunk(f, L, C, s_red, L.mfirst, g0, y0);
    } else {
      u32x4 vg[kFbChunk], vy[kFbChunk];
      fb_load_chunk(L, da, ldg, L.mfirst, vg, vy);
      fb_apply_chunk(f, L, C, s_red, L.mfirst, vg, vy);
    }
    if constexpr (!ONE_CHUNK) {
      for (int mb = L.mfirst + kFbChunk * L.rpp; mb < L.mend; mb += kFbChunk * L.rpp) {
        u32x4 wg[kFbChunk], wy[kFbChunk];
        fb_load_chunk(L, da, ldg, mb, wg, wy);
        fb_apply_chunk(f, L, C, s_red, mb, wg, wy);
      }
    }
  });
}

}  // namespace
