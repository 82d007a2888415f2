// ONE-SHOT GRID BARRIER shared by the launches that keep a grid-wide dependency inside the kernel (round 6): the convolution +
// batch-norm-apply launch (fused_bn.h) and the data-gradient + batch-norm-backward launch.  The protocol is the one
// bn_bwd_onepass_kernel (nnops.hip) has run since round 2 and the saturation stress test covers: everything that is polled or
// counted sits on its own 128-byte line and is shared by few workgroups (a device-scope atomic or load is served at the memory
// side at ~24 ns per access and line: tools/atomic_bench.hip) -- arrivals go to kGbSub counters, the last arrival of each to
// the top counter, the last of those sets kGbRel release words; workgroup b polls release word b % kGbRel.
//
// Conditions (the caller's): the control block is ZERO at launch and used by ONE launch per clearing; the grid is never larger
// than the number of workgroups the chip holds at once (one per CU here), so every workgroup is resident and the barrier cannot
// deadlock -- the spin is bounded all the same: a workgroup that gives up returns true, and its caller poisons what it writes
// and raises the step control word, so that the optimiser skips the step on every rank (include/mbx.h, STEP CONTROL BLOCK).
//
// No __threadfence(): a release fence writes back the whole L2 of the XCD (~45 us per launch, measured).  What one workgroup
// publishes for ALL the others across the barrier must therefore be device-scope atomics (performed at the memory side;
// `s_waitcnt vmcnt(0)` on their acknowledgement orders them before the arrival) read back by device-scope loads.  Ordinary
// stores are only ever re-read by the workgroup that wrote them (same CU, same L2: coherent without any fence).
#pragma once
#include <hip/hip_runtime.h>

#ifndef MBX_GB_SLEEP
#define MBX_GB_SLEEP 8
#endif
constexpr int kGbSub = 16, kGbRel = 32, kGbLine = 32;            // arrival counters / release words / 4-byte words per line
constexpr int kGbCtlWords = kGbLine * (2 + kGbSub + kGbRel);     // line 0: {grid size, timeout flag}; 6.4 KB per barrier

__device__ __forceinline__ unsigned gb_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long gb_ld(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float gb_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by ONE thread of every workgroup, behind `s_waitcnt vmcnt(0)` + a workgroup barrier.  b: this workgroup's index,
// G: workgroups in the grid.  fault bit 0 (tests only): workgroup 0 never arrives, everyone else times out.
// Returns true when the workgroup gave up (flag ctl[1] set; step_poison, if given, incremented).
__device__ __forceinline__ bool grid_barrier_arrive_wait(unsigned* ctl, const unsigned G, const unsigned b, const unsigned spin_limit,
                                                         const int fault, float* step_poison) {
  const unsigned sub = b % kGbSub, n_sub = (G - sub + kGbSub - 1) / kGbSub, n_top = G < (unsigned)kGbSub ? G : (unsigned)kGbSub;
  bool timed_out = false;
  bool last = ((fault & 1) && b == 0) ? false : atomicAdd(ctl + kGbLine * (1 + sub), 1u) == n_sub - 1;
  if (last) last = atomicAdd(ctl + kGbLine * (1 + kGbSub), 1u) == n_top - 1;
  if (last) {
    for (int r = 0; r < kGbRel; ++r)
      __hip_atomic_store(ctl + kGbLine * (2 + kGbSub + r), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const unsigned* rel = ctl + kGbLine * (2 + kGbSub + b % kGbRel);
    unsigned spins = 0;
    while (gb_ld(rel) == 0u) {
      __builtin_amdgcn_s_sleep(MBX_GB_SLEEP);
      if (++spins > spin_limit) {
        ctl[1] = 1u;
        timed_out = true;
        if (step_poison) atomicAdd(step_poison, 1.0f);
        break;
      }
    }
  }
  if (b == 0) ctl[0] = G;
  return timed_out;
}

// The batch-norm APPLY of the layer a training-mode convolution just wrote, run as the tail of that launch (fused_bn.h).
struct FusedApply {
  unsigned* bar;                 // grid-barrier control block (kGbCtlWords, ZERO at launch); NULL: not fused
  unsigned spin_limit; int fault;
  float* step_poison;            // word [0] of the step control block, or NULL
  unsigned short* a; int ld_a;   // activation rows: a + m * ld_a + c  (c = channel of this launch)
  const float* beta;
  float* mean; float* rstd; float* mmean; float* mvar; float* thr;    // published by workgroup 0 (as bn_apply_rows_kernel)
  int relu; float eps, decay; double inv_count;
};


// The batch-norm BACKWARD of the layers whose activation gradient a data-gradient launch writes, run as the tail of that launch
// (fused_bn.h).  Output channels [cb[i], cb[i + 1]) of the launch belong to segment i: a batch-norm layer (or a channel range of
// one) with its pre-BN output y, its gradient dy, statistics and accumulators, all pointers AT THE SEGMENT'S FIRST CHANNEL.
constexpr int kFbSlots = 8;                // accumulator copies (a workgroup adds into copy blockIdx % kFbSlots)
struct FusedBwd {
  unsigned* bar;                           // grid-barrier control block (ZERO at launch); NULL: not fused
  unsigned spin_limit; int fault;
  float* step_poison;
  int n;
  int cb[4];                               // ascending from 0, multiples of 8; unused entries 1 << 30
  const unsigned short* y[4]; int ldy[4];
  unsigned short* dy[4]; int lddy[4];
  const float* mean[4]; const float* rstd[4]; const float* beta[4];
  float* dbeta[4];                         // += sum g (workgroup 0)
  float* acc[4]; int acc_ld[4];            // [kFbSlots][2][acc_ld] float sums {sum g, sum g xhat}, ZERO at launch
  int relu[4];
  float inv_M;
};
