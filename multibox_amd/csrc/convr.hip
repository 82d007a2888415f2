// libmbx: conv_resident_kernel -- stride-1 multi-tap convolutions on SMALL maps with MANY channels: block17's 1x7 / 7x1
// layers (model.py:33-37: 17 x 17 maps, 128 -> 160 -> 192 channels; forward and data gradient, 80 launches of the
// training step) and block8's 1x3 / 3x1 (model.py:53-57).
//
// As implicit GEMMs these launches gather every pixel row once PER TAP through the L2 -> LDS path (a 256 x 64 tile of
// the 1x7 layer pulls 560 KB for 29 MFLOP) and run at 0.14-0.18 of the MFMA peak with no HBM traffic to blame: paced by
// the L2 -> LDS feed (profiles/r03_conv_l2_counters.txt).  Here a tile is a WHOLE IMAGE x a quarter of the output
// channels: the image (17 x 17 x C_in: 74-111 KB) is staged ONCE and stays in LDS, the filter slice of the tile is
// streamed one tap at a time through a ring of D slots by four loader waves, and the four multiplying waves read tap
// (r, s) of output pixel p at the LDS row of input pixel p + (r - pad_t) W + (s - pad_l) -- or at a row of zeros where
// that pixel lies outside the image (the per-lane row indices of all taps are computed once, while the image lands).
// L2 -> LDS bytes per tile: 160-190 KB for 33-49 MFLOP.
//
// Same arithmetic as the implicit GEMM: K order (r, s, c), one v_mfma_f32_16x16x32_bf16 per 32 input channels of a tap,
// lane group fch holding channels 8 (4 j + fch) .. + 8, filter rows as the A operand -- the accumulation sequence of every
// output element is the igemm kernels', so the results are bit-identical (tests/test_gpu_conv.py::test_conv_resident).
// Epilogues: bf16 store (EV = 0), store + batch-norm statistics (EV = 1: statistics row = the image), affine + relu (EV = 3).
#include "conv_common.h"
#ifndef MBX_RES_EALL
#define MBX_RES_EALL 0
#endif
#ifndef MBX_RES_PROBE
#define MBX_RES_PROBE 0
#endif

namespace {

constexpr int kRThreads = 512;                            // waves 0-3 multiply, waves 4-7 load; all eight stage the image
constexpr int kRCW = 4, kRLW = 4;

struct ResK {
  int N, H, W, HW;                                        // images; map (input = output: stride 1, "same" geometry)
  int NG, CPT;                                            // channel groups per image, output channels per group (a multiple of 8)
  int ntiles;                                             // N * NG; tile t = image t / NG, group t % NG
  int per;                                                // consecutive tiles per workgroup: the groups of ONE image where N * NG > CUs (its image staged once)
  int npi;                                                // 1 KB pieces of the image (+ its row of zeros)
  int dh[9], dw[9], dd[9];                                // tap (r, s) reads input pixel (oh + dh, ow + dw) = pixel index + dd: r - pad_t, s - pad_l, dh W + dw
};

// Conflict-free ds_read_b128 of 16 consecutive rows (pixels or filter rows) of C8 16-byte chunks: a read is served in
// groups of 16 lanes = 8 rows x 2 adjacent chunks (4 j + fch, fch in {0, 1} or {2, 3}), which must meet 16 distinct
// 16-byte bank groups.  Row r starts at bank group (r C8) mod 16; the chunk is XOR-ed with an EVEN key (the two chunks of
// a row stay apart in bit 0) that spreads the 8 rows over the bank groups their starts leave free:
//   C8 % 16 == 0 (rows start at group 0):           key = 2 (r & 7)
//   C8 % 16 == 8 (groups 0 / 8 by row parity):      key = 2 ((r >> 1) & 3)
//   C8 % 16 == 4, 12 (four starts by r & 3):        key = 2 ((r >> 2) & 1)
// (any 8 of 16 consecutive rows taken as {0-3, 12-15} or {4-11} + base are distinct under each key: the fragment rows of a
// tap shifted by any number of pixels stay conflict-free)
template <int C8>
__device__ __forceinline__ int rkey(int r) {
  return (C8 % 16 == 0) ? 2 * (r & 7) : (C8 % 16 == 8) ? 2 * ((r >> 1) & 3) : 2 * ((r >> 2) & 1);
}
// K step j of a row whose chunk-0 slot index is `row C8`: slot = row C8 + ((4 j + fch) ^ key).  With E = row C8 +
// (fch ^ (key & 3)) + 4 (key >> 2) this is (E ^ 4 (j & JX)) + 4 (j & ~JX): the key's upper bits flip bits of j, which
// (row C8 leaves those bits clear) is one XOR on E with a constant; the rest of j is a compile-time offset.
template <int C8> constexpr int rjx() { return (C8 % 16 == 0) ? 3 : (C8 % 16 == 8) ? 1 : 0; }
template <int C8>
__device__ __forceinline__ int renc(int row, int fch) {
  const int k = rkey<C8>(row);
  return row * C8 + (fch ^ (k & 3)) + 4 * (k >> 2);
}
template <int C8>
__device__ __forceinline__ int rslot(int E, int j) { return (E ^ (4 * (j & rjx<C8>()))) + 4 * (j & ~rjx<C8>()); }
// ... the same in BYTES off the start of LDS, the region's offset `base` folded in (a multiple of 256: the XOR-ed bits stay clear)
template <int C8>
__device__ __forceinline__ int rencb(int row, int fch, int base) { return base + 16 * renc<C8>(row, fch); }
// (Eb is an LDS ADDRESS -- the 32-bit value of an address_space(3) pointer -- so that the read is one ds_read_b128 off a register
// the XOR produced, with the rest of j in the instruction's offset field; through a generic pointer the compiler adds the
// aperture base back in with an instruction per read)
typedef const __attribute__((address_space(3))) u32x4* lds_u32x4_cptr;
__device__ __forceinline__ int lds_addr(const void* p) { return (int)(unsigned)(size_t)((const __attribute__((address_space(3))) char*)p); }
template <int C8>
__device__ __forceinline__ bf16x8 rread(int Eb, int j) {
  const unsigned a = (unsigned)((Eb ^ (64 * (j & rjx<C8>()))) + 64 * (j & ~rjx<C8>()));
  return __builtin_bit_cast(bf16x8, *(lds_u32x4_cptr)(size_t)a);
}

// s_waitcnt vmcnt(k PT) for a k the unrolled caller folds to a constant
template <int PT>
__device__ __forceinline__ void wait_vmcnt_k(const int k) {
  switch (k) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<PT>(); break;
    case 2: wait_vmcnt<2 * PT>(); break;
    case 3: wait_vmcnt<3 * PT>(); break;
    case 4: wait_vmcnt<4 * PT>(); break;
    case 5: wait_vmcnt<5 * PT>(); break;
    case 6: wait_vmcnt<6 * PT>(); break;
    case 7: wait_vmcnt<(7 * PT < 63 ? 7 * PT : 0)>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

// LDS row 16 a + f of a filter slot holds output channel rperm(a, f) of the tile: blocks 2A and 2A + 1 leave a lane EIGHT
// consecutive channels of one pixel (16-byte stores, conv_igemm3_kernel's order); a trailing unpaired block the plain order
template <int NB>
__device__ __forceinline__ int rperm(int a, int f) {
  return (a < (NB / 2) * 2) ? 32 * (a >> 1) + 8 * (f >> 2) + 4 * (a & 1) + (f & 3) : 16 * a + f;
}

// scheduling pattern of one K step: the NR LDS reads of the next step's fragments interleaved with the first MFMAs
template <int NR, int NM, int I = 0>
__device__ __forceinline__ void sched_interleave() {
  if constexpr (I < NR) {
    // (one read per MFMA up front: the rest of the step's MFMAs cover the latency of the last one)
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, I + 1 < NR ? 1 : NM - (NR - 1), 0);
    sched_interleave<NR, NM, I + 1>();
  }
}

template <int C8, int NB, int RS, int MI>
struct ResG {
  static constexpr int HWMAX = 16 * kRCW * MI;            // pixel slots of the multiplying waves
  static constexpr int KC = C8 / 4;                       // K steps (32 channels) per tap
  static constexpr int SLOT_CH = NB * 16 * C8;            // 16-byte chunks of a tap's filter slice
  static constexpr int PT = (SLOT_CH + 64 * kRLW - 1) / (64 * kRLW);   // pieces per loader wave and tap
  static constexpr int SLOTP = PT * kRLW * 64;            // ... the slot padded to whole rounds of the four loaders
  static constexpr int RED_CH = kRCW * NB * 16 * 2 * 4 / 16;          // statistics reduce: [4 waves][16 NB][2] floats
  static constexpr int img_ch(int hw) { return ((hw + 1) * C8 + 63) / 64 * 64; }   // the image + a row of zeros, in whole pieces
  static constexpr int lds_bytes(int hw, int d) { return (RED_CH + d * SLOTP + img_ch(hw)) * 16; }
  static constexpr int depth(int hw) {                    // ring depth: as many taps as 160 KB hold beside the image
    int d = RS;
    while (d > 2 && lds_bytes(hw, d) > 160 * 1024) --d;
    return d;
  }
};

template <int C8, int NB, int RS, int MI, int HWC, int EV>
__global__ void __launch_bounds__(kRThreads)
conv_resident_kernel(const ConvK p, const ResK q) {
  using G = ResG<C8, NB, RS, MI>;
  constexpr int KC = G::KC, PT = G::PT, SLOTP = G::SLOTP, D = G::depth(HWC), DD = D < RS ? D : RS;
  // taps per publication group: half the ring (the next group lands while this one is multiplied); everything resident: 4
  constexpr int GR = D >= RS ? (RS < 4 ? RS : 4) : (D / 2 > 0 ? D / 2 : 1), G0 = GR < RS ? GR : RS;
  static_assert(D >= RS || D >= 2 * GR || GR == 1, "two groups in the ring");
  constexpr int NP = NB / 2;                              // paired blocks
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  float* const red = reinterpret_cast<float*>(smem);      // [4 waves][16 NB][2]
  u32x4* const ring = smem + G::RED_CH;                   // [D][SLOTP]
  u32x4* const img = ring + D * SLOTP;                    // [HW + 1][C8] (swizzled); row HW = zeros; last: its final piece may overhang

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int first = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int frow = lane & 15, fch = lane >> 4;

  // the image: pieces wave, wave + 8, ... of 64 chunks each; chunk ci = pixel ci / C8, slot ci % C8 holds source chunk
  // slot ^ key(pixel) (the DMA destination is lane-linear: the swizzle is applied to the SOURCE chunk)
  auto issue_image = [&](const int im) {
    const int xo = im * p.x_img_stride;
    for (int i = wave; i < q.npi; i += 8) {
      const int ci = i * 64 + lane;
      const int pix = ci / C8, cs = ci - pix * C8;
      const int c = cs ^ rkey<C8>(pix);
      glds16(xr, img + i * 64, pix < q.HW ? (xo + pix * p.ldx + c * 8) * 2 : (int)kOOB);
    }
  };

  if (wave >= kRCW) {
    // ------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - kRCW;
    int wo[PT];                                           // this lane's filter chunks of a tap, before the tile's / tap's offset
    bool wv[PT];
    int rowch[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int ci = (lw + kRLW * i) * 64 + lane;
      const int row = ci / C8, cs = ci - row * C8;
      const int c = cs ^ rkey<C8>(row);
      wv[i] = row < 16 * NB;
      rowch[i] = wv[i] ? rperm<NB>(row >> 4, row & 15) : 0;                       // channel within the tile
      wo[i] = (rowch[i] * RS * (8 * C8) + c * 8) * 2;
    }
    // The taps of ALL tiles of this workgroup form ONE stream through the ring (tap T of the stream in slot T % D): behind the
    // barrier that frees a group's slots the loaders go on into the NEXT tile's filter slice, so that a tile after the first
    // finds its first group landed (BATCH_SIZE 256: the four channel groups of an image in one workgroup; by the stamps a tile
    // after the first waits 0.1-0.3 us at its P barrier, against 1.5-2 us with the ring restarted per tile).  `base` = slot of the tile's tap 0.
    int im_in = -1, base = 0;
    const int t0 = first * q.per, t1 = min(t0 + q.per, q.ntiles);
    for (int t = t0; t < t1; ++t) {
      const int im = t / q.NG;
      const bool last = t + 1 == t1;
      auto issue_tap = [&](const int rel) {               // rel: tap index off this tile's tap 0 (>= RS: the next tile's)
        const int tt = rel < RS ? t : t + 1, tap = rel < RS ? rel : rel - RS;
        const int g = tt - (tt / q.NG) * q.NG;
        const int cb = g * q.CPT, ce = min(cb + q.CPT, p.C_out);
        const int wbase = cb * RS * (8 * C8) * 2;
        const int sl = D == RS ? tap : (base + rel) % D;
        u32x4* dst = ring + sl * SLOTP + lw * 64;
#pragma unroll
        for (int i = 0; i < PT; ++i)
          glds16(wr, dst + i * (kRLW * 64), (wv[i] && cb + rowch[i] < ce) ? wbase + wo[i] + tap * (8 * C8) * 2 : (int)kOOB);
      };
      auto slot_of = [&](const int rel) { return D == RS ? rel % RS : (base + rel) % D; };
      // Taps are PUBLISHED IN GROUPS of GR (one barrier per group, not per tap: every barrier costs the multiplying waves a
      // drain of their LDS queue and a rendezvous with the loaders' wait + read-back chain -- 0.25 us each by the probes of
      // tools/res_stamps.py, seven of them a third of the 1x7 K loop).  The ring holds two groups: the next one is in flight
      // while the current one is multiplied, and a group's slots are refilled behind the barrier that ends it.
      if (t == t0) {
        issue_image(im);
#pragma unroll
        for (int tap = 0; tap < DD; ++tap) issue_tap(tap);
        wait_vmcnt_k<PT>(DD - G0);                        // the image and the first group have retired (this wave's share) ...
      } else if (im != im_in) {
        issue_image(im);                                  // (behind the D taps issued during the last tile: everything has to retire)
        wait_vmcnt<0>();
      } else {
        wait_vmcnt_k<PT>(DD - G0);                        // the D taps issued during the last tile, less the first group
      }
      im_in = im;
      lds_readback_wait(lds_readback_issue(ring + slot_of(G0 - 1) * SLOTP + (PT - 1) * (kRLW * 64) + lw * 64 + lane));   // ... and landed
      raw_barrier();                                      // P: image + group 0 published
#pragma unroll
      for (int g0 = 0; g0 < RS; g0 += GR) {
        const int e = g0 + GR < RS ? g0 + GR : RS;        // one past the group's last tap
        // taps issued so far, off this tile's tap 0: DD + g0 -- in the workgroup's last tile no further than RS
        if (e < RS) {
          const int e2 = e + GR < RS ? e + GR : RS;       // the next group [e, e2) has retired: behind it, the taps up to issued - 1
          if (last) wait_vmcnt_k<PT>((DD + g0 < RS ? DD + g0 : RS) - e2);
          else wait_vmcnt_k<PT>(DD + g0 - e2);
          lds_readback_wait(lds_readback_issue(ring + slot_of(e2 - 1) * SLOTP + (PT - 1) * (kRLW * 64) + lw * 64 + lane));
        }
        raw_barrier();                                    // B: the group's slots are free, the next group is published
#pragma unroll
        for (int k2 = g0; k2 < e; ++k2)
          if (DD + k2 < RS || !last) issue_tap(DD + k2);
      }
      if constexpr (EV == 1) raw_barrier();               // the multiplying waves' statistics reduce
      if constexpr (D != RS) base = (base + RS) % D;
    }
  } else {
  // ---------------------------------------------------------------------------------------------- multiplying waves
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
  // filter fragments: row 16 a + frow of a slot (byte offset within the slot + the ring's)
  constexpr int RING_OFF0 = G::RED_CH * 16, IMG_OFF0 = RING_OFF0 + D * SLOTP * 16;
  const int RING_OFF = RING_OFF0 + lds_addr(smem), IMG_OFF = IMG_OFF0 + lds_addr(smem);    // (dynamic LDS starts at 0 here: no static LDS)
  static_assert(RING_OFF0 % 256 == 0 && (SLOTP * 16) % 256 == 0, "the XOR-ed address bits of a row must be clear in the region offsets");
  int EA[NB];
#pragma unroll
  for (int a = 0; a < NB; ++a) EA[a] = rencb<C8>(16 * a + frow, fch, RING_OFF);
  // this lane's output pixels (fragment f of wave w = pixels 16 (w + 4 f) + frow) and the taps whose input pixel lies in the image
  // (bit `tap` of vm); the LDS row of every (tap, pixel block) is computed one tap AHEAD inside the K loop (nine VALU
  // operations in the shadow of the MFMAs; all taps up front were ~550 instructions between the image's issue and its first use)
  constexpr bool EALL = MBX_RES_EALL;                      // all taps' rows before the K loop (registers) or one tap ahead inside it
  int E[EALL ? RS : 2][MI];
  int opix[MI], vm[MI];
#pragma unroll
  for (int f = 0; f < MI; ++f) {
    const int pl = 16 * (wave + kRCW * f) + frow;
    const bool pv = pl < q.HW;
    opix[f] = pv ? pl : -1;
    const int oh = pv ? (int)fast_div((unsigned)pl, p.mg_w, p.sh_w) : -(1 << 20);          // (an invalid pixel: no tap is in range)
    const int ow = pv ? pl - oh * q.W : 0;
    int m = 0;
#pragma unroll
    for (int tap = 0; tap < RS; ++tap)       // (bitwise: with && the compiler builds a branch per term)
      m |= ((int)((unsigned)(oh + q.dh[tap]) < (unsigned)q.H) & (int)((unsigned)(ow + q.dw[tap]) < (unsigned)q.W)) << tap;
    vm[f] = m;
  }
  auto tap_rows = [&](const int tap, int (&e)[MI]) {
#pragma unroll
    for (int f = 0; f < MI; ++f) e[f] = rencb<C8>((vm[f] & (1 << tap)) ? opix[f] + q.dd[tap] : q.HW, fch, IMG_OFF);
  };
#ifdef MBX_I5_STAMPS
  const bool stamp = p.stamps && tid == 0 && blockIdx.x < 64;      // debug build (tools/res_stamps.py): tile phases of the first tile
#define MBXR_STAMP(i) do { if (stamp && t == first * q.per + p.dbg) p.stamps[(blockIdx.x * 8) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define MBXR_STAMP(i) do { } while (0)
#endif
  int im_in = -1, base = 0;                               // base: ring slot of the tile's tap 0 (the loaders' stream)
  for (int t = first * q.per, t1 = min(t + q.per, q.ntiles); t < t1; ++t) {
    const int im = t / q.NG, g = t - im * q.NG;
    const int cb = g * q.CPT, ce = min(cb + q.CPT, p.C_out);
    const bool newim = im != im_in;                       // (else: the image is in LDS already, and the last tile's stores may stay in flight)
    im_in = im;
    MBXR_STAMP(0);
    if (newim) issue_image(im);
    // per-channel scale / shift of the affine epilogue (folded batch norm, detect.py:313-326)
    float sc8[NP > 0 ? NP : 1][8], sh8[NP > 0 ? NP : 1][8], sc4[4], sh4[4];
    if constexpr (EV == 3) {
#pragma unroll
      for (int A = 0; A < NP; ++A)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = cb + 32 * A + 8 * fch + j;
          sc8[A][j] = (p.scale && c < ce) ? p.scale[c] : 1.f;
          sh8[A][j] = (p.shiftv && c < ce) ? p.shiftv[c] : 0.f;
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cb + 32 * NP + 4 * fch + j;
        sc4[j] = (p.scale && c < ce) ? p.scale[c] : 1.f;
        sh4[j] = (p.shiftv && c < ce) ? p.shiftv[c] : 0.f;
      }
    }
    if constexpr (EALL) {
#pragma unroll
      for (int tap = 0; tap < RS; ++tap) tap_rows(tap, E[tap]);
    } else {
      tap_rows(0, E[0]);
    }
    if (newim) {
      wait_vmcnt<0>();
      lds_readback_wait(lds_readback_issue(img + lane));
    }
    raw_barrier();                                        // P
    MBXR_STAMP(1);
    // (the rows are the same for every tile; opaque to the compiler here, or it hoists every K step's address out of the
    // tile loop -- a hundred registers of loop invariants, spilled)
#pragma unroll
    for (int a = 0; a < NB; ++a) asm volatile("" : "+v"(EA[a]));
#pragma unroll
    for (int f = 0; f < MI; ++f) asm volatile("" : "+v"(opix[f]), "+v"(vm[f]));

    f32x4 acc[NB][MI];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
      for (int f = 0; f < MI; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K steps s = tap KC + j, SOFTWARE-PIPELINED by hand: the fragments of step s + 1 are read (double-buffered registers)
    // while step s is multiplied, one LDS read per two MFMAs (sched_group_barrier), and nothing crosses a step's end
    // (sched_barrier: the compiler otherwise computes every step's addresses up front and spills them).  The ring's
    // barrier B_tap sits in the READ stream -- in front of the first read of tap + 1, i.e. one step ahead of the MFMAs --
    // so the last step of a tap is multiplied behind it and covers the latency of the next tap's first reads.
    constexpr int NS = RS * KC, NR = NB + MI, NM = NB * MI;
    int sofs[RS];                                         // byte offset of each tap's ring slot (compile-time where the ring holds all taps)
#pragma unroll
    for (int tap = 0; tap < RS; ++tap) sofs[tap] = (D == RS ? tap : (base + tap) % D) * (SLOTP * 16);
    if constexpr (D != RS) base = (base + RS) % D;
    bf16x8 wf[2][NB], pf[2][MI];
#pragma unroll
    for (int a = 0; a < NB; ++a) wf[0][a] = rread<C8>(EA[a] + sofs[0], 0);
#pragma unroll
    for (int f = 0; f < MI; ++f) pf[0][f] = rread<C8>(E[0][f], 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      const int tap1 = (s + 1) / KC, j1 = (s + 1) % KC;
      if (j1 == 0 && (tap1 % GR == 0 || tap1 == RS)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the group's slots (the last: of the image) are done
        raw_barrier();                                        // B: the group's slots may be refilled; the next group is published
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!EALL) { if (s % KC == 0 && s / KC + 1 < RS) tap_rows(s / KC + 1, E[(s / KC + 1) & 1]); }     // the next tap's rows (used KC - 1 steps on)
      // (timing probes, wrong results, debug builds only: -DMBX_RES_PROBE=1 no fragment reads, 2 no MFMAs, 3 neither --
      // compile-time, so that the straight-line schedule of the step stays what it is)
      constexpr bool rd_ = !(MBX_RES_PROBE & 1), mm_ = !(MBX_RES_PROBE & 2);
      if (s + 1 < NS && rd_) {
        const int so = sofs[tap1 < RS ? tap1 : 0];
#pragma unroll
        for (int a = 0; a < NB; ++a) wf[nxt][a] = rread<C8>(EA[a] + so, j1);
#pragma unroll
        for (int f = 0; f < MI; ++f) pf[nxt][f] = rread<C8>(E[EALL ? (tap1 < RS ? tap1 : 0) : (tap1 & 1)][f], j1);
      }
      // pixel-block-major: the fragment read LAST (pixel block MI - 1) is needed last in the next step
      if (mm_) {
#pragma unroll
      for (int f = 0; f < MI; ++f)
#pragma unroll
        for (int a = 0; a < NB; ++a)
          acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][a], pf[cur][f], acc[a][f], 0, 0, 0);
      }
      if (s + 1 < NS) sched_interleave<NR, NM>();
      __builtin_amdgcn_sched_barrier(0);
    }

    MBXR_STAMP(2);
    // ---- epilogue: blocks 2A, 2A + 1 -> 8 consecutive channels (cb + 32 A + 8 fch ..) of the lane's pixel: 16-byte stores;
    // an unpaired last block: 4 consecutive channels (cb + 32 NP + 4 fch ..), 8-byte stores
    float s1[NB][4], s2[NB][4];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[a][r] = 0.f; s2[a][r] = 0.f; }
    const int yo = im * p.y_img_stride;
#pragma unroll
    for (int f = 0; f < MI; ++f) {
      const bool pv = opix[f] >= 0;
      const int pix_off = yo + (pv ? opix[f] : 0) * p.ldy;
#pragma unroll
      for (int A = 0; A < NP; ++A) {
        const int c0 = cb + 32 * A + 8 * fch;
        const bool ok = pv && c0 < ce;                    // CPT % 8 == 0: a group of eight is all in or all out
        float v8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v8[r] = acc[2 * A][f][r]; v8[4 + r] = acc[2 * A + 1][f][r]; }
        if constexpr (EV == 3) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = v8[j] * sc8[A][j] + sh8[A][j];
          if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = relu_f(v8[j]);
          }
        }
        unsigned h2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h2[j] = pack2bf(v8[2 * j], v8[2 * j + 1]);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{h2[0], h2[1], h2[2], h2[3]}, yr, ok ? (int)((pix_off + c0) * 2) : (int)kOOB, 0, 0);
        if constexpr (EV == 1) {
          if (ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned w0 = h2[r >> 1], w1 = h2[2 + (r >> 1)];
              const float f0 = (r & 1) ? bf_hi(w0) : bf_lo(w0), f1 = (r & 1) ? bf_hi(w1) : bf_lo(w1);
              s1[2 * A][r] += f0; s2[2 * A][r] += f0 * f0;
              s1[2 * A + 1][r] += f1; s2[2 * A + 1][r] += f1 * f1;
            }
          }
        }
      }
      if constexpr (NB > 2 * NP) {
        constexpr int a = 2 * NP;
        const int c0 = cb + 16 * a + 4 * fch;
        const bool ok = pv && c0 < ce;
        float v4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[a][f][r];
          if constexpr (EV == 3) { v = v * sc4[r] + sh4[r]; if (p.relu) v = relu_f(v); }
          v4[r] = v;
        }
        const unsigned g2[2] = {pack2bf(v4[0], v4[1]), pack2bf(v4[2], v4[3])};
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{g2[0], g2[1]}, yr, ok ? (int)((pix_off + c0) * 2) : (int)kOOB, 0, 0);
        if constexpr (EV == 1) {
          if (ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float fv = (r & 1) ? bf_hi(g2[r >> 1]) : bf_lo(g2[r >> 1]);
              s1[a][r] += fv; s2[a][r] += fv * fv;
            }
          }
        }
      }
    }
    if constexpr (EV == 1) {
      // ONE statistics row per image: lane sums -> the 16 lanes that share its channels (DPP row sums, eight at a time) -> the
      // four waves, fixed order
      {
        float v[NB * 8];
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[a * 8 + r] = s1[a][r]; v[a * 8 + 4 + r] = s2[a][r]; }
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          float u[8] = {v[a * 8], v[a * 8 + 1], v[a * 8 + 2], v[a * 8 + 3], v[a * 8 + 4], v[a * 8 + 5], v[a * 8 + 6], v[a * 8 + 7]};
          row_sum16_x8(u);
          if (frow == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ch = rperm<NB>(a, 4 * fch + r);
              red[(wave * 16 * NB + ch) * 2] = u[r];
              red[(wave * 16 * NB + ch) * 2 + 1] = u[4 + r];
            }
          }
        }
      }
      lds_barrier();                                      // (the loaders attend)
      if (tid < 16 * NB && cb + tid < ce) {
        float x1 = 0.f, x2 = 0.f;
#pragma unroll
        for (int w = 0; w < kRCW; ++w) { x1 += red[(w * 16 * NB + tid) * 2]; x2 += red[(w * 16 * NB + tid) * 2 + 1]; }
        stats_write(p, im, cb + tid, x1, x2);
      }
    }
    MBXR_STAMP(3);
  }
#undef MBXR_STAMP
  }  // multiplying waves
  if constexpr (EV == 1) {
    // the layer's BN apply as the launch's tail (fused_bn.h): all eight waves; LDS is dead -- parameters where the ring was
    if (p.fa.bar != nullptr) {
      fused_apply_tail<kRThreads>(p, ring, [&](auto&& fn) {
        for (int t = first * q.per, t1 = min(t + q.per, q.ntiles); t < t1; ++t) {
          const int im = t / q.NG, g = t - im * q.NG;
          const int cb = g * q.CPT, ce = min(cb + q.CPT, p.C_out);
          fn(im * q.HW, q.HW, cb, ce - cb);
        }
      });
    }
  }
  if constexpr (EV == 0) {
    // data gradients: the BN backward of the layers whose activation gradient this launch wrote, as its tail (fused_bn.h)
    if (p.fb.bar != nullptr) {
      static_assert(kRThreads * 68 + sizeof(FbShared) <= (G::depth(HWC) * SLOTP + G::img_ch(HWC)) * 16, "the tail's reduce area fits ring + image");
      fused_bwd_tail<kRThreads, false>(p, ring, [&](auto&& fn) {
        for (int t = first * q.per, t1 = min(t + q.per, q.ntiles); t < t1; ++t) {
          const int im = t / q.NG, g = t - im * q.NG;
          const int cb = g * q.CPT, ce = min(cb + q.CPT, p.C_out);
          fn(im * q.HW, q.HW, cb, ce - cb);
        }
      });
    }
  }
}

// ------------------------------------------------------------------------------------------ pointwise, pixels resident
// conv_pwres_kernel: the 1x1 convolutions whose launch time is their EPILOGUE's memory traffic -- the residual "up" convolutions
// of block35 / block17 / block8 (model.py:19-23, 39-43, 59-63: 128 -> 320, 384 -> 1088, 448 -> 2080 channels; relu(skip +
// s (conv + b)) + sign bits) and the data gradients of the fused first 1x1s (96 -> 320, 320 -> 1088, 384 -> 2080; accumulate +
// relu mask): 2.36 ms of the round-4 training step at half the streaming rate of the chip.  As persistent 128 x 256 / 256 x
// 128 tiles every workgroup runs K loop and epilogue back to back, and all of them in step: the chip alternates between an
// operand-feed phase and an HBM phase.  Here a workgroup keeps an 80-PIXEL tile of the input resident in LDS (loaded once:
// 80 x K) and walks the OUTPUT CHANNELS in column tiles of 128: the filter -- 0.1-0.9 MB, L2-resident, a fifth of a
// microsecond away -- streams through a ring of 64-deep slots (four loader waves), each column tile is 120 MFMAs per wave and
// then 40 KB of epilogue traffic whose reads were issued before its K loop.  Per CU the HBM stream is fine-grained (nine
// small epilogues per launch instead of three large ones) and one tile per workgroup needs no persistence: 232 workgroups
// for block17, whatever else holds CUs.  Same K order and MFMA grouping as every implicit-GEMM tile, the shared epilogue
// (conv_common.h): bit-identical results.
struct PwK {
  int npt, NCS, nct, ctper, nunits;                       // pixel tiles, column splits, column tiles (128 channels), column tiles per split
  int npi;                                                // 1 KB pieces of the resident pixel tile
};
constexpr int kPwCol = 128;

// MIH: pixel blocks (of 16) per HALF of the resident pixel tile (5: 160 pixels, 4: 128 where K = 448 leaves no room for more)
template <int C8, int MIH>
struct PwG {
  static constexpr int PIX = 32 * MIH;                    // pixels of a tile
  static constexpr int KC = C8 / 4;                       // K steps of 32
  static constexpr int PIX_CH = (PIX * C8 + 63) / 64 * 64;
  // ring slots: 64-deep (16 KB) where at least three fit beside the pixel tile, else 32-deep (8 KB): what counts is the
  // bytes in flight against the DMA's latency, and a ring of two holds one slot in flight
  static constexpr int ROOM = 160 * 1024 / 16 - PIX_CH;
  static constexpr int KS = (C8 % 8 == 0 && ROOM / (kPwCol * 8) >= 3) ? 2 : 1;      // K steps per ring slot
  static constexpr int NSL = KC / KS;                     // slots per column tile
  static constexpr int C8S = 4 * KS;                      // 16-byte chunks per filter row of a slot
  static constexpr int SLOT_CH = kPwCol * C8S;            // chunks per slot
  static constexpr int PT = SLOT_CH / (64 * kRLW);        // pieces per loader wave and slot
  static constexpr int NST_RAW = ROOM / SLOT_CH;
  static constexpr int NST = NST_RAW > 8 ? 8 : NST_RAW;   // ring depth
  static constexpr int LDS_BYTES = (NST * SLOT_CH + PIX_CH) * 16;
  static_assert(KC % KS == 0 && NST >= 3 && (NST - 1) * PT < 64, "ring");
};

// conv_pwres_kernel, third form (round 5).  What the first two taught (tools/pw_stamps.py, LAB_NOTES): with an 80-pixel tile the K
// loop of a 128-channel column tile was paced by the FOUR LOADER WAVES (96 KB of filter = 24 LDS-DMA instructions each per
// 120 MFMAs of a multiplying wave: 2.1 us against 1.3), and a second multiplying group that wrote one tile out while the first
// multiplied the next made every K loop 3 us -- the SIMD's instruction ISSUE is shared, and the epilogue's ~300 vector
// instructions per tile and wave need the slots the MFMAs leave.  So: ONE group of four multiplying waves, a pixel tile twice
// as large (2 x MIH blocks: the filter bytes per MFMA halve, the loaders keep up), multiplied in two halves per K step so
// that the fragment registers stay those of the small tile; the epilogue's reads AND its per-channel shift row issued
// before the K loop.
template <int C8, int MIH, int EV>
__global__ void __launch_bounds__(kRThreads)
conv_pwres_kernel(const ConvK p, const PwK q) {
  using G = PwG<C8, MIH>;
  constexpr int KC = G::KC, KS = G::KS, NSL = G::NSL, C8S = G::C8S, SLOT_CH = G::SLOT_CH, PT = G::PT, NST = G::NST, PIX = G::PIX;
  constexpr int MI = 2 * MIH, NI = 2, NA = 1;
  // (EV 2: the relu mask comes from the SIGN BITS only -- the tensor-mask path is compiled out)
  constexpr int EVC = EV, BMODE = EV == 2 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  u32x4* const ring = smem;                               // [NST][128 rows][C8S]
  u32x4* const pix = smem + NST * SLOT_CH;                // [PIX pixels][C8] (swizzled)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int first = xcd_remap((int)blockIdx.x, (int)gridDim.x), G_ = (int)gridDim.x;
  const int frow = lane & 15, fch = lane >> 4;

  // the pixel tile: pieces wave, wave + 8, ...; chunk ci = row ci / C8 (pixel m0 + row), slot ci % C8 holds source chunk slot ^ key(row)
  auto issue_pixels = [&](const int m0) {
    for (int i = wave; i < q.npi; i += kRThreads / 64) {
      const int ci = i * 64 + lane;
      const int row = ci / C8, cs = ci - row * C8;
      const int c = cs ^ rkey<C8>(row);
      const int m = m0 + row;
      const bool ok = row < PIX && m < p.M;
      const int img = (int)fast_div((unsigned)(ok ? m : 0), p.mg_hw, p.sh_hw);
      glds16(xr, pix + i * 64, ok ? (img * p.x_img_stride + (m - img * p.HW_out) * p.ldx + c * 8) * 2 : (int)kOOB);
    }
  };

  if (wave >= kRCW) {
    // ------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - kRCW;
    int wo[PT], chn[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int ci = (lw + kRLW * i) * 64 + lane;
      const int row = ci / C8S, cs = ci - row * C8S;
      const int c = cs ^ rkey<C8S>(row);
      // LDS row 32 w + 16 a + f = channel 32 w + rperm<2>(a, f) of the column tile (wave w's 32 channels, igemm3's order)
      chn[i] = 32 * (row >> 5) + rperm<2>((row >> 4) & 1, row & 15);
      wo[i] = (chn[i] * (8 * C8) + c * 8) * 2;
    }
    for (int u = first; u < q.nunits; u += G_) {
      const int pt = u / q.NCS, cs_ = u - pt * q.NCS;
      const int ct0 = cs_ * q.ctper, ct1 = min(ct0 + q.ctper, q.nct);
      const int NS = (ct1 - ct0) * NSL;                   // slots of this unit
      int si = 0, si_ct = ct0, si_kc = 0, si_pos = 0;     // issue cursor: slot index, its column tile / K chunk, ring position
      auto issue_slot = [&]() {
        const int n0 = si_ct * kPwCol;
        u32x4* dst = ring + si_pos * SLOT_CH + lw * 64;
#pragma unroll
        for (int i = 0; i < PT; ++i)
          glds16(wr, dst + i * (kRLW * 64), (n0 + chn[i] < p.C_out) ? n0 * (8 * C8) * 2 + wo[i] + si_kc * (KS * 64) : (int)kOOB);
        ++si;
        if (++si_kc == NSL) { si_kc = 0; ++si_ct; }
        si_pos = si_pos == NST - 1 ? 0 : si_pos + 1;
      };
      issue_pixels(pt * PIX);
      const int ahead = NS < NST ? NS : NST;
      for (int i = 0; i < ahead; ++i) issue_slot();
      // the pixel tile and slot 0 have retired (this wave's share), then landed
      if (ahead == NST) wait_vmcnt<(NST - 1) * PT>(); else wait_vmcnt<0>();
      lds_readback_wait(lds_readback_issue(ring + (PT - 1) * (kRLW * 64) + lw * 64 + lane));
      raw_barrier();                                      // P
      int pos1 = 1 % NST;                                 // ring position of slot s + 1
      for (int s_ = 0; s_ < NS; ++s_) {
        if (s_ + 1 < NS) {
          // slot s + 1 has retired: behind it this wave has issued the slots up to min(NS - 1, s + NST - 1)
          const int newer = (s_ + NST - 1 < NS - 1 ? s_ + NST - 1 : NS - 1) - (s_ + 1);
          if (newer >= NST - 2) wait_vmcnt<(NST - 2) * PT>();
          else if (NST > 3 && newer == NST - 3) wait_vmcnt<(NST - 3 > 0 ? NST - 3 : 0) * PT>();
          else if (NST > 4 && newer == NST - 4) wait_vmcnt<(NST - 4 > 0 ? NST - 4 : 0) * PT>();
          else if (NST > 5 && newer == NST - 5) wait_vmcnt<(NST - 5 > 0 ? NST - 5 : 0) * PT>();
          else if (NST > 6 && newer == NST - 6) wait_vmcnt<(NST - 6 > 0 ? NST - 6 : 0) * PT>();
          else if (NST > 7 && newer == NST - 7) wait_vmcnt<(NST - 7 > 0 ? NST - 7 : 0) * PT>();
          else wait_vmcnt<0>();
          lds_readback_wait(lds_readback_issue(ring + pos1 * SLOT_CH + (PT - 1) * (kRLW * 64) + lw * 64 + lane));
        }
        raw_barrier();                                    // B_s: slot s is free, slot s + 1 is published
        if (si < NS) issue_slot();                        // slot s + NST into the ring position slot s had
        pos1 = pos1 == NST - 1 ? 0 : pos1 + 1;
      }
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------- multiplying waves
  const int lds0 = lds_addr(smem);
  // (pixel block f = rows 16 f + frow: every row key ignores multiples of 16, so its address is block 0's + 16 f rows -- one
  // register for all the blocks)
  int EA[NI], Ep0;
#pragma unroll
  for (int a = 0; a < NI; ++a) EA[a] = rencb<C8S>(32 * wave + 16 * a + frow, fch, lds0);
  Ep0 = rencb<C8>(frow, fch, lds0 + NST * SLOT_CH * 16);
#ifdef MBX_I5_STAMPS
  // debug build (tools/pw_stamps.py): [block][column tile (first 8)][K loop starts, K loop done, epilogue issued] of the first unit
  const bool stamp = p.stamps && tid == 0 && blockIdx.x < 64;
#define MBXP_STAMP(ci, i) do { if (stamp && u == first && (ci) < 8) p.stamps[(blockIdx.x * 8 + (ci)) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define MBXP_STAMP(ci, i) do { } while (0)
#endif
  for (int u = first; u < q.nunits; u += G_) {
    const int pt = u / q.NCS, cs_ = u - pt * q.NCS;
    const int ct0 = cs_ * q.ctper, ct1 = min(ct0 + q.ctper, q.nct);
    const int m0 = pt * PIX;
    issue_pixels(m0);
    wait_vmcnt<0>();
    lds_readback_wait(lds_readback_issue(pix + lane));
    raw_barrier();                                        // P
#pragma unroll
    for (int a = 0; a < NI; ++a) asm volatile("" : "+v"(EA[a]));       // (opaque per unit: no hoisting of every K step's address)
    asm volatile("" : "+v"(Ep0));
    int so = 0;                                           // byte offset of the ring position being read
    const int mlane = m0 + frow;
    for (int ct = ct0; ct < ct1; ++ct) {
      const int clane = ct * kPwCol + 32 * wave + 8 * fch;
      MBXP_STAMP(ct - ct0, 0);
#ifdef MBX_I5_STAMPS
      const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
      // the epilogue's reads (residual skip / accumulate source / sign bits) and its per-channel shift row, issued before the K loop:
      // they land while the column tile is multiplied (the multiplying waves issue no other vector-memory instruction)
      u32x4 la[MI][NA], lb[MI][NA];
#ifdef MBX_I5_STAMPS
      const bool epi_ = !(p.dbg & 16);                     // timing probe (wrong results): no epilogue memory traffic at all
#else
      constexpr bool epi_ = true;
#endif
      if (epi_) conv_epilogue_issue_reads<EVC, false, NA, MI, BMODE>(p, mlane, clane, 0, la, lb);
      float s1[NA][8], s2[NA][8], sc[NA][8], sh[NA][8];
      conv_epilogue_channels<EVC, NA>(p, clane, sc, sh, s1, s2);
      f32x4 acc[NI][MI];
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int f = 0; f < MI; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      // HALF steps h = 2 s + half (K step s, pixel half): half 0 multiplies pixel blocks 0 .. MIH - 1, half 1 the rest, against
      // the SAME filter fragments; software-pipelined like conv_resident_kernel's loop (the fragments of half step h + 1 are
      // read while h is multiplied; the slot barrier sits in the read stream)
      bf16x8 wf[2][NI], pf[2][MIH];
#pragma unroll
      for (int a = 0; a < NI; ++a) wf[0][a] = rread<C8S>(EA[a] + so, 0);
#pragma unroll
      for (int f = 0; f < MIH; ++f) pf[0][f] = rread<C8>(Ep0 + f * (16 * C8 * 16), 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2 * KC; ++h) {
        const int s = h >> 1, half = h & 1;
        const int h1 = h + 1, s1_ = h1 >> 1, half1 = h1 & 1;
        if (half1 == 0 && s1_ % KS == 0) {                  // the next half step opens a new slot (or ends the column tile)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done
          raw_barrier();                                        // B: the slot may be refilled; the next one is published
          so = so == (NST - 1) * SLOT_CH * 16 ? 0 : so + SLOT_CH * 16;
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef MBX_I5_STAMPS
        const bool rd_ = !(p.dbg & 4), mm_ = !(p.dbg & 8);       // timing probes (wrong results): no fragment reads / no MFMAs
#else
        constexpr bool rd_ = true, mm_ = true;
#endif
        if (h1 < 2 * KC && rd_) {
          if (half1 == 0) {
#pragma unroll
            for (int a = 0; a < NI; ++a) wf[s1_ & 1][a] = rread<C8S>(EA[a] + so, s1_ % KS);
          }
#pragma unroll
          for (int f = 0; f < MIH; ++f) pf[h1 & 1][f] = rread<C8>(Ep0 + (MIH * half1 + f) * (16 * C8 * 16), s1_);
        }
        if (mm_) {
#pragma unroll
        for (int f = 0; f < MIH; ++f)
#pragma unroll
          for (int a = 0; a < NI; ++a)
            acc[a][MIH * half + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s & 1][a], pf[h & 1][f], acc[a][MIH * half + f], 0, 0, 0);
        }
        if (h1 < 2 * KC) { if (half1 == 0) sched_interleave<NI + MIH, NI * MIH>(); else sched_interleave<MIH, NI * MIH>(); }
        __builtin_amdgcn_sched_barrier(0);
      }
      MBXP_STAMP(ct - ct0, 1);
#ifdef MBX_I5_STAMPS
      if (stamp && u == first && ct - ct0 < 8) p.stamps[(blockIdx.x * 8 + (ct - ct0)) * 4 + 3] = __builtin_amdgcn_s_memtime() - clk0;   // shader cycles of the K loop
#endif
      if (epi_) conv_epilogue_finish<EVC, false, NI, MI, MI, BMODE>(p, acc, mlane, clane, 0, la, lb, sh, sc, s1, s2);
      MBXP_STAMP(ct - ct0, 2);
    }
  }
#undef MBXP_STAMP
}

int resident_cus();

template <int C8, int MIH, int EV>
int launch_pwres(const ConvK& k, PwK& q, hipStream_t s) {
  using G = PwG<C8, MIH>;
  constexpr int lds = G::LDS_BYTES;
  static_assert(lds <= 160 * 1024, "LDS");
  q.npt = (k.M + G::PIX - 1) / G::PIX;
  q.nct = (k.C_out + kPwCol - 1) / kPwCol;
  const int ncu = resident_cus();
  int ncs = ncu / q.npt;                                  // column splits: as many units as fit one round of workgroups
  if (ncs < 1) ncs = 1;
  if (ncs > q.nct) ncs = q.nct;
  q.ctper = (q.nct + ncs - 1) / ncs;
  q.NCS = (q.nct + q.ctper - 1) / q.ctper;
  q.nunits = q.npt * q.NCS;
  q.npi = G::PIX_CH / 64;
  int grid = q.nunits < ncu ? q.nunits : ncu;
  if (k.max_wg > 0 && grid > k.max_wg) grid = k.max_wg;
  if (k.dry) return MBX_OK;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pwres_kernel<C8, MIH, EV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL((conv_pwres_kernel<C8, MIH, EV>), dim3(grid), dim3(kRThreads), lds, s, k, q);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

int resident_cus() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return ncu;
}

template <int C8, int NB, int RS, int MI, int HWC>
int launch_resident(const ConvK& k, const ResK& q, int grid, hipStream_t s) {
  using G = ResG<C8, NB, RS, MI>;
  constexpr int lds = G::lds_bytes(HWC, G::depth(HWC));
  static_assert(lds <= 160 * 1024, "LDS");
  static_assert((G::depth(HWC) - 1) * G::PT < 64, "vmcnt");
  static_assert(HWC <= G::HWMAX, "pixel slots");
  static bool attr[3] = {false, false, false};
  const int ev = k.epi == MBX_EPI_AFFINE ? 2 : k.stats ? 1 : 0;
  if (!attr[ev]) {
    if (ev == 2) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_resident_kernel<C8, NB, RS, MI, HWC, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if (ev) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_resident_kernel<C8, NB, RS, MI, HWC, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_resident_kernel<C8, NB, RS, MI, HWC, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr[ev] = true;
  }
  if (ev == 2) hipLaunchKernelGGL((conv_resident_kernel<C8, NB, RS, MI, HWC, 3>), dim3(grid), dim3(kRThreads), lds, s, k, q);
  else if (ev) hipLaunchKernelGGL((conv_resident_kernel<C8, NB, RS, MI, HWC, 1>), dim3(grid), dim3(kRThreads), lds, s, k, q);
  else hipLaunchKernelGGL((conv_resident_kernel<C8, NB, RS, MI, HWC, 0>), dim3(grid), dim3(kRThreads), lds, s, k, q);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

// channel groups per image and 16-channel blocks per group for C_out output channels: four groups (N = 64 images -> one
// tile per CU), each a multiple of 8 channels
inline void resident_groups(int C_out, int& NG, int& CPT, int& NB) {
  NG = 4;
  CPT = ((C_out + NG * 8 - 1) / (NG * 8)) * 8;
  NB = (CPT + 15) / 16;
}

}  // namespace

// statistics rows of the resident-image launch (a row per image)
int mbx_resident_rows(int N) { return N; }

// mbx_conv_desc.tile_config = 98: the RESIDENT-IMAGE launch (conv_resident_kernel).  MBX_ERR_UNSUPPORTED for anything but a
// stride-1 convolution with "same" geometry (forward, or the data gradient of one) of 7 taps on a map of 65 .. 289 pixels
// (17 x 17) with C_in 128 / 160 / 192, or of 3 taps on an 8 x 8 map with C_in 192 / 224 / 256; C_out a multiple of 8 and a bf16 store epilogue with or without statistics, or
// the affine (+ relu) epilogue of a folded batch norm.
int mbx_launch_resident(void* convk, int N, int H_out, hipStream_t s) {
  ConvK& k = *reinterpret_cast<ConvK*>(convk);
  if (k.mul != 1 || k.shift || (k.epi != MBX_EPI_STORE && k.epi != MBX_EPI_AFFINE) || k.accumulate || k.skip || k.bits || k.rscale != 0.f ||
      k.bw_n || (k.epi == MBX_EPI_AFFINE && k.stats))
    return MBX_ERR_UNSUPPORTED;
  if (k.H_in != H_out || k.W_in != k.W_out || k.C_out % 8 || k.pad_t < 0 || k.pad_l < 0 || k.pad_t >= k.R || k.pad_l >= k.S)
    return MBX_ERR_UNSUPPORTED;
  ResK q;
  q.N = N; q.H = H_out; q.W = k.W_out; q.HW = H_out * k.W_out;
  int NB;
  resident_groups(k.C_out, q.NG, q.CPT, NB);
  q.ntiles = N * q.NG;
  k.stats_cap = stats_cap_for(N);                          // one add per image and channel
  const int C8 = k.C_in / 8, RS = k.R * k.S;
  if (k.C_in % 32 || q.HW > 289 || q.HW < 64) return MBX_ERR_UNSUPPORTED;
  q.npi = ((q.HW + 1) * C8 + 63) / 64;
  if (RS > 9) return MBX_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t) { q.dh[t] = q.dw[t] = q.dd[t] = 0; }
  for (int t = 0; t < RS; ++t) { q.dh[t] = t / k.S - k.pad_t; q.dw[t] = t % k.S - k.pad_l; q.dd[t] = q.dh[t] * q.W + q.dw[t]; }
  int grid = q.ntiles < resident_cus() ? q.ntiles : resident_cus();
  if (k.max_wg > 0 && grid > k.max_wg) grid = k.max_wg;
  // consecutive tiles per workgroup (BATCH_SIZE 256: the four channel groups of an image -- its image lands once, and the next
  // group's filter slice while this one's outputs are stored)
  q.per = (q.ntiles + grid - 1) / grid;
  grid = (q.ntiles + q.per - 1) / q.per;
#define MBX_RES(C8_, NB_, RS_, MI_, HWC_)                                                                \
  if (C8 == C8_ && NB == NB_ && RS == RS_ && q.HW <= HWC_ && (HWC_ > 64 || q.HW == 64)) {                 \
    if (k.dry) return MBX_OK;                                                                            \
    return launch_resident<C8_, NB_, RS_, MI_, HWC_>(k, q, grid, s);                                     \
  }
  // block8's 1x3 / 3x1 layers (model.py:53-57: 8 x 8 maps, 192 -> 224 -> 256 channels, forward and data gradient): one pixel
  // block per multiplying wave, all three taps resident
  MBX_RES(24, 4, 3, 1, 64) MBX_RES(28, 4, 3, 1, 64) MBX_RES(28, 3, 3, 1, 64) MBX_RES(32, 4, 3, 1, 64)
  if (q.HW <= 64) return MBX_ERR_UNSUPPORTED;
  // block17's 1x7 / 7x1 layers (model.py:33-37: 17 x 17 maps, 128 -> 160 -> 192 channels)
  MBX_RES(16, 3, 7, 5, 289) MBX_RES(20, 3, 7, 5, 289) MBX_RES(20, 2, 7, 5, 289) MBX_RES(24, 3, 7, 5, 289)
#undef MBX_RES
  return MBX_ERR_UNSUPPORTED;
}

// mbx_conv_desc.tile_config = 99: the PIXEL-RESIDENT pointwise launch (conv_pwres_kernel).  MBX_ERR_UNSUPPORTED for anything but
// a 1x1 / stride-1 / unpadded convolution with C_in 96 / 128 / 320 / 384 / 448 and a residual epilogue (+ relu, + sign bits) or a
// bf16 store masked by relu sign bits (with or without an accumulate source).
int mbx_launch_pwres(void* convk, hipStream_t s) {
  ConvK& k = *reinterpret_cast<ConvK*>(convk);
  if (!k.pw || k.shift || k.stats || k.bw_n || k.C_out % 8) return MBX_ERR_UNSUPPORTED;
  const bool res = k.epi == MBX_EPI_RESIDUAL, accm = k.epi == MBX_EPI_STORE && k.bits && !k.skip;     // (masks by the sign bits only)
  if (!res && !accm) return MBX_ERR_UNSUPPORTED;
  PwK q;
  // (pixel tile 160 where the registers and LDS hold it: residual launches; 128 for the accumulate launch of 384 input
  // channels -- three registers short at 160 -- and for K = 448, whose pixel tile would leave the ring no room)
  if (res) {
    switch (k.C_in) {
      case 96: return launch_pwres<12, 5, 4>(k, q, s);
      case 128: return launch_pwres<16, 5, 4>(k, q, s);
      case 320: return launch_pwres<40, 5, 4>(k, q, s);
      case 384: return launch_pwres<48, 5, 4>(k, q, s);
      case 448: return launch_pwres<56, 4, 4>(k, q, s);
      default: return MBX_ERR_UNSUPPORTED;
    }
  }
  switch (k.C_in) {
    case 96: return launch_pwres<12, 5, 2>(k, q, s);
    case 128: return launch_pwres<16, 4, 2>(k, q, s);
    case 320: return launch_pwres<40, 5, 2>(k, q, s);
    case 384: return launch_pwres<48, 4, 2>(k, q, s);
    case 448: return launch_pwres<56, 4, 2>(k, q, s);
    default: return MBX_ERR_UNSUPPORTED;
  }
}
