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

namespace {

constexpr int kRThreads = 512;                            // waves 0-3 multiply, waves 4-7 load; all eight stage the image
constexpr int kRCW = 4, kRLW = 4;

struct ResK {
  int N, H, W, HW;                                        // images; map (input = output: stride 1, "same" geometry)
  int NG, CPT;                                            // channel groups per image, output channels per group (a multiple of 8)
  int ntiles;                                             // N * NG; tile t = image t / NG, group t % NG
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
  constexpr int NP = NB / 2;                              // paired blocks
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  float* const red = reinterpret_cast<float*>(smem);      // [4 waves][16 NB][2]
  u32x4* const ring = smem + G::RED_CH;                   // [D][SLOTP]
  u32x4* const img = ring + D * SLOTP;                    // [HW + 1][C8] (swizzled); row HW = zeros; last: its final piece may overhang

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int first = xcd_remap((int)blockIdx.x, (int)gridDim.x), G_ = (int)gridDim.x;
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
    for (int t = first; t < q.ntiles; t += G_) {
      const int im = t / q.NG, g = t - im * q.NG;
      const int cb = g * q.CPT, ce = min(cb + q.CPT, p.C_out);
      const int wbase = cb * RS * (8 * C8) * 2;
      auto issue_tap = [&](const int tap) {
        u32x4* dst = ring + (tap % D) * SLOTP + lw * 64;
#pragma unroll
        for (int i = 0; i < PT; ++i)
          glds16(wr, dst + i * (kRLW * 64), (wv[i] && cb + rowch[i] < ce) ? wbase + wo[i] + tap * (8 * C8) * 2 : (int)kOOB);
      };
      issue_image(im);
#pragma unroll
      for (int tap = 0; tap < DD; ++tap) issue_tap(tap);
      wait_vmcnt<(DD - 1) * PT>();                        // the image and tap 0 have retired (this wave's share) ...
      lds_readback_wait(lds_readback_issue(ring + (PT - 1) * (kRLW * 64) + lw * 64 + lane));   // ... and landed
      raw_barrier();                                      // P: image + tap 0 published
#pragma unroll
      for (int tap = 0; tap < RS; ++tap) {
        if (tap + 1 < RS) {
          // tap + 1 has retired: behind it this wave has issued the taps up to min(RS - 1, tap + D - 1)
          const int newest = (tap + D - 1 < RS - 1) ? tap + D - 1 : RS - 1;
          switch (newest - (tap + 1)) {                   // (folded: the loop is unrolled)
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<PT>(); break;
            case 2: wait_vmcnt<2 * PT>(); break;
            case 3: wait_vmcnt<3 * PT>(); break;
            case 4: wait_vmcnt<4 * PT>(); break;
            case 5: wait_vmcnt<5 * PT>(); break;
            case 6: wait_vmcnt<6 * PT>(); break;
            case 7: wait_vmcnt<7 * PT>(); break;
            default: wait_vmcnt<0>(); break;
          }
          lds_readback_wait(lds_readback_issue(ring + ((tap + 1) % D) * SLOTP + (PT - 1) * (kRLW * 64) + lw * 64 + lane));
        }
        raw_barrier();                                    // B_tap: slot tap % D is free, tap + 1 is published
        if (tap + D < RS) issue_tap(tap + D);
      }
      if constexpr (EV == 1) raw_barrier();               // the multiplying waves' statistics reduce
    }
    return;
  }

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
  int E[2][MI];
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
#define MBXR_STAMP(i) do { if (stamp && t == first) p.stamps[(blockIdx.x * 8) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define MBXR_STAMP(i) do { } while (0)
#endif
  for (int t = first; t < q.ntiles; t += G_) {
    const int im = t / q.NG, g = t - im * q.NG;
    const int cb = g * q.CPT, ce = min(cb + q.CPT, p.C_out);
    MBXR_STAMP(0);
    issue_image(im);
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
    tap_rows(0, E[0]);
    wait_vmcnt<0>();
    lds_readback_wait(lds_readback_issue(img + lane));
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
    bf16x8 wf[2][NB], pf[2][MI];
#pragma unroll
    for (int a = 0; a < NB; ++a) wf[0][a] = rread<C8>(EA[a], 0);
#pragma unroll
    for (int f = 0; f < MI; ++f) pf[0][f] = rread<C8>(E[0][f], 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      const int tap1 = (s + 1) / KC, j1 = (s + 1) % KC;
      if (j1 == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot (the last tap: of the image) are done
        raw_barrier();                                        // B_tap: its slot may be refilled; tap + 1 is published
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s % KC == 0 && s / KC + 1 < RS) tap_rows(s / KC + 1, E[(s / KC + 1) & 1]);     // the next tap's rows (used KC - 1 steps on)
      if (s + 1 < NS) {
        const int so = (tap1 % D) * SLOTP * 16;
#pragma unroll
        for (int a = 0; a < NB; ++a) wf[nxt][a] = rread<C8>(EA[a] + so, j1);
#pragma unroll
        for (int f = 0; f < MI; ++f) pf[nxt][f] = rread<C8>(E[tap1 & 1][f], j1);
      }
      // pixel-block-major: the fragment read LAST (pixel block MI - 1) is needed last in the next step
#pragma unroll
      for (int f = 0; f < MI; ++f)
#pragma unroll
        for (int a = 0; a < NB; ++a)
          acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][a], pf[cur][f], acc[a][f], 0, 0, 0);
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
// stride-1 convolution with "same" geometry (forward, or the data gradient of one) of 7 taps on a map of at most 17 x 17 = 289
// pixels with C_in 128 / 160 / 192, C_out <= 192 (a multiple of 8) and a bf16 store epilogue with or without statistics, or
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
  const int C8 = k.C_in / 8, RS = k.R * k.S;
  if (k.C_in % 32 || q.HW > 289 || q.HW < 64) return MBX_ERR_UNSUPPORTED;
  q.npi = ((q.HW + 1) * C8 + 63) / 64;
  if (RS > 9) return MBX_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t) { q.dh[t] = q.dw[t] = q.dd[t] = 0; }
  for (int t = 0; t < RS; ++t) { q.dh[t] = t / k.S - k.pad_t; q.dw[t] = t % k.S - k.pad_l; q.dd[t] = q.dh[t] * q.W + q.dw[t]; }
  int grid = q.ntiles < resident_cus() ? q.ntiles : resident_cus();
  if (k.max_wg > 0 && grid > k.max_wg) grid = k.max_wg;
#define MBX_RES(C8_, NB_, RS_)                                                                           \
  if (C8 == C8_ && NB == NB_ && RS == RS_) {                                                             \
    if (k.dry) return MBX_OK;                                                                            \
    return launch_resident<C8_, NB_, RS_, 5, 289>(k, q, grid, s);                                        \
  }
  MBX_RES(16, 3, 7) MBX_RES(20, 3, 7) MBX_RES(20, 2, 7) MBX_RES(24, 3, 7)
#undef MBX_RES
  return MBX_ERR_UNSUPPORTED;
}
