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
  int npt, NCS, nct, ctper, nunits;                       // pixel tiles (80 pixels), column splits, column tiles (128 channels), column tiles per split
  int npi;                                                // 1 KB pieces of the resident pixel tile
};
constexpr int kPwPix = 80, kPwMI = 5, kPwCol = 128;
constexpr int kPwThreads = 768, kPwCW = 8;                // waves 0-3 / 4-7: the two multiplying groups; 8-11: loaders

// Vector-memory READS of the multiplying waves as inline asm, waited for by hand.  The compiler's own s_waitcnt insertion is
// exact inside straight-line code but CONSERVATIVE across a loop's back edge: for epilogue operands fetched one column tile
// ahead it waited for everything outstanding -- the previous tile's stores and the next tile's reads included -- which is the
// serialisation this kernel exists to remove.  Loads the compiler does not see are never waited for by it; pw_wait<N>
// (s_waitcnt vmcnt(N), N = the stores issued since) names every destination register, so their uses stay behind it.
__device__ __forceinline__ u32x4 pw_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffffu,
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void pw_load16(u32x4& dst, const u32x4 rsrc, const unsigned off) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(dst) : "v"(off), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void pw_load1(unsigned& dst, const u32x4 rsrc, const unsigned off) {
  asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "=&v"(dst) : "v"(off), "s"(rsrc) : "memory");
}
template <int EV, int N>
__device__ __forceinline__ void pw_wait(u32x4 (&la)[kPwMI][1], u32x4 (&lb)[kPwMI][1], u32x4 (&sh)[2]) {
  if constexpr (EV == 4) {          // (skip rows + the shift row; lb is not used)
    asm volatile("s_waitcnt vmcnt(%7)"
                 : "+v"(la[0][0]), "+v"(la[1][0]), "+v"(la[2][0]), "+v"(la[3][0]), "+v"(la[4][0]), "+v"(sh[0]), "+v"(sh[1])
                 : "n"(N) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(la[0][0]), "+v"(la[1][0]), "+v"(la[2][0]), "+v"(la[3][0]), "+v"(la[4][0]),
                   "+v"(lb[0][0]), "+v"(lb[1][0]), "+v"(lb[2][0]), "+v"(lb[3][0]), "+v"(lb[4][0])
                 : "n"(N) : "memory");
  }
}
// The reads of conv_epilogue_issue_reads (EV 2 / 4: same offsets, same uniform choices) and the shift row of
// conv_epilogue_channels (EV 4; a range-checked buffer read: zeros past C_out or without a table, as there), for pixel
// blocks 0 .. 4 of a lane whose first pixel is mlane and first channel clane.
template <int EV>
__device__ __forceinline__ void pw_issue_reads(const ConvK& p, const int mlane, const int clane, u32x4 (&la)[kPwMI][1], u32x4 (&lb)[kPwMI][1],
                                               u32x4 (&sh)[2]) {
  static_assert(EV == 2 || EV == 4, "accumulate (+ mask) / residual");
  const u32x4 kr = pw_rsrc(p.skip, p.skip ? p.skip_bytes : 0u);
  const u32x4 ar = pw_rsrc(p.acc_src, p.acc_bytes);
  const u32x4 br = pw_rsrc(p.bits, p.bits ? p.bits_bytes : 0u);
  const bool do_acc = EV == 2 && p.accumulate, do_mask = EV == 2 && p.skip != nullptr;     // (uniform)
  const bool do_bits = EV == 2 && p.bits != nullptr;                                       // (uniform; never with do_mask)
  const int c0 = clane;
#pragma unroll
  for (int b = 0; b < kPwMI; ++b) {
    const int m = mlane + b * 16;
    int img, pix;
    epi_pixel<false>(p, m, img, pix);
    const bool ok = m < p.M && c0 < p.C_out;
    const unsigned so = ok ? (unsigned)((img * p.skip_img_stride + pix * p.ld_skip + c0) * 2) : kOOB;
    if constexpr (EV == 2) { la[b][0] = u32x4{0u, 0u, 0u, 0u}; lb[b][0] = u32x4{0u, 0u, 0u, 0u}; }
    if constexpr (EV == 4) {
      pw_load16(la[b][0], kr, so);
    } else {
      const unsigned ao = ok ? (unsigned)((img * p.acc_img_stride + pix * p.ld_acc + c0) * 2) : kOOB;
      if (do_acc) pw_load16(la[b][0], ar, ao);
      if (do_mask) pw_load16(lb[b][0], kr, so);
      if (do_bits) { unsigned byte_; pw_load1(byte_, br, ok ? (unsigned)(m * p.bits_ld + (c0 >> 3)) : kOOB); lb[b][0].x = byte_; }
    }
  }
  if constexpr (EV == 4) {
    const u32x4 sr = pw_rsrc(p.shiftv, p.shiftv ? (unsigned)p.C_out * 4u : 0u);
    pw_load16(sh[0], sr, (unsigned)c0 * 4u);             // (C_out % 8 == 0: a lane's eight channels are all in or all out)
    pw_load16(sh[1], sr, (unsigned)c0 * 4u + 16u);
  }
}

template <int C8>
struct PwG {
  static constexpr int KC = C8 / 4;                       // K steps of 32
  static constexpr int KS = (C8 % 8 == 0) ? 2 : 1;        // K steps per ring slot (64-deep slots where K allows)
  static constexpr int NSL = KC / KS;                     // slots per column tile
  static constexpr int C8S = 4 * KS;                      // 16-byte chunks per filter row of a slot
  static constexpr int SLOT_CH = kPwCol * C8S;            // chunks per slot (16 or 8 KB)
  static constexpr int PT = SLOT_CH / (64 * kRLW);        // pieces per loader wave and slot
  static constexpr int PIX_CH = (kPwPix * C8 + 63) / 64 * 64;
  static constexpr int NST_RAW = (160 * 1024 / 16 - PIX_CH) / SLOT_CH;
  static constexpr int NST = NST_RAW > 8 ? 8 : NST_RAW;   // ring depth
  static constexpr int LDS_BYTES = (NST * SLOT_CH + PIX_CH) * 16;
  static_assert(KC % KS == 0 && NST >= 3 && (NST - 1) * PT < 64, "ring");
};

template <int C8, int EV>
__global__ void __launch_bounds__(kPwThreads)
conv_pwres_kernel(const ConvK p, const PwK q) {
  using G = PwG<C8>;
  constexpr int KC = G::KC, KS = G::KS, NSL = G::NSL, C8S = G::C8S, SLOT_CH = G::SLOT_CH, PT = G::PT, NST = G::NST, MI = kPwMI;
  constexpr int NI = 2, NA = 1;
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  u32x4* const ring = smem;                               // [NST][128 rows][C8S]
  u32x4* const pix = smem + NST * SLOT_CH;                // [80 pixels][C8] (swizzled)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int first = xcd_remap((int)blockIdx.x, (int)gridDim.x), G_ = (int)gridDim.x;
  const int frow = lane & 15, fch = lane >> 4;

  // the pixel tile: pieces wave, wave + 8, ...; chunk ci = row ci / C8 (pixel m0 + row), slot ci % C8 holds source chunk slot ^ key(row)
  auto issue_pixels = [&](const int m0) {
    for (int i = wave; i < q.npi; i += kPwThreads / 64) {
      const int ci = i * 64 + lane;
      const int row = ci / C8, cs = ci - row * C8;
      const int c = cs ^ rkey<C8>(row);
      const int m = m0 + row;
      const bool ok = row < kPwPix && m < p.M;
      const int img = (int)fast_div((unsigned)(ok ? m : 0), p.mg_hw, p.sh_hw);
      glds16(xr, pix + i * 64, ok ? (img * p.x_img_stride + (m - img * p.HW_out) * p.ldx + c * 8) * 2 : (int)kOOB);
    }
  };

  if (wave >= kPwCW) {
    // ------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - kPwCW;
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
      issue_pixels(pt * kPwPix);
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
  // TWO GROUPS of four (waves 0-3 and 4-7: one wave of each per SIMD) that ALTERNATE column tiles: while one group multiplies
  // column tile i, the other converts, masks and stores tile i - 1 -- the epilogue is ~300 vector instructions and 40 KB of
  // memory traffic per tile and wave, which on four waves alone was a third of every tile's time (stamps: K loop 2.1 us,
  // epilogue 1.1 us) with the matrix cores idle.  One barrier per ring slot for everyone: the multiplying group's sits in its K
  // loop, the other group's between the pixel blocks of its epilogue (whose reads were issued before ITS K loop and whose
  // stores nobody waits for, so it never holds a barrier up for memory).
  // (EV 2: the relu mask comes from the SIGN BITS only -- the tensor-mask path is compiled out, 128 registers hold no 20 more)
  constexpr int EVC = EV, BMODE = EV == 2 ? 2 : 1;
  const int grp = wave >> 2, gw = wave & 3;
  const int lds0 = lds_addr(smem);
  int EA[NI], Ep[MI];
#pragma unroll
  for (int a = 0; a < NI; ++a) EA[a] = rencb<C8S>(32 * gw + 16 * a + frow, fch, lds0);
#pragma unroll
  for (int f = 0; f < MI; ++f) Ep[f] = rencb<C8>(16 * f + frow, fch, lds0 + NST * SLOT_CH * 16);
#ifdef MBX_I5_STAMPS
  // debug build (tools/pw_stamps.py): [block][column tile (first 8)][K loop starts, K loop done, epilogue done] of the first unit
  const bool stamp = p.stamps && (tid & 255) == 0 && blockIdx.x < 64;
#define MBXP_STAMP(ci, i) do { if (stamp && u == first && (ci) < 8) p.stamps[(blockIdx.x * 8 + (ci)) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define MBXP_STAMP(ci, i) do { } while (0)
#endif
  for (int u = first; u < q.nunits; u += G_) {
    const int pt = u / q.NCS, cs_ = u - pt * q.NCS;
    const int ct0 = cs_ * q.ctper, ct1 = min(ct0 + q.ctper, q.nct);
    const int nct = ct1 - ct0;
    const int m0 = pt * kPwPix;
    issue_pixels(m0);
    wait_vmcnt<0>();
    lds_readback_wait(lds_readback_issue(pix + lane));
    raw_barrier();                                        // P
#pragma unroll
    for (int a = 0; a < NI; ++a) asm volatile("" : "+v"(EA[a]));       // (opaque per unit: no hoisting of every K step's address)
#pragma unroll
    for (int f = 0; f < MI; ++f) asm volatile("" : "+v"(Ep[f]));
    const int mlane = m0 + frow;
    f32x4 acc[NI][MI];
    u32x4 la[MI][NA], lb[MI][NA];
    // the epilogue of this group's column tile CE (local index), its MI pixel blocks dealt over the NSL slot barriers of the
    // tile the other group multiplies meanwhile (WITH = false: the unit's last tile -- no one multiplies, no barriers are left)
#define PW_EPILOGUE(CE, WITH)                                                                                        \
  do {                                                                                                               \
    const int clane_e = (ct0 + (CE)) * kPwCol + 32 * gw + 8 * fch;                                                   \
    float s1[NA][8], s2[NA][8], sc[NA][8], sh[NA][8];                                                                \
    conv_epilogue_channels<EVC, NA>(p, clane_e, sc, sh, s1, s2);                                                     \
    _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) {                                                             \
      _Pragma("unroll") for (int b = 0; b < MI; ++b)                                                                 \
        if (b >= (MI * sl) / NSL && b < (MI * (sl + 1)) / NSL)                                                       \
          conv_epilogue_finish<EVC, false, NI, MI, 1, BMODE>(p, acc, mlane, clane_e, b, *reinterpret_cast<u32x4 (*)[1][NA]>(&la[b]), \
                                                      *reinterpret_cast<u32x4 (*)[1][NA]>(&lb[b]), sh, sc, s1, s2);  \
      if (WITH) raw_barrier();                                                                                       \
    }                                                                                                                \
  } while (0)
    int so = 0;                                           // byte offset of the ring position being read
    for (int ci = 0; ci < nct; ++ci) {
      if ((ci & 1) != grp) {
        // ---- the other group multiplies column tile ci: this group's previous tile goes out (or, at the start, nothing does)
        if (ci > 0) { PW_EPILOGUE(ci - 1, true); MBXP_STAMP(ci - 1, 2); }
        else {
#pragma unroll 1
          for (int sl = 0; sl < NSL; ++sl) raw_barrier();
        }
        so += NSL * SLOT_CH * 16;                         // (the ring moved on by this tile's slots)
        while (so >= NST * SLOT_CH * 16) so -= NST * SLOT_CH * 16;
        continue;
      }
      const int clane = (ct0 + ci) * kPwCol + 32 * gw + 8 * fch;
      MBXP_STAMP(ci, 0);
      // the epilogue's reads (residual skip / accumulate source / mask), issued before the K loop: they land while the tile is
      // multiplied (the multiplying waves issue no other vector-memory instruction)
      conv_epilogue_issue_reads<EVC, false, NA, MI, BMODE>(p, mlane, clane, 0, la, lb);
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int f = 0; f < MI; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 wf[2][NI], pf[2][MI];
#pragma unroll
      for (int a = 0; a < NI; ++a) wf[0][a] = rread<C8S>(EA[a] + so, 0);
#pragma unroll
      for (int f = 0; f < MI; ++f) pf[0][f] = rread<C8>(Ep[f], 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < KC; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        const int j1 = (s + 1) % KS;
        if (j1 == 0) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done
          raw_barrier();                                        // B: the slot may be refilled; the next one is published
          so = so == (NST - 1) * SLOT_CH * 16 ? 0 : so + SLOT_CH * 16;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < KC) {
#pragma unroll
          for (int a = 0; a < NI; ++a) wf[nxt][a] = rread<C8S>(EA[a] + so, j1);
#pragma unroll
          for (int f = 0; f < MI; ++f) pf[nxt][f] = rread<C8>(Ep[f], s + 1);
        }
#pragma unroll
        for (int f = 0; f < MI; ++f)
#pragma unroll
          for (int a = 0; a < NI; ++a)
            acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][a], pf[cur][f], acc[a][f], 0, 0, 0);
        if (s + 1 < KC) sched_interleave<NI + MI, NI * MI>();
        __builtin_amdgcn_sched_barrier(0);
      }
      MBXP_STAMP(ci, 1);
    }
    if (((nct - 1) & 1) == grp) { PW_EPILOGUE(nct - 1, false); MBXP_STAMP(nct - 1, 2); }      // the unit's last tile
  }
#undef MBXP_STAMP
#undef PW_EPILOGUE
}

template <int C8>
int launch_pwres(const ConvK& k, const PwK& q, int grid, hipStream_t s) {
  using G = PwG<C8>;
  constexpr int lds = G::LDS_BYTES;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr[2] = {false, false};
  const int ev = k.epi == MBX_EPI_RESIDUAL ? 1 : 0;
  if (!attr[ev]) {
    if (ev) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pwres_kernel<C8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pwres_kernel<C8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr[ev] = true;
  }
  if (ev) hipLaunchKernelGGL((conv_pwres_kernel<C8, 4>), dim3(grid), dim3(kPwThreads), lds, s, k, q);
  else hipLaunchKernelGGL((conv_pwres_kernel<C8, 2>), dim3(grid), dim3(kPwThreads), lds, s, k, q);
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

// mbx_conv_desc.tile_config = 99: the PIXEL-RESIDENT pointwise launch (conv_pwres_kernel).  MBX_ERR_UNSUPPORTED for anything but
// a 1x1 / stride-1 / unpadded convolution with C_in 96 / 128 / 320 / 384 / 448 and a residual epilogue (+ relu, + sign bits) or a
// bf16 store masked by relu sign bits (with or without an accumulate source).
int mbx_launch_pwres(void* convk, hipStream_t s) {
  ConvK& k = *reinterpret_cast<ConvK*>(convk);
  if (!k.pw || k.shift || k.stats || k.bw_n || k.C_out % 8) return MBX_ERR_UNSUPPORTED;
  const bool res = k.epi == MBX_EPI_RESIDUAL, accm = k.epi == MBX_EPI_STORE && k.bits && !k.skip;     // (masks by the sign bits only)
  if (!res && !accm) return MBX_ERR_UNSUPPORTED;
  PwK q;
  q.npt = (k.M + kPwPix - 1) / kPwPix;
  q.nct = (k.C_out + kPwCol - 1) / kPwCol;
  const int ncu = resident_cus();
  int ncs = ncu / q.npt;
  if (ncs < 1) ncs = 1;
  if (ncs > q.nct) ncs = q.nct;
  q.ctper = (q.nct + ncs - 1) / ncs;
  q.NCS = (q.nct + q.ctper - 1) / q.ctper;
  q.nunits = q.npt * q.NCS;
  const int C8 = k.C_in / 8;
  q.npi = (kPwPix * C8 + 63) / 64;
  int grid = q.nunits < ncu ? q.nunits : ncu;
  if (k.max_wg > 0 && grid > k.max_wg) grid = k.max_wg;
#define MBX_PW(C8_)                                                   \
  if (k.C_in == 8 * C8_) {                                           \
    if (k.dry) return MBX_OK;                                        \
    return launch_pwres<C8_>(k, q, grid, s);                         \
  }
  MBX_PW(12) MBX_PW(16) MBX_PW(40) MBX_PW(48) MBX_PW(56)
#undef MBX_PW
  return MBX_ERR_UNSUPPORTED;
}
