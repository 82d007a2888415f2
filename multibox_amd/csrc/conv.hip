// libmbx: implicit-GEMM convolution on MFMA (gfx950), forward + data gradient in one kernel,
// and the weight-gradient kernel.  NHWC bf16 views, KRSC bf16 filters, fp32 accumulate.
//
// GEMM view (forward):  Y[m, n] = sum_k P[m, k] * W[n, k]
//   m = (img, oh, ow) output pixel, n = output channel, k = (r, s, c) filter tap x input channel.
// The MFMA is issued with the FILTER tile as the A operand and the PIXEL tile as the B operand
// (v_mfma_f32_16x16x32_bf16: D[row = channel][col = pixel]): each lane holds four consecutive channels
// of one pixel.  The fp32 tile is staged through LDS at the end and written row-wise (16 B per lane).
//
// LDS tiles are [rows][64 k] bf16 = 128-B rows of eight 16-B chunks, XOR-swizzled
// (chunk ^ (row & 7)) so the ds_read_b128 fragment reads are bank-conflict free; a 3-deep ring filled
// by LDS-DMA, one barrier per 64-deep K step, two tiles of global latency in flight.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "conv_common.h"

// conv5.hip: persistent igemm5 launches (mbx_conv_desc.tile_config = kI5Flag + index + 1)
int mbx_launch_igemm5(void* convk, int index, hipStream_t s);
// conv7.hip: the panel-resident pointwise launch (mbx_conv_desc.tile_config = kI7Cfg)
int mbx_launch_igemm7(void* convk, hipStream_t s);
// convd.hip: the direct 3x3 launch for few channels on large maps (mbx_conv_desc.tile_config = kDirectCfg)
int mbx_launch_direct3(void* convk, int N, int H_out, hipStream_t s);
int mbx_direct3_grid(int N, int H_out, int W_out);
int mbx_launch_directw(void* convk, int N, int H_out, hipStream_t s);
int mbx_directw_grid(int N, int H_out, int W_out);
// convr.hip: the resident-image launch for multi-tap convolutions on small maps (mbx_conv_desc.tile_config = kResidentCfg)
int mbx_launch_resident(void* convk, int N, int H_out, hipStream_t s);
int mbx_resident_rows(int N);
int mbx_launch_pwres(void* convk, hipStream_t s);
extern const int mbx_i5_tiles[][2];
extern const int mbx_i5_num_tiles;

namespace {

// EV: epilogue variant fixed at compile time (no per-element branching):
//   0 store, 1 store + BN statistics partials, 2 accumulate into y (+ optional relu mask from `skip`),
//   3 affine (+relu), 4 residual (+relu), 5 float32 store, 6 store + batch-norm BACKWARD statistics (data gradients).
// SH: the stride-2 data gradient (input dilated by 2).  Its own instantiation, so that the parity walk below costs
// the other 400 launches per step nothing (as a run-time branch in one kernel it cost them 0.6 ms per step).
// MODE 1: pointwise (1x1, unit stride, no padding) -- the two-instruction address path, no tap walk: 60 % of the launches.
// NSTG: ring depth.  3 = two tiles in flight; 2 = one, for 64 KB of LDS per 128x128 eight-wave block, so that TWO
// blocks share a CU and one's epilogue (HBM-bound: skip read / accumulate / store) overlaps the other's loop.
// (the body is a device function of (problem, block id, number of blocks) so that conv_igemm3_pair_kernel can run TWO
// problems in one grid; conv_igemm3_kernel passes blockIdx.x / gridDim.x)
template <int BM, int BN, int WNW, int WMW, int EV, int MODE = 0, int NSTG = 3>
__device__ __forceinline__ void conv_igemm3_body(const ConvK& p, const int bid_, const int nblk_) {
  constexpr bool SH = MODE == 2, PW = MODE == 1;
  static_assert(WNW * WMW == 4 || WNW * WMW == 8, "four or eight waves");
  constexpr int NT = 64 * WNW * WMW, RPP = NT / 8;      // threads, tile rows filled per DMA pass
  constexpr int TN = BN / WNW, TM = BM / WMW, NI = TN / 16, MI = TM / 16;
  constexpr int PI = BM / RPP, WI = BN / RPP, NL = PI + WI;
  constexpr int STAGE = (BM + BN) * 8;                 // 16-B slots per stage
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wn = wave % WNW, wm = wave / WNW;
  int lid = xcd_remap(bid_, nblk_);
  int split = 0;                                        // split-K slice of this block (float32 partial tiles only)
  if constexpr (EV == 5) { const int nt = p.tiles_m * p.tiles_n; split = lid / nt; lid -= split * nt; }
  const int tile_n = lid % p.tiles_n, tile_m = lid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ (lrow & 7);            // source chunk of this lane's LDS slot (pixel tile)
  // The FILTER tile has its own swizzle: its rows are read by the MFMA A operand in a PERMUTED order (see the epilogue:
  // a lane must end up with 8 consecutive output channels), rows 8 (f >> 2) + (f & 3) + 4 (a & 1) + 32 (a >> 1) for
  // fragment row f of block a, and the key ((row >> 3) & 3) << 1 | (row >> 1) & 1 makes exactly those reads
  // bank-conflict free (the 16 lanes a ds_read_b128 services together hit 16 different 16-byte bank groups).
  const int chunkw = (tid & 7) ^ ((((lrow >> 3) & 3) << 1) | ((lrow >> 1) & 1));
  int hb[PI], wb[PI], ro[PI];
#pragma unroll
  for (int i = 0; i < PI; ++i) {
    const int m = m0 + lrow + RPP * i;
    const bool mv = m < p.M;
    const unsigned mm = mv ? (unsigned)m : 0u;
    int img, oh, ow;
    if (SH && p.parity) decode_pixel_parity(p, mm, img, oh, ow); else decode_pixel(p, mm, img, oh, ow);
    hb[i] = mv ? oh * p.mul - p.pad_t : -(1 << 24);
    wb[i] = ow * p.mul - p.pad_l;
    ro[i] = SH ? img * p.x_img_stride * 2 : (img * p.x_img_stride + (hb[i] * p.W_in + wb[i]) * p.ldx) * 2;
    if (PW && !mv) ro[i] = (int)kOOB;
  }
  int wo[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int n = n0 + lrow + RPP * i;
    wo[i] = n < p.C_out ? n * p.Ktot * 2 : -1;
  }

  f32x4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fch = lane >> 4;
  // fragment read offsets (16-B slots): row*8 + ((kk*4 + fch) ^ (row & 7)); row & 7 == frow & 7
  const int fr0 = frow * 8 + (fch ^ (frow & 7));
  const int fr1 = frow * 8 + ((4 + fch) ^ (frow & 7));
  // filter fragments: block a reads tile rows 32 (a >> 1) + 4 (a & 1) + [8 (frow >> 2) + (frow & 3)]  (slot = row * 8 + chunk ^ key)
  const int fwrow = 8 * (frow >> 2) + (frow & 3), fwkey = ((frow >> 2) << 1) | ((frow >> 1) & 1);
  const int fw0 = fwrow * 8 + (fch ^ fwkey);
  const int fw1 = fwrow * 8 + ((4 + fch) ^ fwkey);
  // stride-2 data gradient, tile inside one parity class: only taps with kr = kr0 (mod 2), ks = ks0 (mod 2) meet
  // non-zero rows of the dilated input -- walk those only (steps of 2), a quarter of the K tiles
  int kr0 = 0, ks0 = 0, kstep = 1, nk = (p.Ktot + 63) >> 6;
  if (SH && p.skip_taps) {
    const int mlast = min(m0 + BM, p.M) - 1;
    const int c0 = pixel_class(p, m0);
    if (c0 == pixel_class(p, mlast)) {
      kr0 = (p.pad_t + (c0 >> 1)) & 1;
      ks0 = (p.pad_l + (c0 & 1)) & 1;
      kstep = 2;
      nk = ((p.R - kr0 + 1) >> 1) * ((p.S - ks0 + 1) >> 1) * (p.C_in >> 6);
    }
  }
  int kt0 = 0;                                          // first K step of this block (split-K slices start later)
  if constexpr (EV == 5) { kt0 = split * p.kps; nk = min(p.kps, nk - kt0); }
  int kc = chunk * 8 + kt0 * 64, kr = kr0, ks = ks0;
  while (kc >= p.C_in) { kc -= p.C_in; if ((ks += kstep) >= p.S) { ks = ks0; kr += kstep; } }
  const int ldx2 = p.ldx * 2;
  int st_issue = 0, st_comp = 0;                        // ring positions
  // LDS-DMA of K tile LT into ring slot st_issue.  (A macro, not a lambda: the three call sites must be
  // straight-line code so that the MFMA accumulators never leave the accumulator registers.)
#define MBX_ISSUE_TILE(LT)                                                                                   \
  do {                                                                                                       \
    u32x4* sp = smem + st_issue * STAGE + wave * 64;                                                         \
    const bool kv = kr < p.R;                                                                                \
    if (PW) {                                                                                                \
      _Pragma("unroll") for (int i = 0; i < PI; ++i)                                                         \
        glds16(xr, sp + i * NT, (kv && ro[i] >= 0) ? (ro[i] + kc * 2) : (int)kOOB);                          \
    } else if (!SH) {                                                                                        \
      const int toff = (kr * p.W_in + ks) * ldx2 + kc * 2;                                                   \
      _Pragma("unroll") for (int i = 0; i < PI; ++i) {                                                       \
        const bool ok = kv && ((unsigned)(hb[i] + kr) < (unsigned)p.H_in) &&                                 \
                        ((unsigned)(wb[i] + ks) < (unsigned)p.W_in);                                         \
        glds16(xr, sp + i * NT, ok ? (ro[i] + toff) : (int)kOOB);                                            \
      }                                                                                                      \
    } else {                                                                                                 \
      _Pragma("unroll") for (int i = 0; i < PI; ++i) {                                                       \
        const int hn = hb[i] + kr, wn_ = wb[i] + ks;                                                         \
        const bool ok = kv && (((hn | wn_) & 1) == 0) && ((unsigned)(hn >> 1) < (unsigned)p.H_in) &&         \
                        ((unsigned)(wn_ >> 1) < (unsigned)p.W_in);                                           \
        glds16(xr, sp + i * NT, ok ? (ro[i] + ((hn >> 1) * p.W_in + (wn_ >> 1)) * ldx2 + kc * 2) : (int)kOOB); \
      }                                                                                                      \
    }                                                                                                        \
    /* the filter lane's own K position (its chunk differs from the pixel lane's): linear in K, except in the     \
       tap-skipping walk, where C_in % 64 == 0 and a K tile lies inside one tap */                             \
    const int kb = (SH && kstep == 2) ? ((kr * p.S + ks) * p.C_in + kc + (chunkw - chunk) * 8) * 2                \
                                      : (((LT) + kt0) * 64 + chunkw * 8) * 2;                                  \
    const bool kvw = (SH && kstep == 2) ? kv : (kb < p.Ktot * 2);                                              \
    u32x4* sw = sp + BM * 8;                                                                                 \
    _Pragma("unroll") for (int i = 0; i < WI; ++i)                                                           \
      glds16(wr, sw + i * NT, (kvw && wo[i] >= 0) ? (wo[i] + kb) : (int)kOOB);                               \
    kc += 64;                                                                                                \
    while (kc >= p.C_in) { kc -= p.C_in; if ((ks += kstep) >= p.S) { ks = ks0; kr += kstep; } }              \
    st_issue = st_issue == NSTG - 1 ? 0 : st_issue + 1;                                                      \
  } while (0)

  static_assert(NSTG == 2 || NSTG == 3, "ring depth");
  // LANDING HAND-OFF (every ring in this file and in conv5.hip).  A K tile staged by LDS-DMA is read by OTHER waves.
  // `s_waitcnt vmcnt` (the issuing wave's DMAs have retired) + `s_barrier` turned out NOT to order those reads behind
  // the DMA's LDS write: with another stream's kernels sharing the CUs (input augmentation on a side stream, RCCL in
  // data-parallel runs) a reader saw a slot's PREVIOUS contents about once in 10^5 launches of the 2-deep tiles
  // (round 2: one wave's 32 pixels x 1 channel NaN, then the whole stem; tools/side_stream_stress.py).  The 3-deep
  // rings had the same wait -> barrier -> read distance and were only protected by their DMA being issued a K step
  // earlier -- a timing margin, not an ordering.  The rule now, everywhere: a wave publishes a tile only after it has
  // READ BACK one of its own DMA destinations of that tile (its last piece) and that read has returned
  // (`lgkmcnt(0)`) -- its LDS-DMA writes drain through the LDS pipe in front of the read -- and only then joins the
  // barrier behind which the other waves read.  In the deep rings the wait + read-back of tile it+1 sit in the MIDDLE of
  // step `it` (behind the first half of its MFMAs / of the loaders' DMA issue) so that the read's round trip is hidden;
  // the 2-deep rings, which wait for the tile they have just issued, pay it at the end of the step (+0.2 ms per step).
  u32x4* const probe_base = smem + (PI + WI - 1) * NT + wave * 64 + lane;   // this lane's slot of the wave's LAST piece
  MBX_ISSUE_TILE(0);
  if (NSTG == 3 && nk > 1) { MBX_ISSUE_TILE(1); wait_vmcnt<NL>(); } else wait_vmcnt<0>();
  lds_readback_wait(lds_readback_issue(probe_base));
  raw_barrier();
  for (int it = 0; it < nk; ++it) {
    const bool more = it + NSTG - 1 < nk;
    if (more) MBX_ISSUE_TILE(it + NSTG - 1);
    st_comp = st_comp == NSTG - 1 ? 0 : st_comp + 1;    // now the slot of tile it+1; tile `it` is in st_prev
    const int st_prev = st_comp == 0 ? NSTG - 1 : st_comp - 1;
#ifndef MBX_NO_PROBE_I3                      // (debug builds only: A/B of what the hand-off costs)
    unsigned probe;
    if constexpr (NSTG == 3) {
      // 3-deep: tile it+1 was issued a whole step ago: wait for it HERE (tile it+2 is in flight meanwhile) and start
      // the read-back; it returns while this step's fragments are read and multiplied.  (No control flow may sit
      // inside the MFMA block below: the accumulators would leave the accumulator registers.)
      if (more) wait_vmcnt<NL>(); else wait_vmcnt<0>();
      probe = lds_readback_issue(probe_base + st_comp * STAGE);
    }
#endif
    {                                                   // MFMA on tile `it`
      const u32x4* cP = smem + st_prev * STAGE + (wm * TM) * 8;
      const u32x4* cW = smem + st_prev * STAGE + BM * 8 + (wn * TN) * 8;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int fr = kk ? fr1 : fr0, fw = kk ? fw1 : fw0;
        bf16x8 wf[NI], pf[MI];
#pragma unroll
        for (int a = 0; a < NI; ++a) wf[a] = __builtin_bit_cast(bf16x8, cW[(a >> 1) * 256 + (a & 1) * 32 + fw]);
#pragma unroll
        for (int b = 0; b < MI; ++b) pf[b] = __builtin_bit_cast(bf16x8, cP[b * 128 + fr]);
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
          for (int b = 0; b < MI; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
      }
    }
#ifndef MBX_NO_PROBE_I3
    if constexpr (NSTG == 2) {
      wait_vmcnt<0>();                                  // tile it+1 (issued at the top of this step) has retired
      probe = lds_readback_issue(probe_base + st_comp * STAGE);
    }
    lds_readback_wait(probe);                           // the read-back has returned: publish
#else
    if (NSTG == 3 && more) wait_vmcnt<NL>(); else wait_vmcnt<0>();
#endif
    raw_barrier();
  }
#undef MBX_ISSUE_TILE
  wait_vmcnt<0>();
  // ---------------------------------------------------------------- epilogue: straight from the accumulators
  // acc[a][b][r] = output channel  wn TN + 32 (a >> 1) + 8 fch + 4 (a & 1) + r  of pixel  wm TM + 16 b + frow  (the filter rows
  // were fed to the MFMA in that order): blocks 2A and 2A + 1 together give a lane EIGHT CONSECUTIVE channels of one pixel
  // = one 16-byte bf16 store, and the four lanes of a pixel cover 64 contiguous bytes.  No LDS staging, no barrier, the
  // ring is not needed any more: every global read of the epilogue (residual skip, accumulate source, ReLU mask) is an
  // independent 16-byte load that can be in flight with all the others of the lane (round 2 staged the tile through
  // LDS and walked it row-wise in 16-32 passes, each behind a barrier and each stalled on its own loads -- as long as
  // the K loop on the K <= 448 layers).
  static_assert(NI % 2 == 0, "wave tile: a multiple of 32 output channels");
  constexpr int NA = NI / 2;
  const int cl0 = wn * TN + fch * 8;                           // + 32 A: first of this lane's 8 channels inside the tile
  if constexpr (EV == 5) {
    // float32 head outputs (tiny launches): scalar stores
#pragma unroll
    for (int b = 0; b < MI; ++b) {
      const int m = m0 + wm * TM + b * 16 + frow;
      if (m >= p.M) continue;
      const int img = (int)fast_div((unsigned)m, p.mg_hw, p.sh_hw), pix = m - img * p.HW_out;
      float* yrow = reinterpret_cast<float*>(p.y) + split * p.y_split_stride + img * p.y_img_stride + pix * p.ldy;
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        const int c0 = n0 + cl0 + 32 * (a >> 1) + 4 * (a & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c0 + r < p.C_out) yrow[c0 + r] = acc[a][b][r];
      }
    }
  } else {
    float s1[NA][8], s2[NA][8];
    // (reads in flight per lane: all of the tile's, or two pixel blocks at a time where that would take > 64 registers)
    conv_epilogue_direct<EV, SH, NI, MI, ((EV == 2 || EV == 6) && MI * NA >= 8 && MI % 2 == 0) ? 2 : MI>(p, acc, m0 + wm * TM + frow, n0 + cl0, s1, s2);
    if constexpr (EV == 1 || EV == 6) {
      // batch-norm statistics partials of this tile (of the STORED, bf16-rounded values), in a fixed order: the lane's
      // MI pixels (above) -> the 16 lanes that share its channels (DPP row sums) -> the WMW waves along the pixel
      // dimension through LDS (the ring is idle: every wave is past the last barrier of the K loop)
      float* red = reinterpret_cast<float*>(smem);             // [WMW][BN][2]
#pragma unroll
      for (int A = 0; A < NA; ++A) { row_sum16_x8(s1[A]); row_sum16_x8(s2[A]); }
#pragma unroll
      for (int A = 0; A < NA; ++A)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x1 = s1[A][j], x2 = s2[A][j];
          if (frow == 0) {
            red[((wm * BN) + cl0 + 32 * A + j) * 2] = x1;
            red[((wm * BN) + cl0 + 32 * A + j) * 2 + 1] = x2;
          }
        }
      __syncthreads();
      if (tid < BN && n0 + tid < p.C_out) {
        float x1 = 0.f, x2 = 0.f;
#pragma unroll
        for (int w = 0; w < WMW; ++w) { x1 += red[(w * BN + tid) * 2]; x2 += red[(w * BN + tid) * 2 + 1]; }
        if constexpr (EV == 6) bw_stats_write(p, tile_m, n0 + tid, x1, x2); else stats_write(p, tile_m, n0 + tid, x1, x2);
      }
    }
  }
}

template <int BM, int BN, int WNW, int WMW, int EV, int MODE = 0, int NSTG = 3>
__global__ void __launch_bounds__(64 * WNW * WMW)
conv_igemm3_kernel(const ConvK p) {
  conv_igemm3_body<BM, BN, WNW, WMW, EV, MODE, NSTG>(p, (int)blockIdx.x, (int)gridDim.x);
}

// TWO independent problems of the same instantiation in one grid (blocks [0, n0) run p0, the rest p1): the sibling
// convolutions of a batch-norm group (block35's two 3x3 branches, forward and data gradient) are 9-11 us launches that do
// not fill the chip; a kernel costs >= 4.4 us whatever it does.  Same code per block: results bit-identical to two launches.
template <int BM, int BN, int WNW, int WMW, int EV, int MODE = 0, int NSTG = 3>
__global__ void __launch_bounds__(64 * WNW * WMW)
conv_igemm3_pair_kernel(const ConvK p0, const ConvK p1, const int n0) {
  if ((int)blockIdx.x < n0) conv_igemm3_body<BM, BN, WNW, WMW, EV, MODE, NSTG>(p0, (int)blockIdx.x, n0);
  else conv_igemm3_body<BM, BN, WNW, WMW, EV, MODE, NSTG>(p1, (int)blockIdx.x - n0, (int)gridDim.x - n0);
}


// ------------------------------------------------------------------------- weight gradient
// dW[n][kc] += sum_m dY[m][n] * P[m][kc]; GEMM whose reduction runs over PIXELS, the slow NHWC
// dimension of both operands: tiles are staged [pixel][channel] exactly as they lie in HBM and
// the MFMA fragments (8 consecutive pixels of one channel) come out of LDS through the gfx950
// transpose read ds_read_b64_tr_b16.  Split over pixels across blockIdx.y, fp32 atomics.
struct WgradK {
  const unsigned short* x; int x_img_stride, ldx, H_in, W_in, C_in;
  const unsigned short* dy; int dy_img_stride, ld_dy;
  int C_out, S, Ktot, stride, pad_t, pad_l, W_out, HW_out, M;
  float* dw; float* db; float scale;
  unsigned x_bytes, dy_bytes;
  unsigned mg_hw, sh_hw, mg_w, sh_w;     // magic-number division by HW_out and W_out
  int H_out;
  int tiles_n, tiles_k, m_per_split;
};

__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

// ------------------------------------------------------------------------------------------
// The kernel: 128 x 128 output tile, 64 pixels per step, tiles staged by LDS-DMA into an NST-deep ring.
// One DMA wave-instruction fills 4 rows x 16 chunks; the swizzle moves to the source chunk
// (chunk = slot ^ swz(row), constant per lane because row & 3 and row bit 3 are lane constants).
// Pixel rows advance by 16 per instruction: decoded once, then stepped incrementally; 1x1 convs
// (contiguous pixel rows) take a two-instruction address path.  Fragment addresses are per-lane
// constants + immediates.
struct WgradK2 {
  WgradK b;
  int pw;            // x rows contiguous: offset = m * ldx (1x1, stride 1, no padding, dense images)
  int ydense;        // dy rows contiguous: offset = m * ld_dy
  int dbg;           // MBX_WG_DBG ablation switches (diagnosis only): 1 no DMA after the prologue, 2 no LDS reads / MFMA
};

// NG = 2: eight waves; the two 4-wave groups reduce disjoint halves of the block's pixel range into the
// SAME (n, k) tile and are summed through LDS before the atomics -> half the atomic bytes per FLOP
// (the fp32 atomics ran at the chip-wide atomic rate and cost ~35 % of the 1x1 weight gradients).
// LIN: x and dy rows are both contiguous in the pixel index (1x1 convs on dense views): offsets are m * ld, no
// (image, row, column) bookkeeping -- compile-time, like the pointwise mode of the igemm kernel.
// BIAS: the bias gradient (column sums of dy) rides along -- only the residual "up" convs have a bias.
// The block's work is given by the caller: output tile (tile_n, tile_k) and pixel range [blk_begin, blk_end) -- from
// blockIdx for the one-layer launch, from a work-item table for the grouped launch (conv_wgrad_grouped_kernel).
template <int NST, int NG, bool LIN, bool BIAS>
__device__ __forceinline__ void wgrad_wide_body(const WgradK2& q, u32x4* smem, const int tile_n, const int tile_k,
                                                const int blk_begin, const int blk_end) {
  const WgradK& p = q.b;
  constexpr int STAGE = 2 * 64 * 16;                              // smem: [NST][Y 64x16 | X 64x16] slots
  const int lane = threadIdx.x & 63;
  const int wave8 = wave_id();
  const int grp = NG == 2 ? (wave8 >> 2) : 0, wave = wave8 & 3;
  const int tid = threadIdx.x & 255;                   // thread index inside its 4-wave group
  const int wn = wave & 1, wk = wave >> 1;
  const int n0 = tile_n * 128, k0 = tile_k * 128;
  // group 0 takes the first half of the block's pixels (rounded up to 64), group 1 the rest
  const int half = NG == 2 ? ((((blk_end - blk_begin) + 1) / 2 + 63) & ~63) : (blk_end - blk_begin);
  const int m_begin = blk_begin + grp * half;
  const int m_end = NG == 2 ? (grp == 0 ? min(blk_end, blk_begin + half) : blk_end) : blk_end;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.dy, p.dy_bytes);

  // ---- DMA lane constants
  const int r4 = lane >> 4;                                   // row within the 4-row piece
  const int swz = ((r4 & 3) | (((wave >> 1) & 1) << 2)) << 1; // swz(row) for rows 16*i + 4*wave + r4
  const int lc = (lane & 15) ^ swz;                           // logical chunk this lane fetches
  const int kcol = k0 + lc * 8;
  const bool kvalid = kcol < p.Ktot;
  const int tap = (kvalid ? kcol : 0) / p.C_in, tc = (kvalid ? kcol : 0) - tap * p.C_in;
  const int tr = kvalid ? tap / p.S : (1 << 24), ts = tap - (tap / p.S) * p.S;
  const int toff = ((tr * p.W_in + ts) * p.ldx + tc) * 2;
  const int ycol = (n0 + lc * 8 < p.C_out) ? (n0 + lc * 8) * 2 : -1;
  const int ldx2 = p.ldx * 2, ldy2 = p.ld_dy * 2;
  // running pixel = next row this lane will fetch (advances by 16 per DMA instruction)
  int m_run = m_begin + wave * 4 + r4;
  int img = 0, oh = 0, ow = 0;
  {
    const unsigned mm = (unsigned)min(m_run, p.M - 1);
    img = (int)fast_div(mm, p.mg_hw, p.sh_hw);
    const int pix = (int)mm - img * p.HW_out;
    oh = (int)fast_div((unsigned)pix, p.mg_w, p.sh_w);
    ow = pix - oh * p.W_out;
  }

  // ---- fragment (transpose-read) lane constants, byte offsets inside a 16 KB tile
  const int g = lane >> 4, t = lane & 15, fq = t >> 2, pp = t & 3;
  const int lb = (8 * g + fq) * 256 + (pp & 1) * 8;
  const int sx = ((fq & 3) | ((g & 1) << 2)) << 1;
  int yo[4], xo[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    yo[a] = lb + (((2 * (wn * 4 + a) + (pp >> 1)) ^ sx) << 4);
    xo[a] = lb + (((2 * (wk * 4 + a) + (pp >> 1)) ^ sx) << 4) + 64 * 256;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = BIAS && p.db != nullptr && tile_k == 0 && tid < 128 && n0 + tid < p.C_out;

  const int nsteps = (half + 63) >> 6;                   // same trip count for both groups (barriers are block-wide)
  int st_issue = 0, st_comp = 0;
  typedef s16x4 __attribute__((address_space(3))) * lds_tr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // DMA of pixel step LT into ring slot st_issue (macro: straight-line call sites, see igemm)
#define MBX_ISSUE_STEP()                                                                                      \
  do {                                                                                                        \
    u32x4* sp = smem + (grp * NST + st_issue) * STAGE + wave * 64;                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
      const bool mv = m_run < m_end;                                                                          \
      int xoff, yoff;                                                                                         \
      if (LIN || q.pw) xoff = (mv && kvalid) ? (m_run * ldx2 + kcol * 2) : (int)kOOB;                         \
      else {                                                                                                  \
        const int hb = oh * p.stride - p.pad_t + tr, wb = ow * p.stride - p.pad_l + ts;                       \
        const bool ok = mv && ((unsigned)hb < (unsigned)p.H_in) && ((unsigned)wb < (unsigned)p.W_in);         \
        xoff = ok ? ((img * p.x_img_stride + ((oh * p.stride - p.pad_t) * p.W_in + ow * p.stride - p.pad_l) * p.ldx) * 2 + toff) \
                  : (int)kOOB;                                                                                \
      }                                                                                                       \
      if (LIN || q.ydense) yoff = (mv && ycol >= 0) ? (m_run * ldy2 + ycol) : (int)kOOB;                      \
      else yoff = (mv && ycol >= 0) ? ((img * p.dy_img_stride + (oh * p.W_out + ow) * p.ld_dy) * 2 + ycol) : (int)kOOB; \
      glds16(yr, sp + i * 256, yoff);                                                                         \
      glds16(xr, sp + 1024 + i * 256, xoff);                                                                  \
      m_run += 16;                                                                                            \
      if (!LIN && !(q.pw && q.ydense)) {                                                                      \
        ow += 16;                                                                                             \
        while (ow >= p.W_out) { ow -= p.W_out; ++oh; }                                                        \
        while (oh >= p.H_out) { oh -= p.H_out; ++img; }                                                       \
      }                                                                                                       \
    }                                                                                                         \
    st_issue = st_issue == NST - 1 ? 0 : st_issue + 1;                                                        \
  } while (0)

  static_assert(NST == 2, "ring depth 2: one step in flight");
  if (nsteps > 0) MBX_ISSUE_STEP();
  wait_vmcnt<0>();
  raw_barrier();
  for (int it = 0; it < nsteps; ++it) {
    if (it + 1 < nsteps) MBX_ISSUE_STEP();
    {
      const char* base = reinterpret_cast<const char*>(smem + (grp * NST + st_comp) * STAGE);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 yf[4], xf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + yo[a] + kk * 8192));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + yo[a] + kk * 8192 + 1024));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          yf[a] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + xo[b] + kk * 8192));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + xo[b] + kk * 8192 + 1024));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          xf[b] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
      }
      if (do_bias) {
        const unsigned short* cY = reinterpret_cast<const unsigned short*>(base);
        const int ch = tid >> 3, e = tid & 7;
        for (int r = 0; r < 64; ++r) bsum += bf2f(cY[r * 128 + ((ch ^ wg_swz(r)) << 3) + e]);
      }
      st_comp = st_comp == NST - 1 ? 0 : st_comp + 1;
    }
    wait_vmcnt<0>();
    if (it + 1 < nsteps) {            // 2-deep ring: landing probe, as in conv_igemm3_kernel (NSTG == 2)
      lds_readback_wait(lds_readback_issue(smem + (grp * NST + (st_issue ^ 1)) * STAGE + wave * 64 + lane));
    }
    raw_barrier();
  }
#undef MBX_ISSUE_STEP
  wait_vmcnt<0>();
  if (NG == 2) {                                        // group 1 -> LDS -> group 0 (rings are idle now)
    float* xch = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) xch[((a * 4 + b) * 4 + r) * 256 + tid] = acc[a][b][r];
    }
    __syncthreads();
    if (grp == 1) {
      if (do_bias) atomicAdd(p.db + n0 + tid, bsum * p.scale);
      return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] += xch[((a * 4 + b) * 4 + r) * 256 + tid];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int nb = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int kc = k0 + wk * 64 + b * 16 + (lane & 15);
      if (kc >= p.Ktot) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nb + r < p.C_out) atomicAdd(p.dw + (size_t)(nb + r) * p.Ktot + kc, acc[a][b][r] * p.scale);
    }
  }
  if (do_bias) atomicAdd(p.db + n0 + tid, bsum * p.scale);
}

// The NARROW tile: 64 output channels x 128 filter columns per block (dy rows of 128 B in LDS, their own swizzle).
// Half the tile for the same grid means every block covers twice the pixels: half the fp32 atomics per launch, and
// channel counts like 32 / 48 / 64 / 160 / 192 / 320 / 1088 stop padding a 128-wide tile.  dy must be pixel-dense.
template <int NST, int NG, bool LIN, bool BIAS>
__device__ __forceinline__ void wgrad_narrow_body(const WgradK2& q, u32x4* smem, const int tile_n, const int tile_k,
                                                  const int blk_begin, const int blk_end) {
  const WgradK& p = q.b;
  constexpr int STAGE = 64 * 8 + 64 * 16;                         // smem: [NST][Y 64x8 | X 64x16] slots
  const int lane = threadIdx.x & 63;
  const int wave8 = wave_id();
  const int grp = NG == 2 ? (wave8 >> 2) : 0, wave = wave8 & 3;
  const int tid = threadIdx.x & 255;                   // thread index inside its 4-wave group
  const int wn = wave & 1, wk = wave >> 1;
  const int n0 = tile_n * 64, k0 = tile_k * 128;
  // group 0 takes the first half of the block's pixels (rounded up to 64), group 1 the rest
  const int half = NG == 2 ? ((((blk_end - blk_begin) + 1) / 2 + 63) & ~63) : (blk_end - blk_begin);
  const int m_begin = blk_begin + grp * half;
  const int m_end = NG == 2 ? (grp == 0 ? min(blk_end, blk_begin + half) : blk_end) : blk_end;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.dy, p.dy_bytes);

  // ---- DMA lane constants
  const int r4 = lane >> 4;                                   // row within the 4-row piece
  const int swz = ((r4 & 3) | (((wave >> 1) & 1) << 2)) << 1; // swz(row) for rows 16*i + 4*wave + r4
  const int lc = (lane & 15) ^ swz;                           // logical chunk this lane fetches
  const int kcol = k0 + lc * 8;
  const bool kvalid = kcol < p.Ktot;
  const int tap = (kvalid ? kcol : 0) / p.C_in, tc = (kvalid ? kcol : 0) - tap * p.C_in;
  const int tr = kvalid ? tap / p.S : (1 << 24), ts = tap - (tap / p.S) * p.S;
  const int toff = ((tr * p.W_in + ts) * p.ldx + tc) * 2;
  // dy rows: a DMA instruction fills 8 rows x 8 chunks; rows 32*i + 8*wave + r8; swz(row) = (bit1 | bit3 << 1) << 1
  const int r8 = lane >> 3;
  const int swzy = ((((r8 >> 1) & 1) | ((wave & 1) << 1)) << 1);
  const int lcy = (lane & 7) ^ swzy;
  const int ycol = (n0 + lcy * 8 < p.C_out) ? (n0 + lcy * 8) * 2 : -1;
  const int ldx2 = p.ldx * 2, ldy2 = p.ld_dy * 2;
  int my_run = m_begin + wave * 8 + r8;                      // next dy row of this lane (advances by 32 per instruction)
  // running pixel = next row this lane will fetch (advances by 16 per DMA instruction)
  int m_run = m_begin + wave * 4 + r4;
  int img = 0, oh = 0, ow = 0;
  {
    const unsigned mm = (unsigned)min(m_run, p.M - 1);
    img = (int)fast_div(mm, p.mg_hw, p.sh_hw);
    const int pix = (int)mm - img * p.HW_out;
    oh = (int)fast_div((unsigned)pix, p.mg_w, p.sh_w);
    ow = pix - oh * p.W_out;
  }

  // ---- fragment (transpose-read) lane constants, byte offsets inside a 16 KB tile
  const int g = lane >> 4, t = lane & 15, fq = t >> 2, pp = t & 3;
  const int lb = (8 * g + fq) * 256 + (pp & 1) * 8;
  const int sx = ((fq & 3) | ((g & 1) << 2)) << 1;
  const int lby = (8 * g + fq) * 128 + (pp & 1) * 8;          // dy tile: 128-byte rows
  const int sxy = ((((fq >> 1) & 1) | ((g & 1) << 1)) << 1);
  int yo[2], xo[4];
#pragma unroll
  for (int a = 0; a < 2; ++a) yo[a] = lby + (((2 * (wn * 2 + a) + (pp >> 1)) ^ sxy) << 4);
#pragma unroll
  for (int b = 0; b < 4; ++b) xo[b] = lb + (((2 * (wk * 4 + b) + (pp >> 1)) ^ sx) << 4) + 64 * 128;

  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = BIAS && p.db != nullptr && tile_k == 0 && tid < 64 && n0 + tid < p.C_out;

  const int nsteps = (half + 63) >> 6;                   // same trip count for both groups (barriers are block-wide)
  int st_issue = 0, st_comp = 0;
  typedef s16x4 __attribute__((address_space(3))) * lds_tr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // DMA of pixel step LT into ring slot st_issue (macro: straight-line call sites, see igemm)
#define MBX_ISSUE_STEP()                                                                                      \
  do {                                                                                                        \
    u32x4* sp = smem + (grp * NST + st_issue) * STAGE + wave * 64;                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
      const bool mv = m_run < m_end;                                                                          \
      int xoff, yoff;                                                                                         \
      if (LIN || q.pw) xoff = (mv && kvalid) ? (m_run * ldx2 + kcol * 2) : (int)kOOB;                         \
      else {                                                                                                  \
        const int hb = oh * p.stride - p.pad_t + tr, wb = ow * p.stride - p.pad_l + ts;                       \
        const bool ok = mv && ((unsigned)hb < (unsigned)p.H_in) && ((unsigned)wb < (unsigned)p.W_in);         \
        xoff = ok ? ((img * p.x_img_stride + ((oh * p.stride - p.pad_t) * p.W_in + ow * p.stride - p.pad_l) * p.ldx) * 2 + toff) \
                  : (int)kOOB;                                                                                \
      }                                                                                                       \
      if (i < 2) {                                                                                            \
        yoff = (my_run < m_end && ycol >= 0) ? (my_run * ldy2 + ycol) : (int)kOOB;                            \
        glds16(yr, sp + i * 256, yoff);                                                                       \
        my_run += 32;                                                                                         \
      }                                                                                                       \
      glds16(xr, sp + 512 + i * 256, xoff);                                                                   \
      m_run += 16;                                                                                            \
      if (!LIN && !(q.pw && q.ydense)) {                                                                      \
        ow += 16;                                                                                             \
        while (ow >= p.W_out) { ow -= p.W_out; ++oh; }                                                        \
        while (oh >= p.H_out) { oh -= p.H_out; ++img; }                                                       \
      }                                                                                                       \
    }                                                                                                         \
    st_issue = st_issue == NST - 1 ? 0 : st_issue + 1;                                                        \
  } while (0)

  static_assert(NST == 2, "ring depth 2: one step in flight");
  if (nsteps > 0) MBX_ISSUE_STEP();
  wait_vmcnt<0>();
  raw_barrier();
  for (int it = 0; it < nsteps; ++it) {
    if (it + 1 < nsteps) MBX_ISSUE_STEP();
    {
      const char* base = reinterpret_cast<const char*>(smem + (grp * NST + st_comp) * STAGE);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 yf[2], xf[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + yo[a] + kk * 4096));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + yo[a] + kk * 4096 + 512));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          yf[a] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + xo[b] + kk * 8192));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + xo[b] + kk * 8192 + 1024));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          xf[b] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
      }
      if (do_bias) {
        const unsigned short* cY = reinterpret_cast<const unsigned short*>(base);
        const int ch = tid >> 3, e = tid & 7;
        for (int r = 0; r < 64; ++r)
          bsum += bf2f(cY[r * 64 + ((ch ^ ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1)) << 3) + e]);
      }
      st_comp = st_comp == NST - 1 ? 0 : st_comp + 1;
    }
    wait_vmcnt<0>();
    if (it + 1 < nsteps) {            // 2-deep ring: landing probe, as in conv_igemm3_kernel (NSTG == 2)
      lds_readback_wait(lds_readback_issue(smem + (grp * NST + (st_issue ^ 1)) * STAGE + wave * 64 + lane));
    }
    raw_barrier();
  }
#undef MBX_ISSUE_STEP
  wait_vmcnt<0>();
  if (NG == 2) {                                        // group 1 -> LDS -> group 0 (rings are idle now)
    float* xch = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) xch[((a * 4 + b) * 4 + r) * 256 + tid] = acc[a][b][r];
    }
    __syncthreads();
    if (grp == 1) {
      if (do_bias) atomicAdd(p.db + n0 + tid, bsum * p.scale);
      return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] += xch[((a * 4 + b) * 4 + r) * 256 + tid];
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int nb = n0 + wn * 32 + a * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int kc = k0 + wk * 64 + b * 16 + (lane & 15);
      if (kc >= p.Ktot) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nb + r < p.C_out) atomicAdd(p.dw + (size_t)(nb + r) * p.Ktot + kc, acc[a][b][r] * p.scale);
    }
  }
  if (do_bias) atomicAdd(p.db + n0 + tid, bsum * p.scale);
}


// ------------------------------------------------------------------------------------------
// wgrad5: the body of the grouped launch.  SIXTEEN waves with fixed roles: waves 0-7 COMPUTE a (64 NY channels) x
// (64 NX filter columns) output tile, waves 8-15 LOAD.
//  * Roles.  Measured on the all-waves-do-both form (MBX_WG_DBG ablations of the round-2 wgrad3 body): the LDS-DMA
//    issue of a 64-pixel step (~0.5 us of the CU's address path) and its LDS reads + MFMAs (~0.8 us) ADD UP when every
//    wave does first the one and then the other behind a common barrier -- the matrix pipes idle while the block
//    issues loads.  With loader waves the two run side by side: the compute waves issue no vector-memory instruction
//    in the loop and never wait on vmcnt.  Eight loaders beat four by 9 % (a wave issues one DMA per ~100 cycles).
//  * Tile shapes.  The launch is bound by the bytes the CUs pull through the L2 -> LDS path (~45 GB/s per CU), so the
//    planner picks, per layer, the (NY, NX) that moves the fewest bytes: a stage is NY + NX sub-images of
//    [64 pixels][64 channels] (8 KB, 128-byte rows, the transposing-read swizzle of the old narrow dy tile), so
//    channel counts like 160 / 192 / 320 / 1088 and filter widths like 384 / 896 / 1120 stop padding a fixed 128 x 256
//    tile (32 % of the MFMAs and a quarter of the bytes of the step were padding).
//  * Three stages, two steps of global latency in flight (counted vmcnt in the loaders only).  Loader lw fills rows
//    8 lw .. 8 lw + 7 of every sub-image: ONE pixel row per lane and step.
//  * The bias gradient (column sums of dy) is summed by the loaders out of the landed dy sub-images.
//  * single != 0: this block is the only adder of its dw tile (the pixel range is the whole layer) -> plain stores.
#ifndef MBX_WG_LOADERS
#define MBX_WG_LOADERS 8
#endif
static_assert(MBX_WG_LOADERS == 8, "wgrad5: one 8-row piece of every sub-image per loader wave");
constexpr int kWgLoaders = 8;
constexpr int kWgStageSlots = 6 * 512;                            // 16-byte slots per stage: up to six 8 KB sub-images
__device__ __forceinline__ int wg_swz64(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

template <int NY, int NX>
__device__ __forceinline__ void wgrad5_body(const WgradK2& q, u32x4* smem, const int tile_n, const int tile_k,
                                            const int blk_begin, const int blk_end, const int single) {
  static_assert(NY + NX <= 6 && NY >= 1 && NX >= 1, "stage = at most six sub-images (48 KB)");
  const WgradK& p = q.b;
  constexpr int NSUB = NY + NX, STAGE = kWgStageSlots, NST = 3;
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();                                     // 0..15
  const int n0 = tile_n * 64 * NY, k0 = tile_k * 64 * NX;
  const int nsteps = (blk_end - blk_begin + 63) >> 6;
  const bool bias = p.db != nullptr && tile_k == 0;               // block-uniform

  if (wave >= 8) {
    // ------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 8;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.dy, p.dy_bytes);
    const int r8 = lane >> 3;                                     // this lane's row inside the 8-row piece
    const int lc = (lane & 7) ^ wg_swz64(8 * lw + r8);            // logical chunk fetched into physical slot lane & 7
    int toff[NX], tr[NX], ts[NX], ycol[NY];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int kcol = k0 + j * 64 + lc * 8;
      const bool kv = kcol < p.Ktot;
      const int kc = kv ? kcol : 0;
      const int tap = kc / p.C_in, tc = kc - tap * p.C_in;
      tr[j] = kv ? tap / p.S : (1 << 24);                         // an invalid column fails the row bound check below
      ts[j] = tap - (tap / p.S) * p.S;
      toff[j] = ((tr[j] * p.W_in + ts[j]) * p.ldx + tc) * 2;
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) ycol[j] = (n0 + j * 64 + lc * 8 < p.C_out) ? (n0 + j * 64 + lc * 8) * 2 : -1;
    const int ldy2 = p.ld_dy * 2;
    int m_run = blk_begin + 8 * lw + r8;                          // + 64 per step
    int img, oh, ow;
    {
      const unsigned mm = (unsigned)min(m_run, p.M - 1);
      img = (int)fast_div(mm, p.mg_hw, p.sh_hw);
      const int pix = (int)mm - img * p.HW_out;
      oh = (int)fast_div((unsigned)pix, p.mg_w, p.sh_w);
      ow = pix - oh * p.W_out;
    }
    int st_issue = 0, st_bias = 0;
// one pixel step's LDS-DMA in two halves (dy sub-images; x sub-images + cursor advance): the landing hand-off of the
// NEXT step (wait + read-back, see conv_igemm3_kernel) sits between them, its LDS round trip covered by the second half
#define MBX_ISSUE_STEP5_Y()                                                                                    \
  do {                                                                                                         \
    u32x4* d = smem + st_issue * STAGE + lw * 64;                                                              \
    const bool mv = m_run < blk_end;                                                                           \
    const int yb = q.ydense ? m_run * ldy2 : (img * p.dy_img_stride + (oh * p.W_out + ow) * p.ld_dy) * 2;      \
    _Pragma("unroll") for (int j = 0; j < NY; ++j)                                                             \
      glds16(yr, d + j * 512, (mv && ycol[j] >= 0) ? yb + ycol[j] : (int)kOOB);                                \
  } while (0)
#define MBX_ISSUE_STEP5_X()                                                                                    \
  do {                                                                                                         \
    u32x4* d = smem + st_issue * STAGE + lw * 64;                                                              \
    const bool mv = m_run < blk_end;                                                                           \
    const int h0 = oh * p.stride - p.pad_t, w0 = ow * p.stride - p.pad_l;                                      \
    const int rb = (img * p.x_img_stride + (h0 * p.W_in + w0) * p.ldx) * 2;                                    \
    _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                           \
      const bool ok = mv && ((unsigned)(h0 + tr[j]) < (unsigned)p.H_in) && ((unsigned)(w0 + ts[j]) < (unsigned)p.W_in); \
      glds16(xr, d + (NY + j) * 512, ok ? rb + toff[j] : (int)kOOB);                                           \
    }                                                                                                          \
    m_run += 64;                                                                                               \
    ow += 64;                                                                                                  \
    while (ow >= p.W_out) { ow -= p.W_out; ++oh; }                                                             \
    while (oh >= p.H_out) { oh -= p.H_out; ++img; }                                                            \
    st_issue = st_issue == NST - 1 ? 0 : st_issue + 1;                                                         \
  } while (0)
#define MBX_ISSUE_STEP5() do { MBX_ISSUE_STEP5_Y(); MBX_ISSUE_STEP5_X(); } while (0)

    if (nsteps > 0) MBX_ISSUE_STEP5();
    if (nsteps > 1) { MBX_ISSUE_STEP5(); wait_vmcnt<NSUB>(); } else wait_vmcnt<0>();
    int st_pub = 0;                                       // ring slot of the step published next
    lds_readback_wait(lds_readback_issue(smem + (NSUB - 1) * 512 + lw * 64 + lane));   // this lane's slot of the wave's last piece
    raw_barrier();                                        // step 0 has landed
    const int lt = threadIdx.x - 512;                     // 0..511: bias sums of slot lt (row lt >> 3) of every dy image
    float bs[NY][8];
#pragma unroll
    for (int j = 0; j < NY; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) bs[j][e] = 0.f;
    for (int it = 0; it < nsteps; ++it) {
      const bool more = it + 2 < nsteps;
      st_pub = st_pub == NST - 1 ? 0 : st_pub + 1;        // slot of step it+1
#ifndef MBX_NO_PROBE_WG
      if (more) { MBX_ISSUE_STEP5_Y(); wait_vmcnt<NY>(); } else wait_vmcnt<0>();   // step it+1 has retired (this wave's share)
      const unsigned probe = lds_readback_issue(smem + st_pub * STAGE + (NSUB - 1) * 512 + lw * 64 + lane);
      if (more) MBX_ISSUE_STEP5_X();
#else                                                 // (debug builds only: A/B of what the hand-off costs)
      const unsigned probe = 0;
      if (more) { MBX_ISSUE_STEP5(); wait_vmcnt<NSUB>(); } else wait_vmcnt<0>();
#endif
      if (bias) {                                         // dy images of the step being multiplied (landed, read-only now)
        const u32x4* img_y = smem + st_bias * STAGE;
#pragma unroll
        for (int j = 0; j < NY; ++j) {
          const u32x4 v = img_y[j * 512 + lt];
          const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { bs[j][2 * e] += bf2f(w[e] & 0xffffu); bs[j][2 * e + 1] += bf2f(w[e] >> 16); }
        }
        st_bias = st_bias == NST - 1 ? 0 : st_bias + 1;
      }
#ifndef MBX_NO_PROBE_WG
      lds_readback_wait(probe);                           // read-back returned: publish step it+1
#else
      (void)probe;
#endif
      raw_barrier();
    }
#undef MBX_ISSUE_STEP5_X
#undef MBX_ISSUE_STEP5_Y
#undef MBX_ISSUE_STEP5
    if (bias) {                                           // 64 rows per channel -> LDS -> one adder per channel
      lds_barrier();                                      // (the compute waves are past their last LDS read)
      float* red = reinterpret_cast<float*>(smem);        // [NY][64 rows][64 channels]
      const int row = lt >> 3, cg = (lt & 7) ^ wg_swz64(lt >> 3);     // logical chunk of this thread's slot
#pragma unroll
      for (int j = 0; j < NY; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(j * 64 + row) * 64 + cg * 8 + e] = bs[j][e];
      lds_barrier();
      if (lt < 64 * NY && n0 + lt < p.C_out) {
        const int j = lt >> 6, c = lt & 63;
        float t = 0.f;
        for (int r = 0; r < 64; ++r) t += red[(j * 64 + r) * 64 + c];
        atomicAdd(p.db + n0 + lt, t * p.scale);
      }
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------- compute waves
  // 2 (channels) x 4 (columns) waves; a wave owns A = 2 NY channel blocks x NX column blocks of 16 x 16
  constexpr int A = 2 * NY;
  const int wn = wave & 1, wk = wave >> 1;
  // fragment (transpose-read) lane constants, byte offsets inside a stage
  const int g = lane >> 4, t = lane & 15, fq = t >> 2, pp = t & 3;
  const int lby = (8 * g + fq) * 128 + (pp & 1) * 8;
  const int sxy = ((((fq >> 1) & 1) | ((g & 1) << 1)) << 1);
  int yo[A], xo[NX];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int nb = wn * A + a;                                    // channel block 0 .. 4 NY - 1 of the tile
    yo[a] = (nb >> 2) * 8192 + lby + (((2 * (nb & 3) + (pp >> 1)) ^ sxy) << 4);
  }
#pragma unroll
  for (int b = 0; b < NX; ++b) {
    const int kb = wk * NX + b;                                   // column block 0 .. 4 NX - 1
    xo[b] = (NY + (kb >> 2)) * 8192 + lby + (((2 * (kb & 3) + (pp >> 1)) ^ sxy) << 4);
  }
  f32x4 acc[A][NX];
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < NX; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int st_comp = 0;
  typedef s16x4 __attribute__((address_space(3))) * lds_tr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  raw_barrier();                                          // step 0 has landed
  for (int it = 0; it < nsteps; ++it) {
    if (!(q.dbg & 2)) {
      const char* base = reinterpret_cast<const char*>(smem + st_comp * STAGE);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 yf[A], xf[NX];
        // 192 x 192 tiles: 72 accumulator + 36 fragment registers leave no room for nine loop-invariant fragment
        // addresses under the 128-register budget of a 16-wave block (they spilled: the launch then needs scratch,
        // which costs a ~0.1 ms stall per step when the queue sets it up) -- recompute them from lby instead
        int lby_v = lby;
        if constexpr (A * NX >= 18) asm volatile("" : "+v"(lby_v));
        auto yoff = [&](int a) {
          if constexpr (A * NX >= 18) { const int nb = wn * A + a; return (nb >> 2) * 8192 + lby_v + (((2 * (nb & 3) + (pp >> 1)) ^ sxy) << 4); }
          else return yo[a];
        };
        auto xoff = [&](int b) {
          if constexpr (A * NX >= 18) { const int kb = wk * NX + b; return (NY + (kb >> 2)) * 8192 + lby_v + (((2 * (kb & 3) + (pp >> 1)) ^ sxy) << 4); }
          else return xo[b];
        };
#pragma unroll
        for (int a = 0; a < A; ++a) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + yoff(a) + kk * 4096));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + yoff(a) + kk * 4096 + 512));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          yf[a] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int b = 0; b < NX; ++b) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + xoff(b) + kk * 4096));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(base + xoff(b) + kk * 4096 + 512));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          xf[b] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
          for (int b = 0; b < NX; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
      }
    }
    st_comp = st_comp == NST - 1 ? 0 : st_comp + 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's LDS reads are done before the stage is reused
    raw_barrier();
  }
  if (bias) { lds_barrier(); lds_barrier(); }             // the loaders reduce the bias sums through LDS meanwhile
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const int nb = n0 + (wn * A + a) * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int b = 0; b < NX; ++b) {
      const int kc = k0 + (wk * NX + b) * 16 + (lane & 15);
      if (kc >= p.Ktot) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nb + r >= p.C_out) continue;
        float* dst = p.dw + (size_t)(nb + r) * p.Ktot + kc;
        if (single) *dst = acc[a][b][r] * p.scale;        // dw is zero before the launch and this block is its only adder
        else atomicAdd(dst, acc[a][b][r] * p.scale);
      }
    }
  }
}

template <int NST, int NG, bool LIN, bool BIAS>
__global__ void __launch_bounds__(kThreads * NG)
conv_wgrad2_kernel(const WgradK2 q) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  const WgradK& p = q.b;
  const int ntiles = p.tiles_n * p.tiles_k;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = lid % ntiles, split = lid / ntiles;
  const int blk_begin = split * p.m_per_split;
  wgrad_wide_body<NST, NG, LIN, BIAS>(q, smem, tile % p.tiles_n, tile / p.tiles_n, blk_begin, min(p.M, blk_begin + p.m_per_split));
}

template <int NST, int NG, bool LIN, bool BIAS>
__global__ void __launch_bounds__(kThreads * NG)
conv_wgrad2n_kernel(const WgradK2 q) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  const WgradK& p = q.b;
  const int ntiles = p.tiles_n * p.tiles_k;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = lid % ntiles, split = lid / ntiles;
  const int blk_begin = split * p.m_per_split;
  wgrad_narrow_body<NST, NG, LIN, BIAS>(q, smem, tile % p.tiles_n, tile / p.tiles_n, blk_begin, min(p.M, blk_begin + p.m_per_split));
}

// ------------------------------------------------------------------------------------------
// GROUPED weight gradient: ONE launch for the weight gradients of many layers (a whole backward segment).
// The weight gradient is off the critical path of the backward pass (only the optimiser consumes dW), so the engine
// keeps every layer's dy and runs them together: the grid is a table of work items (layer, output tile, pixel
// range), longest first.  With hundreds of tiles from dozens of layers in flight, a tile's pixel reduction is
// split across blocks only when it is longer than a fair share of one CU's work: most tiles have ONE adder
// (no atomic traffic to speak of, bit-reproducible), loops run for hundreds of steps instead of a dozen, and
// there is one launch tail per segment instead of one per layer.
struct WgradLayer { WgradK2 q; int narrow, lin, pad0, pad1; };
struct WgradItem { int layer, tile_n, tile_k, m_begin, m_end, single, cfg, pad2; };   // single: the only adder of its dw tile;
                                                                                         // cfg: tile shape, index into kWgCfgs
constexpr int kWgradLds = 3 * 3 * 16384;               // wgrad5: three stages x up to six 8 KB sub-images (NY dy + NX x) = 144 KB
constexpr int kQueues = 8;                             // one work queue per XCD
constexpr int kCounterStride = 32;                     // ints: every queue head on its own 128-byte line

// PERSISTENT: one 1024-thread block per CU (8 MFMA + 8 loader waves, 144 KB + 16 B of LDS); a block reads the id of the XCD it runs on and pulls items from THAT
// XCD's queue (one returning atomic per item), then steals from the other queues.  The host deals whole panel groups
// -- all output tiles of one (layer, pixel range), which stream the same dy / x rows -- to one queue, so the ~32
// blocks that share an L2 walk the same pixels at the same time and all but the first read of a row is an L2 hit
// (LDS-DMA from the XCD's L2 runs at twice the rate of the Infinity Cache, MI355X_MICROARCH.md "Indexed rows").
// Placement is for speed only: any block may run any item.
// tile shapes (NY, NX) the grouped launch is instantiated for: index = WgradItem.cfg
constexpr int kWgCfgs[][2] = {{1, 4}, {1, 5}, {2, 2}, {2, 3}, {2, 4}, {3, 2}, {3, 3}};
constexpr int kNumWgCfgs = sizeof(kWgCfgs) / sizeof(kWgCfgs[0]);

__global__ void __launch_bounds__(64 * (8 + kWgLoaders))
conv_wgrad_grouped_kernel(const WgradLayer* __restrict__ layers, const WgradItem* __restrict__ items,
                          const int* __restrict__ qrange /*[2][kQueues]: begin | end*/, int* __restrict__ heads) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  volatile int* s_idx = reinterpret_cast<volatile int*>(smem + kWgradLds / 16);     // one word past the rings
  const int xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & (kQueues - 1);         // HW_REG_XCC_ID[3:0]
  int n_done = 0;                                        // work items this block has processed (block-uniform)
  for (int q = 0; q < kQueues; ++q) {
    const int qx = (xcd + q) & (kQueues - 1);
    const int qb = qrange[qx], qe = qrange[kQueues + qx];
    if (qb >= qe) continue;
    for (;;) {
      lds_barrier();                                     // everyone is done with the previous item (LDS, s_idx)
      // dequeue by a LOADER lane: its wave has nothing in flight, while a compute wave's returning atomic would queue
      // behind the dw atomics it has just fired (vmcnt retires in order)
      if (threadIdx.x == 512) *s_idx = qb + atomicAdd(heads + qx * kCounterStride, 1);
      lds_barrier();
      const int idx = __builtin_amdgcn_readfirstlane(*s_idx);
      if (idx >= qe) break;
      const WgradItem it = items[idx];                   // block-uniform: scalar loads
      const WgradK2 q2 = layers[it.layer].q;
#define MBX_WG_CASE(I)                                                                                         \
      case I: wgrad5_body<kWgCfgs[I][0], kWgCfgs[I][1]>(q2, smem, it.tile_n, it.tile_k, it.m_begin, it.m_end, it.single); break;
      switch (it.cfg) {
        MBX_WG_CASE(0) MBX_WG_CASE(1) MBX_WG_CASE(2) MBX_WG_CASE(3) MBX_WG_CASE(4) MBX_WG_CASE(5) MBX_WG_CASE(6)
        default: break;
      }
#undef MBX_WG_CASE
      ++n_done;
    }
  }
  // Leave the queue heads at zero for the next launch: the LAST block to get here (every other block has left its
  // dequeue loops) resets them.  (A hipMemsetAsync in front of the kernel did this first; as a memset NODE of a captured
  // graph replayed hundreds of times with the host far ahead it ended in GPU memory-access faults, 4 of 10 400-step
  // runs -- the launch now carries no memset.)
  // The tally line behind the counters is never reset: [0] work items processed, [1] launches completed, both cumulative.
  // The host checks items == launches x n_items at its health checks (ops.WgradGroup.completed_ok): a launch that found
  // stale heads (after an aborted launch) would otherwise compute nothing -- dW all zeros -- without any error.
  lds_barrier();
  if (threadIdx.x == 0) {
    unsigned long long* tally = reinterpret_cast<unsigned long long*>(heads + (kQueues + 1) * kCounterStride);
    if (n_done) atomicAdd(tally, (unsigned long long)n_done);
    int* exits = heads + kQueues * kCounterStride;
    if (atomicAdd(exits, 1) == (int)gridDim.x - 1) {
      for (int q = 0; q < kQueues; ++q) atomicExch(heads + q * kCounterStride, 0);
      atomicExch(exits, 0);
      atomicAdd(tally + 1, 1ull);
    }
  }
}

// ------------------------------------------------------------------------------------------ split-K reduce
// Long-K convolutions with few output tiles (the 3x3 head convolutions on the 1536-channel feature map: K = 13 824, at most
// 64 tiles of 128 x 64 on 256 CUs) run as `ksplit` K slices per tile -- float32 partial tiles [slice][M][ldp] from the
// float32-store instantiation of conv_igemm3_kernel -- and this kernel adds the slices IN SLICE ORDER (deterministic),
// rounds to bf16, stores y and writes the batch-norm statistics partials of the stored values (one row of [C][2] per 64
// pixels), i.e. everything the EV = 1 epilogue does.  One workgroup per 64 pixels; lane = one 4-channel group of one row.
constexpr int kSplitRows = 16;                        // pixels per workgroup of the reduce launch = per statistics row
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ part, int ksplit, long long slice_stride, int M, int C, int ldp,
                     unsigned short* __restrict__ y, int HW_out, long long y_img_stride, int ldy, float* __restrict__ stats,
                     int stats_mod, int stats_ld, float stats_cap) {
  __shared__ float red[kSplitRows][256][2];             // [row][channel][value | value^2] of the STORED values (C <= 256 per pass)
  const int m0 = blockIdx.x * kSplitRows, tid = threadIdx.x;
  for (int cb = 0; cb < C; cb += 256) {
    const int cw = min(256, C - cb), c4 = (cw + 3) >> 2;     // 4-channel groups of this pass
    if (cb) __syncthreads();                            // (red is reused per pass)
    for (int item = tid; item < kSplitRows * c4; item += 256) {
      const int r = item / c4, gq = item - r * c4, m = m0 + r, c = cb + 4 * gq;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        const float* src = part + (size_t)m * ldp + c;
        a = *reinterpret_cast<const float4*>(src);
        int sl = 1;
        for (; sl + 3 < ksplit; sl += 4) {              // four slices' loads in flight, added in slice order
          const float4 b0 = *reinterpret_cast<const float4*>(src + (sl + 0) * slice_stride);
          const float4 b1 = *reinterpret_cast<const float4*>(src + (sl + 1) * slice_stride);
          const float4 b2 = *reinterpret_cast<const float4*>(src + (sl + 2) * slice_stride);
          const float4 b3 = *reinterpret_cast<const float4*>(src + (sl + 3) * slice_stride);
          a.x = (((a.x + b0.x) + b1.x) + b2.x) + b3.x; a.y = (((a.y + b0.y) + b1.y) + b2.y) + b3.y;
          a.z = (((a.z + b0.z) + b1.z) + b2.z) + b3.z; a.w = (((a.w + b0.w) + b1.w) + b2.w) + b3.w;
        }
        for (; sl < ksplit; ++sl) {
          const float4 b = *reinterpret_cast<const float4*>(src + sl * slice_stride);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
      }
      const float v[4] = {a.x, a.y, a.z, a.w};
      const unsigned q2[2] = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      const unsigned q[4] = {q2[0] & 0xffffu, q2[0] >> 16, q2[1] & 0xffffu, q2[1] >> 16};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float f = (m < M && c + j < C) ? bf2f((unsigned short)q[j]) : 0.f;
        if (4 * gq + j < 256) { red[r][4 * gq + j][0] = f; red[r][4 * gq + j][1] = f * f; }
      }
      if (m < M) {
        const int img = m / HW_out, pix = m - img * HW_out;
        unsigned short* dst = y + img * y_img_stride + (long long)pix * ldy + c;
        if (c + 3 < C) *reinterpret_cast<u32x2*>(dst) = u32x2{q2[0], q2[1]};
        else for (int j = 0; j < 4; ++j) if (c + j < C) dst[j] = (unsigned short)q[j];
      }
    }
    if (stats) {
      __syncthreads();
      if (tid < cw) {
        float x1 = 0.f, x2 = 0.f;
#pragma unroll
        for (int r = 0; r < kSplitRows; ++r) { x1 += red[r][tid][0]; x2 += red[r][tid][1]; }
        if (stats_mod) {                              // (fixed-point integer adds: stats_write, conv_common.h)
          unsigned long long* o = reinterpret_cast<unsigned long long*>(stats) + ((size_t)((int)blockIdx.x % stats_mod) * stats_ld + cb + tid) * 2;
          stats_add_fixed(o, x1, x2, stats_cap);
        } else {
          float* o = stats + ((size_t)blockIdx.x * stats_ld + cb + tid) * 2;
          o[0] = x1;
          o[1] = x2;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------- host side
struct TileCfg { int BM, BN; float eff; };
const TileCfg kCfgs[] = {{128, 128, 1.00f}, {128, 64, 0.85f}, {64, 128, 0.85f}, {128, 32, 0.60f}, {64, 64, 0.70f},
                          {256, 128, 1.30f}, {128, 128, 1.10f}, {256, 64, 1.0f},    // 5-7: eight-wave blocks
                          {128, 128, 1.10f},                                       // 8: eight waves, 2-deep ring, two blocks per CU
                          {128, 64, 0.85f}, {64, 64, 0.70f},                       // 9, 10: four waves, 2-deep ring (3+ blocks per CU)
                          {128, 64, 0.85f}, {256, 64, 1.0f}, {128, 128, 1.0f}};    // 11-13: 2-deep: 8w 128x64, 8w 256x64, 4w 128x128

int pick_cfg(long M, int C_out) {
  int best = 0;
  double best_t = 1e300;
  for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i) {
    if (i >= 5) continue;                        // eight-wave tiles are opt-in (MBX_FORCE_CFG / chooser below)
    const TileCfg& c = kCfgs[i];
    const long tiles = ((M + c.BM - 1) / c.BM) * ((C_out + c.BN - 1) / c.BN);
    const long lds = 3L * (c.BM + c.BN) * 128;
    long per_cu = (160 * 1024) / lds;
    if (per_cu > 2) per_cu = 2;
    const long rounds = (tiles + 256 * per_cu - 1) / (256 * per_cu);
    const double t = (double)rounds * c.BM * c.BN / c.eff;
    if (t < best_t) { best_t = t; best = i; }
  }
  return best;
}



constexpr int kNumCfgs = 14;
static_assert(kGbCtlWords * 4 == MBX_GRID_BARRIER_BYTES && kFbSlots == MBX_BN_BWD_SLOTS, "include/mbx.h and csrc/grid_barrier.h agree");
constexpr int kI5Flag = 32;      // mbx_conv_desc.tile_config = 32 + t: igemm5 tile t (conv5.hip), persistent launch
constexpr int kI7Cfg = 65;       // mbx_conv_desc.tile_config = 65: igemm7 (conv7.hip), persistent pointwise launch with the filter panel in LDS
constexpr int kDirectCfg = 96;   // mbx_conv_desc.tile_config = 96: the direct 3x3 launch (convd.hip)
constexpr int kDirectWCfg = 97;  // mbx_conv_desc.tile_config = 97: the whole-width direct 3x3 launch for narrow maps (convd.hip)
constexpr int kPwResCfg = 99;    // mbx_conv_desc.tile_config = 99: the pixel-resident pointwise launch for the epilogue-bound 1x1 layers (convr.hip)
constexpr int kResidentCfg = 98; // mbx_conv_desc.tile_config = 98: the resident-image launch for 1x7 / 7x1 layers on small maps (convr.hip)
constexpr int kSplitFlag = 128;  // mbx_conv_desc.tile_config = 128 + S: split-K in S slices (float32 partials + reduce launch)
constexpr int kSplitMax = 32;
// slices really used and K steps per slice for a request of S slices over nk K steps (no empty slice)
inline void splitk_geom(int nk, int S, int& ksplit, int& kps) {
  if (S > nk) S = nk;
  if (S < 1) S = 1;
  kps = (nk + S - 1) / S;
  ksplit = (nk + kps - 1) / kps;
}
int choose_cfg(long M, int C_out, int desc_cfg) {
  static int force = -2;
  if (force == -2) { const char* e = getenv("MBX_FORCE_CFG"); force = e ? atoi(e) : -1; }
  if (force >= 0) return force;
  if (desc_cfg > 0 && desc_cfg <= kNumCfgs) return desc_cfg - 1;       // the caller measured (mbx_conv_desc.tile_config)
  // measured on MI355X (tools/kbench.py, all B=64 layer shapes, plain / residual / accumulate epilogues): the
  // eight-wave 128x128 tile with the 2-deep ring (two blocks per CU: one's epilogue and prologue overlap the
  // other's loop) wins wherever its blocks fill both slots of most CUs and the 128-wide column tile is not
  // mostly padding; then the eight-wave 256x128 tile (one block per CU); otherwise the model.
  {
    const long cols = (C_out + 127) / 128, tiles8 = ((M + 127) / 128) * cols;
    if (tiles8 >= 384 && (double)C_out >= 0.6 * (double)(cols * 128)) return 8;
  }
  if (C_out >= 256) {
    const long tiles5 = ((M + 255) / 256) * ((C_out + 127) / 128);
    const long rounds5 = (tiles5 + 255) / 256;              // one 8-wave block per CU
    if (tiles5 >= 128 && (double)tiles5 / (double)(rounds5 * 256) >= 0.7) return 5;
  }
  return pick_cfg(M, C_out);
}

// mbx_conv_pair: the first pass over each descriptor runs in CAPTURE mode (ConvK.dry == 2): the launcher records what it
// would have launched instead of launching it
struct PairCapture { ConvK k; int bm, bn, wnw, wmw, nstg, ev, mode, valid; };
thread_local PairCapture g_capture;

template <int BM, int BN, int WNW, int WMW, int EV, int NSTG>
int launch_pair_inst(const ConvK& a, const ConvK& b, hipStream_t s) {
  static bool attr = false;
  const size_t lds = NSTG * (size_t)(BM + BN) * 128;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm3_pair_kernel<BM, BN, WNW, WMW, EV, 0, NSTG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int n0 = a.tiles_m * a.tiles_n, n1 = b.tiles_m * b.tiles_n;
  hipLaunchKernelGGL((conv_igemm3_pair_kernel<BM, BN, WNW, WMW, EV, 0, NSTG>), dim3(n0 + n1), dim3(64 * WNW * WMW), lds, s, a, b, n0);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

template <int BM, int BN, int WNW, int WMW, int NSTG = 3>
int launch_igemm(ConvK& k, hipStream_t s) {
  k.tiles_m = (k.M + BM - 1) / BM;
  k.tiles_n = (k.C_out + BN - 1) / BN;
  k.stats_cap = stats_cap_for(k.tiles_m);               // one add per pixel tile and channel
  if (k.dry == 2) {
    g_capture.k = k; g_capture.bm = BM; g_capture.bn = BN; g_capture.wnw = WNW; g_capture.wmw = WMW; g_capture.nstg = NSTG;
    g_capture.ev = k.epi == MBX_EPI_STORE_F32 ? 5 : k.epi == MBX_EPI_RESIDUAL ? 4 : k.epi == MBX_EPI_AFFINE ? 3
                   : k.bw_n ? 6 : k.stats ? 1 : (k.accumulate || k.skip || k.bits) ? 2 : 0;
    g_capture.mode = k.shift ? 2 : k.pw ? 1 : 0;
    g_capture.valid = 1;
    return MBX_OK;
  }
  {
    const size_t lds = NSTG * (size_t)(BM + BN) * 128;          // the ring; the epilogue works out of the accumulators
    const int ev = k.epi == MBX_EPI_STORE_F32 ? 5 : k.epi == MBX_EPI_RESIDUAL ? 4 : k.epi == MBX_EPI_AFFINE ? 3
                   : k.bw_n ? 6 : k.stats ? 1 : (k.accumulate || k.skip || k.bits) ? 2 : 0;
    static bool attr_set3[7] = {false, false, false, false, false, false, false};
#define MBX_LAUNCH_EV(EV)                                                                                     \
    case EV:                                                                                                  \
      if (!attr_set3[EV]) {                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm3_kernel<BM, BN, WNW, WMW, EV, 0, NSTG>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm3_kernel<BM, BN, WNW, WMW, EV, 1, NSTG>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        attr_set3[EV] = true;                                                                                 \
      }                                                                                                       \
      if (k.pw)                                                                                               \
        hipLaunchKernelGGL((conv_igemm3_kernel<BM, BN, WNW, WMW, EV, 1, NSTG>), dim3(k.tiles_m * k.tiles_n * (EV == 5 ? k.ksplit : 1)), \
                           dim3(64 * WNW * WMW), lds, s, k);                                                  \
      else                                                                                                    \
        hipLaunchKernelGGL((conv_igemm3_kernel<BM, BN, WNW, WMW, EV, 0, NSTG>), dim3(k.tiles_m * k.tiles_n * (EV == 5 ? k.ksplit : 1)), \
                           dim3(64 * WNW * WMW), lds, s, k);                                                  \
      break;
    static bool attr_sh[2] = {false, false};
#define MBX_LAUNCH_SH(EV, SLOT)                                                                               \
    do {                                                                                                      \
      if (!attr_sh[SLOT]) {                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm3_kernel<BM, BN, WNW, WMW, EV, 2, NSTG>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        attr_sh[SLOT] = true;                                                                                 \
      }                                                                                                       \
      hipLaunchKernelGGL((conv_igemm3_kernel<BM, BN, WNW, WMW, EV, 2, NSTG>), dim3(k.tiles_m * k.tiles_n),      \
                         dim3(64 * WNW * WMW), lds, s, k);                                                    \
    } while (0)
    if (k.dry) return (k.shift && ev != 0 && ev != 2) ? MBX_ERR_UNSUPPORTED : MBX_OK;       // (the stride-2 walk: store / accumulate only)
    if (k.shift) {                               // stride-2 data gradient: its own instantiations (store / accumulate)
      if (ev == 0) MBX_LAUNCH_SH(0, 0);
      else if (ev == 2) MBX_LAUNCH_SH(2, 1);
      else return MBX_ERR_UNSUPPORTED;
    } else
    switch (ev) {
      MBX_LAUNCH_EV(0) MBX_LAUNCH_EV(1) MBX_LAUNCH_EV(2) MBX_LAUNCH_EV(3) MBX_LAUNCH_EV(4) MBX_LAUNCH_EV(5) MBX_LAUNCH_EV(6)
    }
#undef MBX_LAUNCH_SH
#undef MBX_LAUNCH_EV
  }
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

void set_magic(unsigned d, unsigned& mg, unsigned& sh) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  sh = 31 + l;
  mg = (unsigned)((1ull << sh) / d + 1);
}

int check_desc(const mbx_conv_desc* d) {
  if (!d || !d->x || !d->N || d->N < 0) return MBX_ERR_INVALID_ARG;
  if (d->C_in <= 0 || d->C_in % 8 || d->ldx % 8 || d->ldx < d->C_in) return MBX_ERR_INVALID_ARG;
  if (d->R <= 0 || d->S <= 0 || d->C_out <= 0 || d->H_out <= 0 || d->W_out <= 0) return MBX_ERR_INVALID_ARG;
  if (d->stride != 1 && d->stride != 2) return MBX_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(d->x) & 15)) return MBX_ERR_INVALID_ARG;
  const long long xe = (long long)d->N * d->x_img_stride;
  if (xe >= (1LL << 30) || (long long)d->N * d->H_out * d->W_out >= (1LL << 31)) return MBX_ERR_UNSUPPORTED;
  return MBX_OK;
}

}  // namespace

extern "C" int mbx_conv_stats_rows(const mbx_conv_desc* d) {
  if (!d) return MBX_ERR_INVALID_ARG;
  if (d->stats_rows_mod > 0) return d->stats_rows_mod;                 // atomic mode: R rows whatever the tiles
  const long M = (long)d->N * d->H_out * d->W_out;
  if (d->tile_config > kSplitFlag) return (int)((M + kSplitRows - 1) / kSplitRows);   // split-K: the reduce launch writes a row per 16 pixels
  if (d->tile_config == kDirectCfg) return mbx_direct3_grid(d->N, d->H_out, d->W_out);        // a row per workgroup
  if (d->tile_config == kDirectWCfg) { const int g = mbx_directw_grid(d->N, d->H_out, d->W_out); return g > 0 ? g : MBX_ERR_UNSUPPORTED; }
  if (d->tile_config == kResidentCfg) return mbx_resident_rows(d->N);                        // a row per image
  if (d->tile_config == kI7Cfg) return (int)((M + 127) / 128);
  if (d->tile_config > kI5Flag) {
    const int i = d->tile_config - kI5Flag - 1;
    if (i >= mbx_i5_num_tiles) return MBX_ERR_INVALID_ARG;
    return (int)((M + mbx_i5_tiles[i][0] - 1) / mbx_i5_tiles[i][0]);
  }
  const TileCfg& c = kCfgs[choose_cfg(M, d->C_out, d->tile_config)];
  return (int)((M + c.BM - 1) / c.BM);
}

static int conv_impl(const mbx_conv_desc* d, mbx_stream_t stream, int dry);
extern "C" int mbx_conv(const mbx_conv_desc* d, mbx_stream_t stream) { return conv_impl(d, stream, 0); }
extern "C" int mbx_conv_supported(const mbx_conv_desc* d) { return conv_impl(d, nullptr, 1); }
extern "C" int mbx_conv_pair(const mbx_conv_desc* a, const mbx_conv_desc* b, mbx_stream_t stream) {
  // both descriptors through every check of mbx_conv in capture mode; one grid if they resolved to the SAME instantiation
  // of one of the pair kernels (4-wave 128x64 / 64x64 tiles, store or store + statistics epilogue, general addressing)
  PairCapture ca, cb;
  g_capture.valid = 0;
  int st = conv_impl(a, stream, 2);
  if (st != MBX_OK) return st;
  if (!g_capture.valid) return MBX_ERR_UNSUPPORTED;       // (a persistent or split-K configuration)
  ca = g_capture;
  g_capture.valid = 0;
  st = conv_impl(b, stream, 2);
  if (st != MBX_OK) return st;
  if (!g_capture.valid) return MBX_ERR_UNSUPPORTED;
  cb = g_capture;
  if (ca.bm != cb.bm || ca.bn != cb.bn || ca.wnw != cb.wnw || ca.wmw != cb.wmw || ca.nstg != cb.nstg || ca.ev != cb.ev ||
      ca.mode != 0 || cb.mode != 0 || (ca.ev != 0 && ca.ev != 1 && ca.ev != 6) || ca.wnw != 2 || ca.wmw != 2 || ca.bn != 64)
    return MBX_ERR_UNSUPPORTED;
  ca.k.dry = cb.k.dry = 0;
  hipStream_t s = mbx_s(stream);
#define MBX_PAIR(BM_, NSTG_)                                                                                      \
  if (ca.bm == BM_ && ca.nstg == NSTG_)                                                                           \
    return ca.ev == 6 ? launch_pair_inst<BM_, 64, 2, 2, 6, NSTG_>(ca.k, cb.k, s)                                    \
           : ca.ev ? launch_pair_inst<BM_, 64, 2, 2, 1, NSTG_>(ca.k, cb.k, s) : launch_pair_inst<BM_, 64, 2, 2, 0, NSTG_>(ca.k, cb.k, s);
  MBX_PAIR(128, 2) MBX_PAIR(128, 3) MBX_PAIR(64, 2) MBX_PAIR(64, 3)
#undef MBX_PAIR
  return MBX_ERR_UNSUPPORTED;
}

extern "C" size_t mbx_conv_splitk_workspace_bytes(const mbx_conv_desc* d) {
  if (!d || d->tile_config <= kSplitFlag || d->tile_config > kSplitFlag + kSplitMax) return 0;
  int ksplit, kps;
  splitk_geom((d->R * d->S * d->C_in + 63) >> 6, d->tile_config - kSplitFlag, ksplit, kps);
  return (size_t)4 * d->N * d->H_out * d->W_out * (((d->C_out + 7) / 8) * 8) * ksplit;
}

static int conv_impl(const mbx_conv_desc* d, mbx_stream_t stream, int dry) {
  int st = check_desc(d);
  if (st != MBX_OK) return st;
  if (!d->w || !d->y) return MBX_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(d->w) & 15)) return MBX_ERR_INVALID_ARG;
  const bool f32 = d->epilogue == MBX_EPI_STORE_F32;
  if (!f32 && (d->C_out % 8 || d->ldy % 8 || (reinterpret_cast<uintptr_t>(d->y) & 15))) return MBX_ERR_INVALID_ARG;
  if (d->epilogue == MBX_EPI_RESIDUAL && (!d->skip || d->ld_skip % 8 || (reinterpret_cast<uintptr_t>(d->skip) & 15)))
    return MBX_ERR_INVALID_ARG;
  if ((long long)d->N * d->y_img_stride >= (1LL << 31)) return MBX_ERR_UNSUPPORTED;
  if (d->stats_partial && (d->epilogue != MBX_EPI_STORE || d->accumulate || d->skip)) return MBX_ERR_INVALID_ARG;
  if (d->epilogue == MBX_EPI_STORE && d->skip && (d->ld_skip % 8 || (reinterpret_cast<uintptr_t>(d->skip) & 15))) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  ConvK k;
  k.x = reinterpret_cast<const unsigned short*>(d->x);
  k.x_img_stride = (int)d->x_img_stride; k.ldx = d->ldx; k.H_in = d->H_in; k.W_in = d->W_in; k.C_in = d->C_in;
  k.w = reinterpret_cast<const unsigned short*>(d->w);
  k.C_out = d->C_out; k.R = d->R; k.S = d->S; k.Ktot = d->R * d->S * d->C_in;
  k.x_bytes = (unsigned)(2 * ((long long)(d->N - 1) * d->x_img_stride + ((long long)d->H_in * d->W_in - 1) * d->ldx + d->C_in));
  k.w_bytes = (unsigned)(2LL * d->C_out * k.Ktot);
  if (d->transposed) { k.mul = 1; k.shift = d->stride == 2 ? 1 : 0; } else { k.mul = d->stride; k.shift = 0; }
  k.pad_t = d->pad_t; k.pad_l = d->pad_l; k.W_out = d->W_out; k.HW_out = d->H_out * d->W_out;
  k.M = d->N * k.HW_out;
  k.y = d->y; k.y_img_stride = (int)d->y_img_stride; k.ldy = d->ldy;
  k.epi = d->epilogue; k.relu = d->relu; k.accumulate = d->accumulate;
  k.scale = d->scale; k.shiftv = d->shift;
  k.skip = reinterpret_cast<const unsigned short*>(d->skip);
  k.skip_img_stride = (int)d->skip_img_stride; k.ld_skip = d->ld_skip; k.rscale = d->rscale;
  k.bits = nullptr; k.bits_ld = 0; k.bits_bytes = 0;
  if (d->relu_bits) {
    // sign bits of a residual output: written by RESIDUAL + relu, read as the mask of a plain STORE (see mbx.h)
    const bool wr = d->epilogue == MBX_EPI_RESIDUAL && d->relu, rd = d->epilogue == MBX_EPI_STORE && !d->skip;
    if ((!wr && !rd) || d->ld_bits % 4 || d->ld_bits < 4 * ((d->C_out + 31) / 32) || (reinterpret_cast<uintptr_t>(d->relu_bits) & 3)) return MBX_ERR_INVALID_ARG;
    if (d->stats_partial || d->bn_bwd_stats) return MBX_ERR_UNSUPPORTED;
    const long long bb = (long long)d->N * d->H_out * d->W_out * d->ld_bits;
    if (bb >= (1LL << 31)) return MBX_ERR_UNSUPPORTED;
    k.bits = reinterpret_cast<unsigned char*>(d->relu_bits); k.bits_ld = d->ld_bits; k.bits_bytes = (unsigned)bb;
  }
  k.stats = d->stats_partial;
  if (d->stats_rows_mod < 0 || d->stats_rows_mod > 1024 || d->stats_ld < 0 || (d->stats_ld && d->stats_ld < d->C_out)) return MBX_ERR_INVALID_ARG;
  k.stats_mod = d->stats_rows_mod; k.stats_ld = d->stats_ld ? d->stats_ld : d->C_out;
  k.stats_cap = 0.f;                                   // (set by the launcher that knows its adders per channel: stats_cap_for)
  k.fa = FusedApply{};
  k.fa.bar = nullptr;
  k.fb = FusedBwd{};
  k.fb.bar = nullptr;
  if (d->bn_bwd) {
    // the BN backward of the layers this data gradient feeds as the tail of the launch (fused_bn.h)
    const mbx_bn_bwd_fused* b = d->bn_bwd;
    if (!b->barrier || (reinterpret_cast<uintptr_t>(b->barrier) & 127) || b->n < 1 || b->n > 4 || b->c_begin[0] != 0) return MBX_ERR_INVALID_ARG;
    if (d->epilogue != MBX_EPI_STORE || d->accumulate || d->skip || d->relu_bits || d->stats_partial || d->bn_bwd_stats || d->bn_apply ||
        d->stride != 1 || d->C_out > 2048 || d->y_img_stride != (int64_t)d->H_out * d->W_out * d->ldy)
      return MBX_ERR_UNSUPPORTED;
    const int tc = d->tile_config;
    if (!((tc > kI5Flag && tc <= kI5Flag + 7) || tc == kDirectWCfg || tc == kResidentCfg)) return MBX_ERR_UNSUPPORTED;
    const long long Mll = (long long)d->N * d->H_out * d->W_out;
    for (int i = 0; i < 4; ++i) k.fb.cb[i] = 1 << 30;
    for (int i = 0; i < b->n; ++i) {
      if (!b->y[i] || !b->dy[i] || !b->mean[i] || !b->rstd[i] || (b->relu[i] && !b->beta[i]) || !b->acc[i] ||
          (reinterpret_cast<uintptr_t>(b->y[i]) & 15) || (reinterpret_cast<uintptr_t>(b->dy[i]) & 15) || b->ld_y[i] % 8 || b->ld_dy[i] % 8 ||
          b->c_begin[i] % 8 || b->c_begin[i] >= d->C_out || (i && b->c_begin[i] <= b->c_begin[i - 1]) || b->acc_ld[i] <= 0)
        return MBX_ERR_INVALID_ARG;
      if (Mll * b->ld_y[i] >= (1LL << 31) || Mll * b->ld_dy[i] >= (1LL << 31)) return MBX_ERR_UNSUPPORTED;
      k.fb.cb[i] = b->c_begin[i];
      k.fb.y[i] = reinterpret_cast<const unsigned short*>(b->y[i]); k.fb.ldy[i] = b->ld_y[i];
      k.fb.dy[i] = reinterpret_cast<unsigned short*>(b->dy[i]); k.fb.lddy[i] = b->ld_dy[i];
      k.fb.mean[i] = b->mean[i]; k.fb.rstd[i] = b->rstd[i]; k.fb.beta[i] = b->beta[i]; k.fb.dbeta[i] = b->dbeta[i];
      k.fb.acc[i] = b->acc[i]; k.fb.acc_ld[i] = b->acc_ld[i]; k.fb.relu[i] = b->relu[i];
    }
    static int fault = -1;
    if (fault < 0) { const char* e = getenv("MBX_DEBUG_BARRIER_FAULT"); fault = (e && e[0] == '3') ? 1 : 0; }   // '3': the fused BACKWARD barriers
    k.fb.n = b->n;
    k.fb.bar = reinterpret_cast<unsigned*>(b->barrier);
    static int probe = -1;
    if (probe < 0) { const char* e = getenv("MBX_FUSED_PROBE"); probe = e ? (atoi(e) & ~1) : 0; }
    k.fb.spin_limit = fault ? (1u << 10) : (1u << 22); k.fb.fault = fault | probe; k.fb.step_poison = b->step_poison;
    k.fb.inv_M = (float)(1.0 / (double)Mll);
  }
  if (d->bn_apply) {
    // the layer's BN apply as the tail of this launch (fused_bn.h): fixed-point statistics rows, plain bf16 store, rows of y
    // contiguous over the images; only the launches whose workgroups are all resident (checked per family below)
    const mbx_bn_apply_desc* b = d->bn_apply;
    if (!b->barrier || !b->a || !b->beta || !b->mean || !b->rstd || (reinterpret_cast<uintptr_t>(b->barrier) & 127) ||
        (reinterpret_cast<uintptr_t>(b->a) & 15) || b->ld_a % 8 || b->ld_a < d->C_out)
      return MBX_ERR_INVALID_ARG;
    if (!d->stats_partial || d->stats_rows_mod < 1 || d->stats_rows_mod > 16 || d->epilogue != MBX_EPI_STORE || d->accumulate || d->skip ||
        d->transposed || d->rscale != 0.f || d->C_out > 2048 || d->y_img_stride != (int64_t)d->H_out * d->W_out * d->ldy ||
        (long long)d->N * d->H_out * d->W_out * b->ld_a >= (1LL << 31))
      return MBX_ERR_UNSUPPORTED;
    const int tc = d->tile_config;
    if (!((tc > kI5Flag && tc <= kI5Flag + 7) || tc == kDirectWCfg || tc == kResidentCfg)) return MBX_ERR_UNSUPPORTED;
    static int fault = -1;
    if (fault < 0) { const char* e = getenv("MBX_DEBUG_BARRIER_FAULT"); fault = (e && e[0] == '2') ? 1 : 0; }   // '2': the FORWARD barriers
    k.fa.bar = reinterpret_cast<unsigned*>(b->barrier);
    static int probe = -1;                                  // MBX_FUSED_PROBE: timing probes of tools/fused_probe.py (WRONG results)
    if (probe < 0) { const char* e = getenv("MBX_FUSED_PROBE"); probe = e ? (atoi(e) & ~1) : 0; }
    k.fa.spin_limit = fault ? (1u << 10) : (1u << 22); k.fa.fault = fault | probe; k.fa.step_poison = b->step_poison;
    k.fa.a = reinterpret_cast<unsigned short*>(b->a); k.fa.ld_a = b->ld_a; k.fa.beta = b->beta;
    k.fa.mean = b->mean; k.fa.rstd = b->rstd; k.fa.mmean = b->moving_mean; k.fa.mvar = b->moving_var; k.fa.thr = b->relu_thr;
    k.fa.relu = b->relu; k.fa.eps = b->eps; k.fa.decay = b->decay;
    k.fa.inv_count = 1.0 / (double)((long long)d->N * d->H_out * d->W_out);
  }
  k.bw_n = 0; k.bw_mod = 1;
  for (int i = 0; i < 4; ++i) { k.bw_cb[i] = 1 << 30; k.bw_ldy[i] = 0; k.bw_sld[i] = 0; k.bw_y[i] = nullptr; k.bw_thr[i] = nullptr; k.bw_stats[i] = nullptr; }
  if (d->bn_bwd_stats) {
    // BN-backward statistics epilogue: plain bf16 store only (a scale is fine), never with the forward statistics
    const mbx_bn_bwd_stats* b = d->bn_bwd_stats;
    if (d->epilogue != MBX_EPI_STORE || d->accumulate || d->skip || d->stats_partial || b->n < 1 || b->n > 4 || b->rows_mod < 1 ||
        b->rows_mod > 1024 || b->c_begin[0] != 0)
      return MBX_ERR_INVALID_ARG;
    for (int i = 0; i < b->n; ++i) {
      if (!b->y[i] || !b->relu_thr[i] || !b->stats[i] || (reinterpret_cast<uintptr_t>(b->y[i]) & 15) || b->ld_y[i] % 8 || b->c_begin[i] % 32 ||
          b->c_begin[i] >= d->C_out || (i && b->c_begin[i] <= b->c_begin[i - 1]) || b->stats_ld[i] <= 0)
        return MBX_ERR_INVALID_ARG;
      k.bw_cb[i] = b->c_begin[i]; k.bw_ldy[i] = b->ld_y[i]; k.bw_sld[i] = b->stats_ld[i];
      k.bw_y[i] = reinterpret_cast<const unsigned short*>(b->y[i]); k.bw_thr[i] = b->relu_thr[i]; k.bw_stats[i] = b->stats[i];
    }
    k.bw_n = b->n; k.bw_mod = b->rows_mod;
  }
  if (d->accumulate && d->acc_src) {
    if (d->ld_acc % 8 || (reinterpret_cast<uintptr_t>(d->acc_src) & 15) || (long long)d->N * d->acc_img_stride >= (1LL << 31))
      return MBX_ERR_INVALID_ARG;
    k.acc_src = reinterpret_cast<const unsigned short*>(d->acc_src); k.acc_img_stride = (int)d->acc_img_stride; k.ld_acc = d->ld_acc;
  } else {
    k.acc_src = reinterpret_cast<const unsigned short*>(d->y); k.acc_img_stride = (int)d->y_img_stride; k.ld_acc = d->ldy;
  }
  {  // ranges of the epilogue's buffer descriptors: last pixel of the last image + the channels of the view
    const long long last_pix = (long long)k.HW_out - 1, span = ((d->C_out + 7) / 8) * 8;
    const long long yb = 2 * ((long long)(d->N - 1) * d->y_img_stride + last_pix * d->ldy + span);
    const long long sb = d->skip ? 2 * ((long long)(d->N - 1) * d->skip_img_stride + last_pix * d->ld_skip + span) : 0;
    const long long ab = 2 * ((long long)(d->N - 1) * k.acc_img_stride + last_pix * k.ld_acc + span);
    if (!f32 && (yb >= (1LL << 31) || sb >= (1LL << 31) || ab >= (1LL << 31))) return MBX_ERR_UNSUPPORTED;
    k.y_bytes = (unsigned)yb; k.skip_bytes = (unsigned)sb; k.acc_bytes = (unsigned)ab;
  }
  set_magic((unsigned)k.HW_out, k.mg_hw, k.sh_hw);
  set_magic((unsigned)k.W_out, k.mg_w, k.sh_w);
  k.pw = (d->R == 1 && d->S == 1 && d->pad_t == 0 && d->pad_l == 0 && d->stride == 1) ? 1 : 0;
  k.skip_taps = 0; k.parity = 0;
  k.work_counter = d->work_counter;
  k.max_wg = d->max_workgroups;
  k.dry = dry;
  k.ksplit = 1; k.kps = 1 << 30; k.y_split_stride = 0;
#ifdef MBX_I5_STAMPS
  {  // debug build (MBX_BUILD_DEFS=-DMBX_I5_STAMPS): MBX_I5_STAMP_PTR = device address of 64 x 8 x 4 uint64 (tools/i5_stamps.py)
    static const unsigned long long sp = getenv("MBX_I5_STAMP_PTR") ? strtoull(getenv("MBX_I5_STAMP_PTR"), nullptr, 10) : 0ull;
    k.stamps = reinterpret_cast<unsigned long long*>(sp);
    k.dbg = getenv("MBX_I5_DBG") ? atoi(getenv("MBX_I5_DBG")) : 0;      // (read per call: tools switch it between launches)
    k.stagger = getenv("MBX_STAGGER") ? atoi(getenv("MBX_STAGGER")) : 0;
  }
#endif
  for (int c = 0; c < 4; ++c) { k.cls_m0[c] = 0; k.cls_hw[c] = 1; k.cls_w[c] = 1; }
  static int notap = -1;
  if (notap < 0) { const char* e = getenv("MBX_NO_TAP_SKIP"); notap = (e && e[0] == '1') ? 1 : 0; }
  if (k.shift && d->C_in % 64 == 0 && !notap) {  // K tiles never straddle filter taps: whole taps can be skipped
    // parity classes of the output pixels (class = (oh & 1) * 2 + (ow & 1)); an empty class has the start of the
    // next one, so pixel_class() steps over it
    int m0 = 0;
    for (int c = 0; c < 4; ++c) {
      const int hc = (d->H_out - (c >> 1) + 1) / 2, wc = (d->W_out - (c & 1) + 1) / 2;
      k.cls_m0[c] = m0;
      k.cls_hw[c] = hc * wc > 0 ? hc * wc : 1;
      k.cls_w[c] = wc > 0 ? wc : 1;
      m0 += d->N * hc * wc;
    }
    k.skip_taps = k.parity = 1;
  }
  if (k.bits && k.shift) return MBX_ERR_UNSUPPORTED;      // (the sign bits are indexed by the raster pixel index)
  hipStream_t s = mbx_s(stream);
  if (d->tile_config > kSplitFlag) {
    // split-K: forward convolutions with a bf16 store (+ statistics) epilogue only; partials in the caller's workspace
    const int S = d->tile_config - kSplitFlag;
    if (S < 2 || S > kSplitMax || d->transposed || d->bn_bwd_stats || d->relu_bits || d->epilogue != MBX_EPI_STORE || d->accumulate || d->skip || d->rscale != 0.f)
      return MBX_ERR_UNSUPPORTED;
    const int ldp = ((d->C_out + 7) / 8) * 8;
    int ksplit, kps;
    splitk_geom((k.Ktot + 63) >> 6, S, ksplit, kps);
    const long long slice = (long long)k.M * ldp;
    if (!d->splitk_ws || (reinterpret_cast<uintptr_t>(d->splitk_ws) & 15) || d->splitk_ws_bytes < (int64_t)(4 * slice * ksplit) ||
        slice * ksplit >= (1LL << 31))
      return MBX_ERR_WORKSPACE;
    ConvK g = k;
    g.epi = MBX_EPI_STORE_F32; g.stats = nullptr; g.relu = 0;
    g.y = d->splitk_ws; g.ldy = ldp; g.y_img_stride = k.HW_out * ldp;
    g.ksplit = ksplit; g.kps = kps; g.y_split_stride = slice;
    st = launch_igemm<128, 64, 2, 2>(g, s);
    if (st != MBX_OK || dry) return st;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((k.M + kSplitRows - 1) / kSplitRows), dim3(256), 0, s, reinterpret_cast<const float*>(d->splitk_ws),
                       ksplit, slice, k.M, k.C_out, ldp, reinterpret_cast<unsigned short*>(k.y), k.HW_out,
                       (long long)k.y_img_stride, k.ldy, k.stats, k.stats_mod, k.stats_ld, stats_cap_for((k.M + kSplitRows - 1) / kSplitRows));
    MBX_LAUNCH_CHECK();
    return MBX_OK;
  }
  if (d->tile_config == kDirectCfg) return mbx_launch_direct3(&k, d->N, d->H_out, s);
  if (d->tile_config == kDirectWCfg) return mbx_launch_directw(&k, d->N, d->H_out, s);
  if (d->tile_config == kResidentCfg) return mbx_launch_resident(&k, d->N, d->H_out, s);
  if (d->tile_config == kPwResCfg) return mbx_launch_pwres(&k, s);
  if (d->tile_config == kI7Cfg) return mbx_launch_igemm7(&k, s);
  if (d->tile_config > kI5Flag) return mbx_launch_igemm5(&k, d->tile_config - kI5Flag - 1, s);
  switch (choose_cfg(k.M, k.C_out, d->tile_config)) {
    case 0: return launch_igemm<128, 128, 2, 2>(k, s);
    case 1: return launch_igemm<128, 64, 2, 2>(k, s);
    case 2: return launch_igemm<64, 128, 2, 2>(k, s);
    case 3: return launch_igemm<128, 32, 1, 4>(k, s);
    case 5: return launch_igemm<256, 128, 2, 4>(k, s);
    case 6: return launch_igemm<128, 128, 2, 4>(k, s);
    case 7: return launch_igemm<256, 64, 1, 8>(k, s);
    case 8: return launch_igemm<128, 128, 2, 4, 2>(k, s);
    case 9: return launch_igemm<128, 64, 2, 2, 2>(k, s);
    case 10: return launch_igemm<64, 64, 2, 2, 2>(k, s);
    case 11: return launch_igemm<128, 64, 2, 4, 2>(k, s);
    case 12: return launch_igemm<256, 64, 1, 8, 2>(k, s);
    case 13: return launch_igemm<128, 128, 2, 2, 2>(k, s);
    default: return launch_igemm<64, 64, 2, 2>(k, s);
  }
}

// Geometry / pointers of one layer's weight gradient (shared by the one-layer launch and the grouped plan).
static int fill_wgrad(const mbx_conv_desc* d, const void* dy, int64_t dy_img_stride, int32_t ld_dy, float scale,
                      float* dw, float* db, WgradK2& k2) {
  int st = check_desc(d);
  if (st != MBX_OK) return st;
  if (!dy || !dw || d->transposed) return MBX_ERR_INVALID_ARG;
  if (ld_dy % 8 || ld_dy < ((d->C_out + 7) / 8) * 8 || (reinterpret_cast<uintptr_t>(dy) & 15)) return MBX_ERR_INVALID_ARG;
  if ((long long)d->N * dy_img_stride >= (1LL << 30)) return MBX_ERR_UNSUPPORTED;
  WgradK& k = k2.b;
  k.x = reinterpret_cast<const unsigned short*>(d->x);
  k.x_img_stride = (int)d->x_img_stride; k.ldx = d->ldx; k.H_in = d->H_in; k.W_in = d->W_in; k.C_in = d->C_in;
  k.dy = reinterpret_cast<const unsigned short*>(dy); k.dy_img_stride = (int)dy_img_stride; k.ld_dy = ld_dy;
  k.C_out = d->C_out; k.S = d->S; k.Ktot = d->R * d->S * d->C_in;
  k.stride = d->stride; k.pad_t = d->pad_t; k.pad_l = d->pad_l; k.W_out = d->W_out; k.HW_out = d->H_out * d->W_out;
  k.M = d->N * k.HW_out;
  k.dw = dw; k.db = db; k.scale = scale;
  set_magic((unsigned)k.HW_out, k.mg_hw, k.sh_hw);
  set_magic((unsigned)k.W_out, k.mg_w, k.sh_w);
  k.H_out = d->H_out;
  k.x_bytes = (unsigned)(2 * ((long long)(d->N - 1) * d->x_img_stride + ((long long)d->H_in * d->W_in - 1) * d->ldx + d->C_in));
  k.dy_bytes = (unsigned)(2 * ((long long)(d->N - 1) * dy_img_stride + ((long long)k.HW_out - 1) * ld_dy + ((d->C_out + 7) / 8) * 8));
  k.tiles_n = (k.C_out + 127) / 128;
  k.tiles_k = (k.Ktot + 127) / 128;
  k.m_per_split = k.M;
  k2.pw = (d->R == 1 && d->S == 1 && d->pad_t == 0 && d->pad_l == 0 && d->stride == 1 &&
           d->x_img_stride == (int64_t)d->H_in * d->W_in * d->ldx) ? 1 : 0;
  k2.ydense = (dy_img_stride == (int64_t)k.HW_out * ld_dy) ? 1 : 0;
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("MBX_WG_DBG"); dbg = e ? atoi(e) : 0; }
  k2.dbg = dbg;
  return MBX_OK;
}

extern "C" int mbx_conv_wgrad_scaled(const mbx_conv_desc* d, const void* dy, int64_t dy_img_stride, int32_t ld_dy,
                                     float scale, float* dw, float* db, mbx_stream_t stream) {
  WgradK2 k2;
  int st = fill_wgrad(d, dy, dy_img_stride, ld_dy, scale, dw, db, k2);
  if (st != MBX_OK) return st;
  MBX_ENTER();
  WgradK& k = k2.b;
  // tile_config 7 / 8: the narrow tile (64 output channels per block; eight waves, 256 / 192 blocks); needs dense dy
  const bool narrow = d->tile_config >= 7 && d->tile_config <= 10 && k2.ydense;
  k.tiles_n = narrow ? (k.C_out + 63) / 64 : (k.C_out + 127) / 128;
  const int tiles = k.tiles_n * k.tiles_k;
  // split the pixel reduction so that ~2 blocks per CU exist, at least 256 pixels per split
  static int env_ng = 0, env_target = 0;
  if (!env_ng) {
    const char* g1 = getenv("MBX_WGRAD_NG");
    env_ng = (g1 && g1[0] == '1') ? 1 : 2;
    const char* e = getenv("MBX_WGRAD_TARGET"); env_target = e ? atoi(e) : -1;
  }
  // mbx_conv_desc.tile_config selects the block shape for the weight gradient too: 0 library default (eight waves,
  // 256 blocks: one per CU), 1 eight waves / 256, 2 four waves / 512 (two per CU), 3 eight waves / 192, 4 four waves / 384,
  // 5 eight waves / 128, 6 eight waves / 224
  int ng = env_ng, target = env_target > 0 ? env_target : (env_ng == 1 ? 512 : 256);
  switch (d->tile_config) {
    case 1: ng = 2; target = 256; break;
    case 2: ng = 1; target = 512; break;
    case 3: ng = 2; target = 192; break;
    case 4: ng = 1; target = 384; break;
    case 5: ng = 2; target = 128; break;
    case 6: ng = 2; target = 224; break;
    case 7: ng = 2; target = 256; break;
    case 8: ng = 2; target = 192; break;
    case 9: ng = 1; target = 512; break;
    case 10: ng = 1; target = 768; break;
    case 11: ng = 2; target = 1; break;      // un-split: one block per tile, every dw element has ONE adder (bit-reproducible)
    default: break;
  }
  int splits = target / tiles;                 // floor: all blocks resident in one round
  const int max_splits = (k.M + 511) / 512;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = (k.M + splits - 1) / splits;
  mps = ((mps + 63) / 64) * 64;
  splits = (k.M + mps - 1) / mps;
  k.m_per_split = mps;
  {
    const bool lin = k2.pw && k2.ydense, bias = db != nullptr;
    const dim3 grid(tiles * splits);
    if (narrow) {
      static bool attr_n[8] = {false, false, false, false, false, false, false, false};
      constexpr int kLdsN = 2 * (64 * 8 + 64 * 16) * 16;              // per group: two stages x 24 KB
#define MBX_WGN(NGV, LINV, BIASV)                                                                                      \
      do {                                                                                                             \
        constexpr int slot = (NGV - 1) * 4 + (LINV ? 2 : 0) + (BIASV ? 1 : 0);                                         \
        if (!attr_n[slot]) {                                                                                           \
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad2n_kernel<2, NGV, LINV, BIASV>),          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, NGV * kLdsN);                         \
          attr_n[slot] = true;                                                                                         \
        }                                                                                                              \
        hipLaunchKernelGGL((conv_wgrad2n_kernel<2, NGV, LINV, BIASV>), grid, dim3(NGV * kThreads), NGV * kLdsN,        \
                           mbx_s(stream), k2);                                                                         \
      } while (0)
#define MBX_WGN_NG(NGV)                                                                                                \
      do {                                                                                                             \
        if (lin) { if (bias) MBX_WGN(NGV, true, true); else MBX_WGN(NGV, true, false); }                               \
        else { if (bias) MBX_WGN(NGV, false, true); else MBX_WGN(NGV, false, false); }                                 \
      } while (0)
      if (ng == 2) MBX_WGN_NG(2); else MBX_WGN_NG(1);
#undef MBX_WGN_NG
#undef MBX_WGN
      MBX_LAUNCH_CHECK();
      return MBX_OK;
    }
    static bool attr_set[8] = {false, false, false, false, false, false, false, false};
#define MBX_WG(NGV, LINV, BIASV)                                                                                       \
    do {                                                                                                               \
      constexpr int slot = (NGV - 1) * 4 + (LINV ? 2 : 0) + (BIASV ? 1 : 0);                                            \
      if (!attr_set[slot]) {                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad2_kernel<2, NGV, LINV, BIASV>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, NGV * 2 * 32768);                       \
        attr_set[slot] = true;                                                                                         \
      }                                                                                                                \
      hipLaunchKernelGGL((conv_wgrad2_kernel<2, NGV, LINV, BIASV>), grid, dim3(NGV * kThreads), NGV * 2 * 32768,       \
                         mbx_s(stream), k2);                                                                           \
    } while (0)
#define MBX_WG_NG(NGV)                                                                                                 \
    do {                                                                                                               \
      if (lin) { if (bias) MBX_WG(NGV, true, true); else MBX_WG(NGV, true, false); }                                   \
      else { if (bias) MBX_WG(NGV, false, true); else MBX_WG(NGV, false, false); }                                     \
    } while (0)
    if (ng == 2) MBX_WG_NG(2); else MBX_WG_NG(1);
#undef MBX_WG_NG
#undef MBX_WG
  }
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

// ------------------------------------------------------------------------------------ grouped weight gradient
// Host-side plan: [WgradLayer x n_jobs | WgradItem x n_items] image that the caller copies to device memory once.
static int plan_cus() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n;
    else
      ncu = 256;                       // no device visible (planning on a CPU-only host): MI355X
  }
  return ncu;
}

struct PlanJob { int cfg, ny, nx, tiles_n, tiles_k, steps, splits; double step_cost; };
struct PlanGroup { int job, m_begin, m_end, tiles; double len; };     // all output tiles of one (layer, pixel range)

static void plan_jobs(const mbx_wgrad_job* jobs, int n_jobs, int flags, std::vector<PlanJob>& pj) {
  // Tile shape per layer: the launch is bound by the bytes a CU pulls through the L2 -> LDS path (8 KB per sub-image and
  // 64-pixel step, ~390 cycles at the measured ~45 GB/s per CU) unless the MFMAs of the step take longer (128 NY NX
  // cycles: 4 NY NX per wave, 16 cycles each, two compute waves per SIMD) -- take the shape with the smallest total.
  // Work unit for splitting: one 64-pixel step of one tile, weighted by that cost.  A tile is split only when it is
  // longer than a quarter of a CU's fair share of the whole group (and never below 16 steps per piece): most tiles keep
  // a single adder.
  double total = 0.0;
  pj.resize(n_jobs);
  for (int j = 0; j < n_jobs; ++j) {
    const mbx_conv_desc& d = jobs[j].desc;
    const long long M = (long long)d.N * d.H_out * d.W_out;
    const int Ktot = d.R * d.S * d.C_in;
    double best = 1e300;
    static int max_cfg = -1;
    if (max_cfg < 0) { const char* e = getenv("MBX_WG_MAXCFG"); max_cfg = e ? atoi(e) : kNumWgCfgs; }   // bisecting aid
    for (int c = 0; c < kNumWgCfgs && c < max_cfg; ++c) {
      const int ny = kWgCfgs[c][0], nx = kWgCfgs[c][1];
      const int tn = (d.C_out + 64 * ny - 1) / (64 * ny), tk = (Ktot + 64 * nx - 1) / (64 * nx);
      const double dma = 390.0 * (ny + nx), mfma = 128.0 * ny * nx;
      const double step = dma > mfma ? dma : mfma;
      const double cost = (double)tn * tk * step;
      if (cost < best) { best = cost; pj[j].cfg = c; pj[j].ny = ny; pj[j].nx = nx; pj[j].tiles_n = tn; pj[j].tiles_k = tk; pj[j].step_cost = step / (390.0 * 6); }
    }
    pj[j].steps = (int)((M + 63) / 64);
    total += (double)pj[j].tiles_n * pj[j].tiles_k * pj[j].steps * pj[j].step_cost;
  }
  const double share = total / plan_cus();
  static double wdiv = 0.0;
  if (wdiv == 0.0) { const char* e = getenv("MBX_WG_WMAX_DIV"); wdiv = e ? atof(e) : 4.0; if (wdiv <= 0.0) wdiv = 4.0; }   // (A/B knob)
  double wmax = share / wdiv;
  if (wmax < 16.0) wmax = 16.0;
  for (int j = 0; j < n_jobs; ++j) {
    int splits = (flags & MBX_WGRAD_DETERMINISTIC) ? 1 : (int)((pj[j].steps * pj[j].step_cost + wmax - 1) / wmax);
    const int max_splits = (pj[j].steps + 15) / 16;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    pj[j].splits = splits;
  }
}

// Panel groups dealt to the XCD queues: longest items first (the short ones fill the tail), each group to the queue
// with the least work so far; a queue keeps that order.  MBX_WGRAD_SCATTER (A/B knob) deals single items instead.
static void plan_queues(const mbx_wgrad_job* jobs, int n_jobs, int flags, const std::vector<PlanJob>& pj,
                        std::vector<WgradItem>& items, int* qrange) {
  std::vector<PlanGroup> groups;
  for (int j = 0; j < n_jobs; ++j) {
    const mbx_conv_desc& d = jobs[j].desc;
    const long long M = (long long)d.N * d.H_out * d.W_out;
    long long mps = (M + pj[j].splits - 1) / pj[j].splits;
    mps = ((mps + 63) / 64) * 64;
    for (long long b = 0; b < M; b += mps) {
      PlanGroup g;
      g.job = j; g.m_begin = (int)b; g.m_end = (int)(b + mps < M ? b + mps : M);
      g.tiles = pj[j].tiles_n * pj[j].tiles_k;
      g.len = (double)((g.m_end - g.m_begin + 63) / 64) * pj[j].step_cost;
      groups.push_back(g);
    }
  }
  std::stable_sort(groups.begin(), groups.end(), [](const PlanGroup& a, const PlanGroup& b) {
    return a.len != b.len ? a.len > b.len : a.tiles > b.tiles;
  });
  std::vector<WgradItem> q[kQueues];
  double load[kQueues] = {0};
  int rr = 0;
  for (const PlanGroup& g : groups) {
    for (int tk = 0; tk < pj[g.job].tiles_k; ++tk)
      for (int tn = 0; tn < pj[g.job].tiles_n; ++tn) {
        int best = 0;
        if (flags & MBX_WGRAD_SCATTER) best = rr++ % kQueues;
        else if (tk == 0 && tn == 0) { for (int x = 1; x < kQueues; ++x) if (load[x] < load[best]) best = x; rr = best; }
        else best = rr;
        WgradItem I;
        memset(&I, 0, sizeof(I));
        I.layer = g.job; I.tile_n = tn; I.tile_k = tk; I.m_begin = g.m_begin; I.m_end = g.m_end;
        I.single = pj[g.job].splits == 1 ? 1 : 0;
        I.cfg = pj[g.job].cfg;
        q[best].push_back(I);
        load[best] += g.len;
      }
  }
  items.clear();
  for (int x = 0; x < kQueues; ++x) {
    qrange[x] = (int)items.size();
    items.insert(items.end(), q[x].begin(), q[x].end());
    qrange[kQueues + x] = (int)items.size();
  }
}

static size_t plan_image_bytes(int n_jobs, size_t n_items, int64_t* items_off, int64_t* queues_off, int64_t* heads_off,
                               int64_t* tally_off = nullptr) {
  size_t o = sizeof(WgradLayer) * (size_t)n_jobs;
  if (items_off) *items_off = (int64_t)o;
  o += sizeof(WgradItem) * n_items;
  o = (o + 127) / 128 * 128;
  if (queues_off) *queues_off = (int64_t)o;
  o += 128;                                            // qrange: 2 x kQueues ints
  if (heads_off) *heads_off = (int64_t)o;
  o += (size_t)(kQueues + 1) * kCounterStride * sizeof(int);            // queue heads + the exit counter, a line each
  if (tally_off) *tally_off = (int64_t)o;
  o += (size_t)kCounterStride * sizeof(int);                            // tally line: items processed | launches completed (uint64)
  return o;
}

extern "C" size_t mbx_wgrad_plan_bytes(const mbx_wgrad_job* jobs, int n_jobs, int flags) {
  if (!jobs || n_jobs <= 0) return 0;
  std::vector<PlanJob> pj;
  std::vector<WgradItem> items;
  int qrange[2 * kQueues];
  plan_jobs(jobs, n_jobs, flags, pj);
  plan_queues(jobs, n_jobs, flags, pj, items, qrange);
  return plan_image_bytes(n_jobs, items.size(), nullptr, nullptr, nullptr);
}

extern "C" int mbx_wgrad_plan(const mbx_wgrad_job* jobs, int n_jobs, int flags, void* host_image, size_t bytes,
                              mbx_wgrad_plan_info* info) {
  if (!jobs || n_jobs <= 0 || !host_image || !info) return MBX_ERR_INVALID_ARG;
  std::vector<PlanJob> pj;
  std::vector<WgradItem> items;
  int qrange[2 * kQueues];
  plan_jobs(jobs, n_jobs, flags, pj);
  plan_queues(jobs, n_jobs, flags, pj, items, qrange);
  int64_t items_off, queues_off, heads_off, tally_off;
  const size_t need = plan_image_bytes(n_jobs, items.size(), &items_off, &queues_off, &heads_off, &tally_off);
  if (bytes < need || items.size() >= (1ull << 30)) return MBX_ERR_WORKSPACE;
  memset(host_image, 0, need);
  char* base = reinterpret_cast<char*>(host_image);
  WgradLayer* layers = reinterpret_cast<WgradLayer*>(base);
  double flops = 0.0;
  for (int j = 0; j < n_jobs; ++j) {
    const mbx_wgrad_job& J = jobs[j];
    WgradLayer& L = layers[j];
    const int status = fill_wgrad(&J.desc, J.dy, J.dy_img_stride, J.ld_dy, J.scale, J.dw, J.db, L.q);
    if (status != MBX_OK) return status;
    L.narrow = 0;
    L.lin = (L.q.pw && L.q.ydense) ? 1 : 0;
    L.q.b.tiles_n = pj[j].tiles_n;
    flops += 2.0 * L.q.b.M * (double)L.q.b.C_out * L.q.b.Ktot;
  }
  memcpy(base + items_off, items.data(), sizeof(WgradItem) * items.size());
  memcpy(base + queues_off, qrange, sizeof(qrange));
  info->n_layers = n_jobs;
  info->n_items = (int32_t)items.size();
  info->layers_off = 0;
  info->items_off = items_off;
  info->queues_off = queues_off;
  info->heads_off = heads_off;
  info->tally_off = tally_off;
  info->flops = flops;
  return MBX_OK;
}

extern "C" int mbx_conv_wgrad_grouped(void* device_image, const mbx_wgrad_plan_info* info, mbx_stream_t stream) {
  return mbx_conv_wgrad_grouped_capped(device_image, info, 0, stream);
}

extern "C" int mbx_conv_wgrad_grouped_capped(void* device_image, const mbx_wgrad_plan_info* info, int max_workgroups,
                                             mbx_stream_t stream) {
  if (!device_image || !info || info->n_items <= 0 || info->n_layers <= 0) return MBX_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(device_image) & 15) return MBX_ERR_INVALID_ARG;
  MBX_ENTER();
  static bool attr = false;
  constexpr int kLds = kWgradLds + 16;                 // + the broadcast word of the item index
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_grouped_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    attr = true;
  }
  char* base = reinterpret_cast<char*>(device_image);
  hipStream_t s = mbx_s(stream);
  // (queue heads: zero in the plan image, and reset by the kernel's last block at the end of every launch)
  int blocks = info->n_items < plan_cus() ? info->n_items : plan_cus();            // one persistent block per CU
  if (max_workgroups > 0 && blocks > max_workgroups) blocks = max_workgroups;     // (the queues are drained by any number)
  hipLaunchKernelGGL(conv_wgrad_grouped_kernel, dim3(blocks), dim3(64 * (8 + kWgLoaders)), kLds, s,
                     reinterpret_cast<const WgradLayer*>(base + info->layers_off),
                     reinterpret_cast<const WgradItem*>(base + info->items_off),
                     reinterpret_cast<const int*>(base + info->queues_off),
                     reinterpret_cast<int*>(base + info->heads_off));
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

extern "C" int mbx_conv_wgrad(const mbx_conv_desc* d, const void* dy, int64_t dy_img_stride, int32_t ld_dy,
                              float* dw, float* db, mbx_stream_t stream) {
  return mbx_conv_wgrad_scaled(d, dy, dy_img_stride, ld_dy, 1.0f, dw, db, stream);
}
