// libmbx: conv_direct3_kernel -- 3x3, stride-1 convolutions with FEW channels (C_in 32 / 64, C_out 32 / 48 / 64) on LARGE maps
// (the stem layers Conv2d_2a / 2b of model.py:93-100 at 147 x 147 x BATCH_SIZE 64: 1.4 M pixels, forward and data gradient).
//
// The implicit-GEMM kernels gather the input NINE times through the L2 -> LDS path (once per filter tap: 800 MB per launch
// for 88 MB of activations) and ran these launches at 2-3x their HBM time (134 / 141 / 86 / 129 us against 36 / 53 / 36 / 53).
// Here a workgroup stages a (8 + 2) x (32 + 2) pixel PATCH of the input once (halo included, zero padding by
// buffer-descriptor range misses), keeps the whole filter in LDS for its lifetime, and multiplies the nine taps out of LDS:
// tap (r, s) of output row `oh` reads the patch row oh + r at columns shifted by s.  Round 1 tried the one-shot form of
// this (load a patch, 0.5 us of MFMA, store: a block waited on latency for most of its life) and dropped it; this one is
// PERSISTENT: one 512-thread workgroup per CU walks tiles first, first + grid, ... with the NEXT tile's patch in flight
// (LDS-DMA into the other of two patch buffers) while the current one is multiplied and stored.
//
// Same arithmetic as the implicit GEMM: K order (r, s, c), one v_mfma_f32_16x16x32_bf16 per 32 input channels of a tap,
// filter rows as the A operand (D[row = channel][col = pixel]) -- the accumulation sequence of every output element is
// the igemm kernels', so the results are bit-identical to them (tests/test_gpu_conv.py::test_conv_direct3).
// Epilogues: bf16 store (EV = 0), store + batch-norm statistics (EV = 1: ONE partial row per workgroup, summed over its
// tiles in a fixed order: mbx_conv_stats_rows = the grid size).
#include "conv_common.h"

namespace {

constexpr int kDTH = 8, kDTW = 32;                        // output tile: 8 rows x 32 columns = one row per wave
constexpr int kDPH = kDTH + 2, kDPW = kDTW + 2;           // input patch with halo
constexpr int kDThreads = 512;

struct DirK { int N, H_out, tiles_h, tiles_w, ntiles; };

// conflict-free ds_read_b128 of 16 consecutive pixels (or filter rows) of CI channels: a pixel is CI / 8 chunks of 16 B;
// 256 / (2 CI) pixels cover the 64 banks once, so the chunk is XOR-ed with the pixel index divided by that count
template <int CI>
__device__ __forceinline__ int dkey(int pix) { return CI == 32 ? ((pix >> 2) & 3) : ((pix >> 1) & 7); }

// Filter rows are fed to the MFMA in a PERMUTED order (as in conv_igemm3_kernel): LDS row 16 a + f of a tap holds output
// channel dperm(a, f), so that blocks 2A and 2A + 1 leave a lane EIGHT consecutive channels of one pixel = one 16-byte store
// (four lanes cover 64 contiguous bytes; with the plain order a 128-byte pixel row went out as four 32-byte pieces and the
// 32 -> 64 layer ran at 2.4x its HBM time).  A trailing unpaired block (C_out 48: channels 32..47) keeps the plain order.
template <int CO>
__device__ __forceinline__ int dperm(int a, int f) {
  return (a < (CO / 32) * 2) ? 32 * (a >> 1) + 8 * (f >> 2) + 4 * (a & 1) + (f & 3) : 16 * a + f;
}

template <int CI, int CO, int EV>
__global__ void __launch_bounds__(kDThreads)
conv_direct3_kernel(const ConvK p, const DirK q) {
  constexpr int C8 = CI / 8, KC = CI / 32, NA = CO / 16;
  constexpr int PCH = kDPH * kDPW * C8;                   // 16-byte chunks of a patch
  constexpr int PROUNDS = (PCH + kDThreads - 1) / kDThreads;
  constexpr int PBUF = PROUNDS * kDThreads;               // chunks per patch buffer (the tail is zero fill)
  constexpr int WCH = 9 * CO * C8;                        // chunks of the filter image [tap][co][c8]
  constexpr int WROUNDS = (WCH + kDThreads - 1) / kDThreads;
  // patch ring: NBUF buffers, the loads run D = NBUF - 1 tiles ahead (three buffers where 160 KB hold them: C_in 32).  One
  // workgroup per CU has nothing else to cover the ~2 us of a patch's HBM latency with: at D = 1 every tile waited for it.
  constexpr int NBUF = CI == 32 ? 3 : 2, D = NBUF - 1;
  constexpr int NS = 2 * (CO / 32 + (CO % 32 ? 1 : 0));   // store instructions of a tile's epilogue, per wave
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  u32x4* const wimg = smem;                               // [9][CO][C8] (swizzled)
  u32x4* const pbuf = smem + WROUNDS * kDThreads;         // the patch ring
  float* const red = reinterpret_cast<float*>(pbuf + NBUF * PBUF);  // [8 waves][CO][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);

  // this thread's patch chunks (the same positions for every tile): patch row / column, source chunk
  int prow[PROUNDS], pcol[PROUNDS], pc8[PROUNDS];
#pragma unroll
  for (int i = 0; i < PROUNDS; ++i) {
    const int ch = i * kDThreads + tid;
    const int pix = ch / C8, c8p = ch - pix * C8;
    prow[i] = pix / kDPW;
    pcol[i] = pix - prow[i] * kDPW;
    pc8[i] = c8p ^ dkey<CI>(pcol[i]);                     // the DMA destination is lane-linear: swizzle the SOURCE chunk
    if (ch >= PCH) prow[i] = -1;
  }
  auto issue_patch = [&](int t, int buf) {
    const int tw = t % q.tiles_w, r_ = t / q.tiles_w, th = r_ % q.tiles_h, img = r_ / q.tiles_h;
    const int h_in0 = th * kDTH - p.pad_t, w_in0 = tw * kDTW - p.pad_l;
    u32x4* dst = pbuf + buf * PBUF + wave * 64;
#pragma unroll
    for (int i = 0; i < PROUNDS; ++i) {
      const int h = h_in0 + prow[i], w = w_in0 + pcol[i];
      const bool ok = prow[i] >= 0 && (unsigned)h < (unsigned)p.H_in && (unsigned)w < (unsigned)p.W_in;
      glds16(xr, dst + i * kDThreads, ok ? (img * p.x_img_stride + (h * p.W_in + w) * p.ldx + pc8[i] * 8) * 2 : (int)kOOB);
    }
  };

  // ---- prologue: the filter image (once) and the first patch
#pragma unroll
  for (int i = 0; i < WROUNDS; ++i) {
    const int ch = i * kDThreads + tid;
    const int row = ch / C8, c8p = ch - row * C8;         // row = tap * CO + LDS row (16 a + f)
    const int tap = row / CO, lr = row - tap * CO;
    const int co = dperm<CO>(lr >> 4, lr & 15);           // the output channel that LDS row holds
    const int c8 = c8p ^ dkey<CI>(lr);
    const bool ok = ch < WCH && co < p.C_out;
    glds16(wr, wimg + i * kDThreads + wave * 64, ok ? ((co * 9 + tap) * CI + c8 * 8) * 2 : (int)kOOB);   // KRSC: [co][r][s][c]
  }
  const int first = (int)blockIdx.x, G = (int)gridDim.x;
  if (first < q.ntiles) issue_patch(first, 0);
  if (D > 1 && first + G < q.ntiles) { issue_patch(first + G, 1); wait_vmcnt<PROUNDS>(); } else wait_vmcnt<0>();
  lds_readback_wait(lds_readback_issue(pbuf + (PROUNDS - 1) * kDThreads + wave * 64 + lane));
  raw_barrier();

  // ---- fragment addresses (16-byte slots)
  const int frow = lane & 15, fch = lane >> 4;
  float s1[NA][4], s2[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[a][r] = 0.f; s2[a][r] = 0.f; }
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
  // EV = 3 (folded batch norm of the inference / --fine_tune forward, detect.py:313-326): y = act(acc * scale[c] + shift[c]),
  // the expression of the implicit-GEMM kernels' affine epilogue; the lane's channels are the same for every tile
  constexpr int NPq = CO / 32;
  float sc8[NPq > 0 ? NPq : 1][8], sh8[NPq > 0 ? NPq : 1][8], sc4[4], sh4[4];
  if constexpr (EV == 3) {
#pragma unroll
    for (int A = 0; A < NPq; ++A)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 32 * A + 8 * fch + j;
        sc8[A][j] = (p.scale && c < p.C_out) ? p.scale[c] : 1.f;
        sh8[A][j] = (p.shiftv && c < p.C_out) ? p.shiftv[c] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 2 * NPq * 16 + 4 * fch + j;
      sc4[j] = (p.scale && c < p.C_out) ? p.scale[c] : 1.f;
      sh4[j] = (p.shiftv && c < p.C_out) ? p.shiftv[c] : 0.f;
    }
  }

  // The filter fragments are the same for every tile: where they fit the register file (a 512-thread workgroup alone on its
  // CU may use 256 registers per lane) they are read from LDS ONCE -- per tile that leaves the two pixel fragments of a tap,
  // a third to a half of the LDS reads (the tile loop was LDS-read bound: ~3k of its ~6k cycles).
  // (as many taps as ~112 registers hold: all nine for 32 -> 32 / 48, seven for 32 -> 64 and 64 -> 32, four for 64 -> 48)
  constexpr int WT = (112 / (KC * NA * 4)) < 9 ? (112 / (KC * NA * 4)) : 9;
  bf16x8 wreg[WT][KC][NA];
#pragma unroll
  for (int tap = 0; tap < WT; ++tap)
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int lr = a * 16 + frow;
        wreg[tap][j][a] = __builtin_bit_cast(bf16x8, wimg[(tap * CO + lr) * C8 + ((4 * j + fch) ^ dkey<CI>(lr))]);
      }
  int buf = 0;
  for (int t = first; t < q.ntiles; t += G) {
    const bool more = t + G < q.ntiles;                   // a next tile exists (its patch must be published below)
    const bool ahead = t + D * G < q.ntiles;              // ... and one D tiles ahead, whose patch is issued now
    const int buf_next = buf == NBUF - 1 ? 0 : buf + 1;
    {
      int bi = buf + D;
      if (bi >= NBUF) bi -= NBUF;                         // the buffer tile t - 1 was multiplied out of (free: barrier below)
      if (ahead) issue_patch(t + D * G, bi);
    }
    const u32x4* pb = pbuf + buf * PBUF;
    f32x4 acc[NA][2];
#pragma unroll
    for (int a = 0; a < NA; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int tap = r * 3 + s;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          bf16x8 wf[NA], pf[2];
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            if (tap < WT) {                             // (folded: the tap loops are unrolled)
              wf[a] = wreg[tap < WT ? tap : 0][j][a];
            } else {
              const int lr = a * 16 + frow;
              wf[a] = __builtin_bit_cast(bf16x8, wimg[(tap * CO + lr) * C8 + ((4 * j + fch) ^ dkey<CI>(lr))]);
            }
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int pc = b * 16 + s + frow;
            pf[b] = __builtin_bit_cast(bf16x8, pb[((wave + r) * kDPW + pc) * C8 + ((4 * j + fch) ^ dkey<CI>(pc))]);
          }
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
        }
      }
    // ---- epilogue: blocks 2A, 2A + 1 -> 8 consecutive channels (32 A + 8 fch ..) of pixel (wave, 16 b + frow): 16-byte stores;
    // an unpaired last block: 4 consecutive channels, 8-byte stores
    {
      const int tw = t % q.tiles_w, r_ = t / q.tiles_w, th = r_ % q.tiles_h, img = r_ / q.tiles_h;
      const int oh = th * kDTH + wave;
      constexpr int NP = CO / 32;                           // paired blocks
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ow = tw * kDTW + b * 16 + frow;
        const bool pv = oh < q.H_out && ow < p.W_out;
        const int pix_off = img * p.y_img_stride + (oh * p.W_out + ow) * p.ldy;
#pragma unroll
        for (int A = 0; A < NP; ++A) {
          const int c0 = 32 * A + 8 * fch;
          const bool ok = pv && c0 < p.C_out;               // C_out % 8 == 0: a group of eight is all in or all out
          unsigned h8[8];
          float v8[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v8[r] = acc[2 * A][b][r]; v8[4 + r] = acc[2 * A + 1][b][r]; }
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
#pragma unroll
          for (int j = 0; j < 8; ++j) h8[j] = (j & 1) ? (h2[j >> 1] >> 16) : (h2[j >> 1] & 0xffffu);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{h2[0], h2[1], h2[2], h2[3]},
                                                 yr, ok ? (int)((pix_off + c0) * 2) : (int)kOOB, 0, 0);
          if constexpr (EV == 1) {
            if (ok) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float f0 = bf2f((unsigned short)h8[r]), f1 = bf2f((unsigned short)h8[4 + r]);
                s1[2 * A][r] += f0; s2[2 * A][r] += f0 * f0;
                s1[2 * A + 1][r] += f1; s2[2 * A + 1][r] += f1 * f1;
              }
            }
          }
        }
        if constexpr (NA > 2 * NP) {                        // C_out 48: channels 32 .. 47 in the plain order
          constexpr int a = 2 * NP;
          const int c0 = a * 16 + 4 * fch;
          const bool ok = pv && c0 < p.C_out;
          unsigned h4[4];
          float v4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[a][b][r];
            if constexpr (EV == 3) { v = v * sc4[r] + sh4[r]; if (p.relu) v = relu_f(v); }
            v4[r] = v;
          }
          const unsigned g2[2] = {pack2bf(v4[0], v4[1]), pack2bf(v4[2], v4[3])};
          h4[0] = g2[0] & 0xffffu; h4[1] = g2[0] >> 16; h4[2] = g2[1] & 0xffffu; h4[3] = g2[1] >> 16;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{g2[0], g2[1]}, yr,
                                                ok ? (int)((pix_off + c0) * 2) : (int)kOOB, 0, 0);
          if constexpr (EV == 1) {
            if (ok) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float f = bf2f((unsigned short)h4[r]); s1[a][r] += f; s2[a][r] += f * f; }
            }
          }
        }
      }
    }
    if (more) {
      // the NEXT tile's patch: retired, then visible (landing read-back).  Behind its DMA this wave has issued D - 1 newer
      // patches and D tiles' stores (memory operations retire in order): a counted wait leaves those in flight -- except in
      // the tail, where fewer were issued and the count would let the patch itself through
      // -- and in a workgroup's FIRST tile when D > 1: its next patch was issued by the prologue, with no earlier tile's stores
      // behind it, so the steady-state count would leave NS of that patch's own pieces un-retired at the publishing barrier
      // (found by the saturation stress test, round 4: bit-identical 2 of 3 times, never wrong on a quiet chip)
      if (ahead && D > 1 && t == first) wait_vmcnt<(D - 1) * PROUNDS + (D - 1) * NS>();
      else if (ahead) wait_vmcnt<(D - 1) * PROUNDS + D * NS>();
      else wait_vmcnt<0>();
      lds_readback_wait(lds_readback_issue(pbuf + buf_next * PBUF + (PROUNDS - 1) * kDThreads + wave * 64 + lane));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of the current patch are done
    raw_barrier();                                        // everyone is: the buffer may be refilled, the next one read
    buf = buf_next;
  }
  if constexpr (EV == 1) {
    // ONE statistics row per workgroup: lane sums -> the 16 lanes that share its channels -> the eight waves, fixed order
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x1 = row_sum16(s1[a][r]), x2 = row_sum16(s2[a][r]);
        if (frow == 0) {
          const int ch = dperm<CO>(a, 4 * fch + r);           // the channel MFMA row 4 fch + r of block a holds
          red[(wave * CO + ch) * 2] = x1;
          red[(wave * CO + ch) * 2 + 1] = x2;
        }
      }
    __syncthreads();
    if (tid < CO && tid < p.C_out) {
      float x1 = 0.f, x2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { x1 += red[(w * CO + tid) * 2]; x2 += red[(w * CO + tid) * 2 + 1]; }
      stats_write(p, (int)blockIdx.x, tid, x1, x2);
    }
  }
}

// ------------------------------------------------------------------------------------------ the network's first layer
// conv_stem_kernel: 3x3, STRIDE 2, C_in = 8 (the packed RGB input: three real channels), C_out <= 32 -- Conv2d_1a_3x3 of
// model.py:90 (299 x 299 -> 149 x 149).  As an implicit GEMM its K is 72 (two 64-deep K tiles, the second 7/8 padding) and
// every output pixel gathers nine 16-byte taps through the L2 -> LDS path: 105 us per step at BATCH_SIZE 64 against 36 us of
// HBM time (59 TFLOP/s), 282 us at the detect batch.  Same scheme as conv_direct3_kernel: a persistent workgroup stages the
// (2 * 8 + 1) x (2 * 32 + 1) input pixels of an 8 x 32 output tile once (one 16-byte chunk per pixel), keeps the filter in LDS
// and multiplies out of LDS: one v_mfma_f32_16x16x32_bf16 per FOUR taps (lane group fch holds tap 4 j + fch of its pixel) --
// K elements in the implicit GEMM's order and grouping (taps 0-3, 4-7, 8 + zeros), so the results are bit-identical to it.
constexpr int kSPH = 2 * kDTH + 1, kSPW = 2 * kDTW + 1;   // input patch of an output tile (stride 2, 3 x 3)
constexpr int kSWRow = 13;                                // filter image row: 12 chunks of K (9 taps + 3 of zeros) padded to 13 (conflict-free reads)

template <int EV>
__global__ void __launch_bounds__(kDThreads)
conv_stem_kernel(const ConvK p, const DirK q) {
  constexpr int CO = 32, NA = 2;
  constexpr int PCH = kSPH * kSPW;                        // 16-byte chunks of a patch (one per pixel)
  constexpr int PROUNDS = (PCH + kDThreads - 1) / kDThreads;
  constexpr int PBUF = PROUNDS * kDThreads;
  constexpr int WCH = CO * kSWRow;
  static_assert(WCH <= kDThreads, "filter image in one round");
  constexpr int NBUF = 3, D = NBUF - 1;
  constexpr int NS = 2;                                   // store instructions of a tile's epilogue, per wave
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  u32x4* const wimg = smem;                               // [CO][kSWRow]
  u32x4* const pbuf = smem + kDThreads;                   // the patch ring
  float* const red = reinterpret_cast<float*>(pbuf + NBUF * PBUF);  // [8 waves][CO][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  int prow[PROUNDS], pcol[PROUNDS];
#pragma unroll
  for (int i = 0; i < PROUNDS; ++i) {
    const int ch = i * kDThreads + tid;
    prow[i] = ch / kSPW;
    pcol[i] = ch - prow[i] * kSPW;
    if (ch >= PCH) prow[i] = -1;
  }
  auto issue_patch = [&](int t, int buf) {
    const int tw = t % q.tiles_w, r_ = t / q.tiles_w, th = r_ % q.tiles_h, img = r_ / q.tiles_h;
    const int h_in0 = 2 * th * kDTH - p.pad_t, w_in0 = 2 * tw * kDTW - p.pad_l;
    u32x4* dst = pbuf + buf * PBUF + wave * 64;
#pragma unroll
    for (int i = 0; i < PROUNDS; ++i) {
      const int h = h_in0 + prow[i], w = w_in0 + pcol[i];
      const bool ok = prow[i] >= 0 && (unsigned)h < (unsigned)p.H_in && (unsigned)w < (unsigned)p.W_in;
      glds16(xr, dst + i * kDThreads, ok ? (img * p.x_img_stride + (h * p.W_in + w) * p.ldx) * 2 : (int)kOOB);
    }
  };
  {  // the filter image: LDS row 16 a + f holds output channel dperm(a, f); chunk c of a row = tap c (KRSC with C = 8), 9.. zeros
    const int row = tid / kSWRow, c = tid - row * kSWRow;
    const int co = dperm<CO>(row >> 4, row & 15);
    const bool ok = tid < WCH && c < 9 && co < p.C_out;
    glds16(wr, wimg + wave * 64, ok ? (co * 9 + c) * 16 : (int)kOOB);
  }
  const int first = (int)blockIdx.x, G = (int)gridDim.x;
  if (first < q.ntiles) issue_patch(first, 0);
  if (first + G < q.ntiles) { issue_patch(first + G, 1); wait_vmcnt<PROUNDS>(); } else wait_vmcnt<0>();
  lds_readback_wait(lds_readback_issue(pbuf + (PROUNDS - 1) * kDThreads + wave * 64 + lane));
  raw_barrier();

  const int frow = lane & 15, fch = lane >> 4;
  float s1[NA][4], s2[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[a][r] = 0.f; s2[a][r] = 0.f; }
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
  float sc8[8], sh8[8];
  if constexpr (EV == 3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 8 * fch + j;
      sc8[j] = (p.scale && c < p.C_out) ? p.scale[c] : 1.f;
      sh8[j] = (p.shiftv && c < p.C_out) ? p.shiftv[c] : 0.f;
    }
  }
  // this lane's patch slots: K step j multiplies tap min(4 j + fch, 8) (taps past 8 meet the filter image's zero chunks) of
  // output pixels (wave, 16 b + frow): input pixel (2 wave + r, 2 (16 b + frow) + s)
  int poff[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = min(4 * j + fch, 8), r = tap / 3, sx = tap - 3 * r;
#pragma unroll
    for (int b = 0; b < 2; ++b) poff[j][b] = (2 * wave + r) * kSPW + 2 * (16 * b + frow) + sx;
  }
  bf16x8 wreg[3][NA];                                     // the filter fragments: the same for every tile
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int a = 0; a < NA; ++a) wreg[j][a] = __builtin_bit_cast(bf16x8, wimg[(a * 16 + frow) * kSWRow + 4 * j + fch]);
  int buf = 0;
  for (int t = first; t < q.ntiles; t += G) {
    const bool more = t + G < q.ntiles;
    const bool ahead = t + D * G < q.ntiles;
    const int buf_next = buf == NBUF - 1 ? 0 : buf + 1;
    {
      int bi = buf + D;
      if (bi >= NBUF) bi -= NBUF;
      if (ahead) issue_patch(t + D * G, bi);
    }
    const u32x4* pb = pbuf + buf * PBUF;
    f32x4 acc[NA][2];
#pragma unroll
    for (int a = 0; a < NA; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bf16x8 wf[NA], pf[2];
#pragma unroll
      for (int a = 0; a < NA; ++a) wf[a] = wreg[j][a];
#pragma unroll
      for (int b = 0; b < 2; ++b) pf[b] = __builtin_bit_cast(bf16x8, pb[poff[j][b]]);
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
    }
    {
      const int tw = t % q.tiles_w, r_ = t / q.tiles_w, th = r_ % q.tiles_h, img = r_ / q.tiles_h;
      const int oh = th * kDTH + wave;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ow = tw * kDTW + b * 16 + frow;
        const int c0 = 8 * fch;
        const bool ok = oh < q.H_out && ow < p.W_out && c0 < p.C_out;
        const int pix_off = img * p.y_img_stride + (oh * p.W_out + ow) * p.ldy;
        unsigned h8[8];
        float v8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v8[r] = acc[0][b][r]; v8[4 + r] = acc[1][b][r]; }
        if constexpr (EV == 3) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = v8[j] * sc8[j] + sh8[j];
          if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = relu_f(v8[j]);
          }
        }
        unsigned h2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h2[j] = pack2bf(v8[2 * j], v8[2 * j + 1]);
#pragma unroll
        for (int j = 0; j < 8; ++j) h8[j] = (j & 1) ? (h2[j >> 1] >> 16) : (h2[j >> 1] & 0xffffu);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{h2[0], h2[1], h2[2], h2[3]},
                                               yr, ok ? (int)((pix_off + c0) * 2) : (int)kOOB, 0, 0);
        if constexpr (EV == 1) {
          if (ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float f0 = bf2f((unsigned short)h8[r]), f1 = bf2f((unsigned short)h8[4 + r]);
              s1[0][r] += f0; s2[0][r] += f0 * f0;
              s1[1][r] += f1; s2[1][r] += f1 * f1;
            }
          }
        }
      }
    }
    if (more) {
      // (the counted waits of conv_direct3_kernel: D - 1 newer patches and D tiles' stores behind the next patch -- one tile's
      // stores fewer in the workgroup's first tile, whose next patch the prologue issued)
      if (ahead && t == first) wait_vmcnt<(D - 1) * PROUNDS + (D - 1) * NS>();
      else if (ahead) wait_vmcnt<(D - 1) * PROUNDS + D * NS>();
      else wait_vmcnt<0>();
      lds_readback_wait(lds_readback_issue(pbuf + buf_next * PBUF + (PROUNDS - 1) * kDThreads + wave * 64 + lane));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
    buf = buf_next;
  }
  if constexpr (EV == 1) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x1 = row_sum16(s1[a][r]), x2 = row_sum16(s2[a][r]);
        if (frow == 0) {
          const int ch = dperm<CO>(a, 4 * fch + r);
          red[(wave * CO + ch) * 2] = x1;
          red[(wave * CO + ch) * 2 + 1] = x2;
        }
      }
    __syncthreads();
    if (tid < CO && tid < p.C_out) {
      float x1 = 0.f, x2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { x1 += red[(w * CO + tid) * 2]; x2 += red[(w * CO + tid) * 2 + 1]; }
      stats_write(p, (int)blockIdx.x, tid, x1, x2);
    }
  }
}

constexpr int stem_lds() {
  constexpr int PCH = kSPH * kSPW, PBUF = (PCH + kDThreads - 1) / kDThreads * kDThreads;
  return (kDThreads + 3 * PBUF) * 16 + 8 * 32 * 2 * 4;
}

int launch_stem(const ConvK& k, const DirK& q, int grid, hipStream_t s) {
  constexpr int lds = stem_lds();
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr[3] = {false, false, false};
  const int ev = k.epi == MBX_EPI_AFFINE ? 2 : k.stats ? 1 : 0;
  if (!attr[ev]) {
    if (ev == 2) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if (ev) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr[ev] = true;
  }
  if (ev == 2) hipLaunchKernelGGL((conv_stem_kernel<3>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  else if (ev) hipLaunchKernelGGL((conv_stem_kernel<1>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  else hipLaunchKernelGGL((conv_stem_kernel<0>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

// ------------------------------------------------------------------------------------------ whole-width tiles
// conv_directw_kernel: the same direct scheme for NARROW maps (block35's 35 x 35: three 3x3 convolutions per block, forward
// and data gradient, C_in 32 / 48 / 64, C_out 32 / 48 / 64 -- 0.7 ms of the training step at 170-300 TFLOP/s as implicit
// GEMMs, whose 16-byte gather pieces are 64-96 byte runs of a 480-byte pixel: nine taps' worth of badly coalesced L2 -> LDS
// traffic).  conv_direct3_kernel's 8 x 32 tiles fill a 35-wide map to 55 %; here a tile is TH whole rows (TH * W <= 256
// pixels: 7 x 35), the pixels of a tile are numbered row-major and dealt 32 per wave, and the K range is walked LINEARLY in
// 32-element groups exactly as the implicit GEMM does -- group g = K elements [32 g, 32 g + 32) of (tap, channel), lane
// group fch holding the 8 channels of chunk 4 g + fch -- so that any C_in that is a multiple of 8 works (48: a group
// straddles two taps) and the results stay bit-identical to the implicit GEMM.  The filter image is the KRSC rows as they
// lie in memory (K linear), rows padded to an odd chunk count (conflict-free fragment reads).
struct DirW { int N, H_out, TH, PH, PW, tiles_h, ntiles, npix; };
constexpr int kWMaxPix = 340;                             // patch pixels a buffer holds ((7 + 2) x (35 + 2) = 333)

template <int C8>
__device__ __forceinline__ int wkey(int pcol) { return C8 == 4 ? ((pcol >> 2) & 3) : C8 == 8 ? ((pcol >> 1) & 7) : 0; }

template <int C8, int CO, int EV>
__device__ __forceinline__ void directw_body(const ConvK& p, const DirW& q, const int first, const int G) {
  constexpr int CI = 8 * C8, NA = CO / 16, NG = (9 * C8 + 3) / 4;
  constexpr int PROUNDS = (kWMaxPix * C8 + kDThreads - 1) / kDThreads;
  constexpr int PBUF = PROUNDS * kDThreads;
  constexpr int WROW = 4 * NG + 1;                        // chunks per filter-image row: the K range, zero fill, one of padding
  constexpr int WCH = CO * WROW;
  constexpr int WROUNDS = (WCH + kDThreads - 1) / kDThreads;
  constexpr int NS = 2 * (CO / 32 + (CO % 32 ? 1 : 0));   // store instructions of a tile's epilogue, per wave
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  u32x4* const wimg = smem;
  u32x4* const pbuf = smem + WROUNDS * kDThreads;         // two patch buffers
  float* const red = reinterpret_cast<float*>(pbuf + 2 * PBUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int npatch = q.PH * q.PW;
  int prow[PROUNDS], pcol[PROUNDS], pc8[PROUNDS];
#pragma unroll
  for (int i = 0; i < PROUNDS; ++i) {
    const int ch = i * kDThreads + tid;
    const int pix = ch / C8, cs = ch - pix * C8;
    prow[i] = pix / q.PW;
    pcol[i] = pix - prow[i] * q.PW;
    pc8[i] = cs ^ wkey<C8>(pcol[i]);                      // the DMA destination is lane-linear: swizzle the SOURCE chunk
    if (pix >= npatch) prow[i] = -1;
  }
  auto issue_patch = [&](int t, int buf) {
    const int img = t / q.tiles_h, th = t - img * q.tiles_h;
    const int h_in0 = th * q.TH - p.pad_t, w_in0 = -p.pad_l;
    u32x4* dst = pbuf + buf * PBUF + wave * 64;
#pragma unroll
    for (int i = 0; i < PROUNDS; ++i) {
      const int h = h_in0 + prow[i], w = w_in0 + pcol[i];
      const bool ok = prow[i] >= 0 && (unsigned)h < (unsigned)p.H_in && (unsigned)w < (unsigned)p.W_in;
      glds16(xr, dst + i * kDThreads, ok ? (img * p.x_img_stride + (h * p.W_in + w) * p.ldx + pc8[i] * 8) * 2 : (int)kOOB);
    }
  };
#pragma unroll
  for (int i = 0; i < WROUNDS; ++i) {                     // the filter image: row 16 a + f = output channel dperm(a, f), K linear
    const int ch = i * kDThreads + tid;
    const int row = ch / WROW, c = ch - row * WROW;
    const int co = dperm<CO>(row >> 4, row & 15);
    const bool ok = ch < WCH && c < 9 * C8 && co < p.C_out;
    glds16(wr, wimg + i * kDThreads + wave * 64, ok ? (co * 9 * CI + c * 8) * 2 : (int)kOOB);
  }
  if (first < q.ntiles) issue_patch(first, 0);
  wait_vmcnt<0>();
  lds_readback_wait(lds_readback_issue(pbuf + (PROUNDS - 1) * kDThreads + wave * 64 + lane));
  raw_barrier();

  const int frow = lane & 15, fch = lane >> 4;
  // this lane's two output pixels (tile-relative, row-major numbering) and the patch slots of their K chunks
  int orow[2], ocol[2];
  bool oval[2];
  int poff[NG][2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int pidx = 32 * wave + 16 * b + frow;
    oval[b] = pidx < q.npix;
    const int pp = oval[b] ? pidx : 0;
    orow[b] = pp / p.W_out;
    ocol[b] = pp - orow[b] * p.W_out;
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k8 = 4 * g + fch;
    const int tap = k8 < 9 * C8 ? k8 / C8 : 0, c8 = k8 < 9 * C8 ? k8 - tap * C8 : 0;   // (past the K range the filter chunk is zero)
    const int r = tap / 3, sx = tap - 3 * r;
#pragma unroll
    for (int b = 0; b < 2; ++b) poff[g][b] = ((orow[b] + r) * q.PW + ocol[b] + sx) * C8 + (c8 ^ wkey<C8>(ocol[b] + sx));
  }
  float s1[NA][4], s2[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[a][r] = 0.f; s2[a][r] = 0.f; }
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
  constexpr int NPq = CO / 32;
  float sc8[NPq > 0 ? NPq : 1][8], sh8[NPq > 0 ? NPq : 1][8], sc4[4], sh4[4];
  if constexpr (EV == 3) {
#pragma unroll
    for (int A = 0; A < NPq; ++A)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 32 * A + 8 * fch + j;
        sc8[A][j] = (p.scale && c < p.C_out) ? p.scale[c] : 1.f;
        sh8[A][j] = (p.shiftv && c < p.C_out) ? p.shiftv[c] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 2 * NPq * 16 + 4 * fch + j;
      sc4[j] = (p.scale && c < p.C_out) ? p.scale[c] : 1.f;
      sh4[j] = (p.shiftv && c < p.C_out) ? p.shiftv[c] : 0.f;
    }
  }
  // filter fragments in registers (as many K groups as ~112 registers hold), the rest read from LDS per tile
  constexpr int WBUD = NA >= 4 ? 80 : 112;                // (64 output channels: 32 accumulator + 32 statistics registers more)
  constexpr int WT = (WBUD / (NA * 4)) < NG ? (WBUD / (NA * 4)) : NG;
  bf16x8 wreg[WT][NA];
#pragma unroll
  for (int g = 0; g < WT; ++g)
#pragma unroll
    for (int a = 0; a < NA; ++a) wreg[g][a] = __builtin_bit_cast(bf16x8, wimg[(a * 16 + frow) * WROW + 4 * g + fch]);

  int buf = 0;
  for (int t = first; t < q.ntiles; t += G) {
    const bool more = t + G < q.ntiles;
    if (more) issue_patch(t + G, buf ^ 1);                // (the other buffer: free since the barrier that ended tile t - G)
    const u32x4* pb = pbuf + buf * PBUF;
    f32x4 acc[NA][2];
#pragma unroll
    for (int a = 0; a < NA; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      bf16x8 wf[NA], pf[2];
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        if (g < WT) wf[a] = wreg[g < WT ? g : 0][a];
        else wf[a] = __builtin_bit_cast(bf16x8, wimg[(a * 16 + frow) * WROW + 4 * g + fch]);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) pf[b] = __builtin_bit_cast(bf16x8, pb[poff[g][b]]);
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
    }
    {
      const int img = t / q.tiles_h, th = t - img * q.tiles_h;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oh = th * q.TH + orow[b];
        const bool pv = oval[b] && oh < q.H_out;
        const int pix_off = img * p.y_img_stride + (oh * p.W_out + ocol[b]) * p.ldy;
#pragma unroll
        for (int A = 0; A < NPq; ++A) {
          const int c0 = 32 * A + 8 * fch;
          const bool ok = pv && c0 < p.C_out;
          unsigned h8[8];
          float v8[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v8[r] = acc[2 * A][b][r]; v8[4 + r] = acc[2 * A + 1][b][r]; }
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
#pragma unroll
          for (int j = 0; j < 8; ++j) h8[j] = (j & 1) ? (h2[j >> 1] >> 16) : (h2[j >> 1] & 0xffffu);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{h2[0], h2[1], h2[2], h2[3]},
                                                 yr, ok ? (int)((pix_off + c0) * 2) : (int)kOOB, 0, 0);
          if constexpr (EV == 1) {
            if (ok) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float f0 = bf2f((unsigned short)h8[r]), f1 = bf2f((unsigned short)h8[4 + r]);
                s1[2 * A][r] += f0; s2[2 * A][r] += f0 * f0;
                s1[2 * A + 1][r] += f1; s2[2 * A + 1][r] += f1 * f1;
              }
            }
          }
        }
        if constexpr (NA > 2 * NPq) {                       // C_out 48: channels 32 .. 47 in the plain order
          constexpr int a = 2 * NPq;
          const int c0 = a * 16 + 4 * fch;
          const bool ok = pv && c0 < p.C_out;
          unsigned h4[4];
          float v4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[a][b][r];
            if constexpr (EV == 3) { v = v * sc4[r] + sh4[r]; if (p.relu) v = relu_f(v); }
            v4[r] = v;
          }
          const unsigned g2[2] = {pack2bf(v4[0], v4[1]), pack2bf(v4[2], v4[3])};
          h4[0] = g2[0] & 0xffffu; h4[1] = g2[0] >> 16; h4[2] = g2[1] & 0xffffu; h4[3] = g2[1] >> 16;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{g2[0], g2[1]}, yr,
                                                ok ? (int)((pix_off + c0) * 2) : (int)kOOB, 0, 0);
          if constexpr (EV == 1) {
            if (ok) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float f = bf2f((unsigned short)h4[r]); s1[a][r] += f; s2[a][r] += f * f; }
            }
          }
        }
      }
    }
    if (more) {
      wait_vmcnt<NS>();                                   // the next patch has retired (behind it: this tile's NS stores only)
      lds_readback_wait(lds_readback_issue(pbuf + (buf ^ 1) * PBUF + (PROUNDS - 1) * kDThreads + wave * 64 + lane));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
    buf ^= 1;
  }
  if constexpr (EV == 1) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x1 = row_sum16(s1[a][r]), x2 = row_sum16(s2[a][r]);
        if (frow == 0) {
          const int ch = dperm<CO>(a, 4 * fch + r);
          red[(wave * CO + ch) * 2] = x1;
          red[(wave * CO + ch) * 2 + 1] = x2;
        }
      }
    __syncthreads();
    if (tid < CO && tid < p.C_out) {
      float x1 = 0.f, x2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { x1 += red[(w * CO + tid) * 2]; x2 += red[(w * CO + tid) * 2 + 1]; }
      stats_write(p, first, tid, x1, x2);
    }
    // the layer's BN apply as the launch's tail (fused_bn.h): the tiles this workgroup stored, whole rows of the map
    if (p.fa.bar != nullptr) {
      const int HW = q.H_out * p.W_out;
      fused_apply_tail<kDThreads>(p, smem, [&](auto&& fn) {
        for (int t = first; t < q.ntiles; t += G) {
          const int img = t / q.tiles_h, th = t - img * q.tiles_h;
          fn(img * HW + th * q.TH * p.W_out, min(q.TH, q.H_out - th * q.TH) * p.W_out, 0, p.C_out);
        }
      });
    }
  }
}

template <int C8, int CO>
constexpr int directw_lds_bytes() {
  constexpr int NG = (9 * C8 + 3) / 4, WROW = 4 * NG + 1;
  constexpr int PBUF = (kWMaxPix * C8 + kDThreads - 1) / kDThreads * kDThreads;
  constexpr int WB = (CO * WROW + kDThreads - 1) / kDThreads * kDThreads;
  return (WB + 2 * PBUF) * 16 + 8 * CO * 2 * 4;
}

// data gradients: the BN backward of the layers whose activation gradient a whole-width launch wrote, as its tail (fused_bn.h)
template <int C8, int CO>
__device__ __forceinline__ void directw_bwd_tail(const ConvK& p, const DirW& q, const int first, const int G) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  static_assert(kDThreads * 68 + sizeof(FbShared) <= directw_lds_bytes<C8, CO>(), "the tail's reduce area fits");
  const int HW = q.H_out * p.W_out;
  fused_bwd_tail<kDThreads, false>(p, smem, [&](auto&& fn) {
    for (int t = first; t < q.ntiles; t += G) {
      const int img = t / q.tiles_h, th = t - img * q.tiles_h;
      fn(img * HW + th * q.TH * p.W_out, min(q.TH, q.H_out - th * q.TH) * p.W_out, 0, p.C_out);
    }
  });
}

template <int C8, int CO, int EV>
__global__ void __launch_bounds__(kDThreads)
conv_directw_kernel(const ConvK p, const DirW q) {
  directw_body<C8, CO, EV>(p, q, (int)blockIdx.x, (int)gridDim.x);
  if constexpr (EV == 0) {
    if (p.fb.bar != nullptr) directw_bwd_tail<C8, CO>(p, q, (int)blockIdx.x, (int)gridDim.x);
  }
}

template <int C8, int CO>
constexpr int directw_lds() {
  constexpr int NG = (9 * C8 + 3) / 4, WROW = 4 * NG + 1;
  constexpr int PBUF = (kWMaxPix * C8 + kDThreads - 1) / kDThreads * kDThreads;
  constexpr int WB = (CO * WROW + kDThreads - 1) / kDThreads * kDThreads;
  return (WB + 2 * PBUF) * 16 + 8 * CO * 2 * 4;
}

template <int C8, int CO>
int launch_directw(const ConvK& k, const DirW& q, int grid, hipStream_t s) {
  constexpr int lds = directw_lds<C8, CO>();
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr[3] = {false, false, false};
  const int ev = k.epi == MBX_EPI_AFFINE ? 2 : k.stats ? 1 : 0;
  if (!attr[ev]) {
    if (ev == 2) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_directw_kernel<C8, CO, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if (ev) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_directw_kernel<C8, CO, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_directw_kernel<C8, CO, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr[ev] = true;
  }
  if (ev == 2) hipLaunchKernelGGL((conv_directw_kernel<C8, CO, 3>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  else if (ev) hipLaunchKernelGGL((conv_directw_kernel<C8, CO, 1>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  else hipLaunchKernelGGL((conv_directw_kernel<C8, CO, 0>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

// tile height of the whole-width launch for an output W_out wide (0: the map is too wide for it)
inline int directw_th(int W_out) {
  if (W_out < 8 || W_out > 64) return 0;
  int th = 256 / W_out;
  const int by_patch = kWMaxPix / (W_out + 2) - 2;
  if (by_patch < th) th = by_patch;
  return th >= 2 ? th : 0;
}

template <int CI, int CO>
constexpr int direct3_lds() {
  constexpr int C8 = CI / 8;
  constexpr int PCH = kDPH * kDPW * C8, PBUF = (PCH + kDThreads - 1) / kDThreads * kDThreads;
  constexpr int WCH = 9 * CO * C8, WB = (WCH + kDThreads - 1) / kDThreads * kDThreads;
  return (WB + (CI == 32 ? 3 : 2) * PBUF) * 16 + 8 * CO * 2 * 4;
}

template <int CI, int CO>
int launch_direct3(const ConvK& k, const DirK& q, int grid, hipStream_t s) {
  constexpr int lds = direct3_lds<CI, CO>();
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr[3] = {false, false, false};
  const int ev = k.epi == MBX_EPI_AFFINE ? 2 : k.stats ? 1 : 0;
  if (!attr[ev]) {
    if (ev == 2) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct3_kernel<CI, CO, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if (ev) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct3_kernel<CI, CO, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct3_kernel<CI, CO, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr[ev] = true;
  }
  if (ev == 2) hipLaunchKernelGGL((conv_direct3_kernel<CI, CO, 3>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  else if (ev) hipLaunchKernelGGL((conv_direct3_kernel<CI, CO, 1>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  else hipLaunchKernelGGL((conv_direct3_kernel<CI, CO, 0>), dim3(grid), dim3(kDThreads), lds, s, k, q);
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

int direct3_cus() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return ncu;
}

}  // namespace

// workgroups (= statistics rows) of the direct launch for an N x H_out x W_out output
int mbx_direct3_grid(int N, int H_out, int W_out) {
  const long nt = (long)N * ((H_out + kDTH - 1) / kDTH) * ((W_out + kDTW - 1) / kDTW);
  const int ncu = direct3_cus();
  return nt < ncu ? (int)nt : ncu;
}

// workgroups (= statistics rows) of the whole-width direct launch (tile_config 97)
int mbx_directw_grid(int N, int H_out, int W_out) {
  const int th = directw_th(W_out);
  if (!th) return 0;
  const long nt = (long)N * ((H_out + th - 1) / th);
  const int ncu = direct3_cus();
  return nt < ncu ? (int)nt : ncu;
}

// mbx_conv_desc.tile_config = 97: the whole-width direct 3x3 launch (conv_directw_kernel) -- 3x3 / stride 1, forward or data
// gradient, C_in 32 / 48 / 64, C_out <= 64 (a multiple of 8), maps 8..64 wide; bf16 store with or without statistics, or affine.
int mbx_launch_directw(void* convk, int N, int H_out, hipStream_t s) {
  ConvK& k = *reinterpret_cast<ConvK*>(convk);
  if (k.R != 3 || k.S != 3 || k.mul != 1 || k.shift || (k.epi != MBX_EPI_STORE && k.epi != MBX_EPI_AFFINE) || k.accumulate || k.skip || k.bits ||
      k.rscale != 0.f || k.bw_n || (k.epi == MBX_EPI_AFFINE && k.stats))
    return MBX_ERR_UNSUPPORTED;
  if ((k.C_in != 32 && k.C_in != 48 && k.C_in != 64) || k.C_out > 64 || k.C_out % 8 || k.pad_t < 0 || k.pad_t > 2 || k.pad_l < 0 || k.pad_l > 2)
    return MBX_ERR_UNSUPPORTED;
  DirW q;
  q.TH = directw_th(k.W_out);
  if (!q.TH || k.W_in > k.W_out + 2) return MBX_ERR_UNSUPPORTED;   // (the patch is W_out + 2 columns wide: VALID / SAME / full padding all fit)
  q.N = N; q.H_out = H_out;
  q.PH = q.TH + 2; q.PW = k.W_out + 2;
  q.tiles_h = (H_out + q.TH - 1) / q.TH;
  q.ntiles = N * q.tiles_h;
  q.npix = q.TH * k.W_out;
  int grid = mbx_directw_grid(N, H_out, k.W_out);
  if ((k.fa.bar || k.fb.bar) && k.max_wg > 0 && grid > k.max_wg) grid = k.max_wg;    // (grid-barrier launches honour the caller's CU cap)
  k.stats_cap = stats_cap_for(grid);                       // one add per workgroup and channel
  if (k.dry) return MBX_OK;
  const int co = k.C_out <= 32 ? 32 : k.C_out <= 48 ? 48 : 64;
#define MBX_DW(C8_, CO_) if (k.C_in == 8 * C8_ && co == CO_) return launch_directw<C8_, CO_>(k, q, grid, s);
  MBX_DW(4, 32) MBX_DW(4, 48) MBX_DW(4, 64) MBX_DW(6, 32) MBX_DW(6, 48) MBX_DW(6, 64) MBX_DW(8, 32) MBX_DW(8, 48)
#undef MBX_DW
  return MBX_ERR_UNSUPPORTED;                             // 64 -> 64: filter image + two patches do not fit 160 KB of LDS
}

// mbx_conv_desc.tile_config = 96: the direct 3x3 launch.  MBX_ERR_UNSUPPORTED for anything but a 3x3 / stride-1 convolution
// (forward, or the data gradient of one) with C_in in {32, 64}, C_out <= 64 (a multiple of 8) and a bf16 store epilogue
// with or without statistics, or the affine (+ relu) epilogue of a folded batch norm.
int mbx_launch_direct3(void* convk, int N, int H_out, hipStream_t s) {
  ConvK& k = *reinterpret_cast<ConvK*>(convk);
  if (k.mul == 2 && !k.shift && k.C_in == 8) {
    // the network's first layer (forward only: stride 2, packed RGB input): conv_stem_kernel
    if (k.R != 3 || k.S != 3 || (k.epi != MBX_EPI_STORE && k.epi != MBX_EPI_AFFINE) || k.accumulate || k.skip || k.bits || k.rscale != 0.f || k.bw_n ||
        (k.epi == MBX_EPI_AFFINE && k.stats) || k.C_out > 32 || k.C_out % 8 || k.pad_t < 0 || k.pad_t > 2 || k.pad_l < 0 || k.pad_l > 2 ||
        k.Ktot != 72)
      return MBX_ERR_UNSUPPORTED;
    DirK q;
    q.N = N; q.H_out = H_out;
    q.tiles_h = (H_out + kDTH - 1) / kDTH;
    q.tiles_w = (k.W_out + kDTW - 1) / kDTW;
    const long nt = (long)N * q.tiles_h * q.tiles_w;
    if (nt >= (1L << 30)) return MBX_ERR_UNSUPPORTED;
    q.ntiles = (int)nt;
    if (k.dry) return MBX_OK;
    k.stats_cap = stats_cap_for(mbx_direct3_grid(N, H_out, k.W_out));
    return launch_stem(k, q, mbx_direct3_grid(N, H_out, k.W_out), s);
  }
  if (k.R != 3 || k.S != 3 || k.mul != 1 || k.shift || (k.epi != MBX_EPI_STORE && k.epi != MBX_EPI_AFFINE) || k.accumulate || k.skip || k.bits ||
      k.rscale != 0.f || k.bw_n || (k.epi == MBX_EPI_AFFINE && k.stats))
    return MBX_ERR_UNSUPPORTED;
  if ((k.C_in != 32 && k.C_in != 64) || k.C_out > 64 || k.C_out % 8 || k.pad_t < 0 || k.pad_t > 2 || k.pad_l < 0 || k.pad_l > 2 ||
      (k.C_in == 64 && k.C_out > 48))
    return MBX_ERR_UNSUPPORTED;
  DirK q;
  q.N = N; q.H_out = H_out;
  q.tiles_h = (H_out + kDTH - 1) / kDTH;
  q.tiles_w = (k.W_out + kDTW - 1) / kDTW;
  const long nt = (long)N * q.tiles_h * q.tiles_w;
  if (nt >= (1L << 30)) return MBX_ERR_UNSUPPORTED;
  q.ntiles = (int)nt;
  const int grid = mbx_direct3_grid(N, H_out, k.W_out);
  k.stats_cap = stats_cap_for(grid);
  if (k.dry) return MBX_OK;
  const int co = k.C_out <= 32 ? 32 : k.C_out <= 48 ? 48 : 64;
  if (k.C_in == 32) {
    if (co == 32) return launch_direct3<32, 32>(k, q, grid, s);
    if (co == 48) return launch_direct3<32, 48>(k, q, grid, s);
    return launch_direct3<32, 64>(k, q, grid, s);
  }
  if (co == 32) return launch_direct3<64, 32>(k, q, grid, s);
  if (co == 48) return launch_direct3<64, 48>(k, q, grid, s);
  return MBX_ERR_UNSUPPORTED;                             // 64 -> 64: filter image + two patches do not fit 160 KB of LDS
}
