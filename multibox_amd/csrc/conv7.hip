// libmbx: conv_igemm7_kernel -- the persistent implicit GEMM for POINTWISE (1x1, unit stride, unpadded) convolutions with
// a short reduction (K <= 384), with the FILTER PANEL RESIDENT IN LDS (round 3).
//
// Why.  The short-K 1x1 launches of the residual stages (the "up" convolutions 384 -> 1088 / 128 -> 320 forward, the data
// gradients of the fused 1x1s 320 -> 1088 / 96 -> 320 / 384 -> 2080) are 2.5 ms of the step.  Their K loop is paced by the
// bytes a CU pulls through the L2 -> LDS path (the step time of every igemm5 tile follows its bytes),
// and a streaming 128 x 128 tile moves 32 KB per 64-deep step -- half of it filter rows that EVERY tile of the same output
// channels re-reads.  Here a workgroup is bound to one 128-channel column tile for the whole launch: it loads that tile's
// filter panel (128 rows x K, <= 96 KB) ONCE and then streams only pixel tiles: 16 KB per step for the same 2.1 MFLOP --
// 128 FLOP per byte through the DMA path against 64 (128 x 128 streaming) or 85 (256 x 128 streaming).
//
// Structure = conv_igemm5_kernel's (conv5.hip): 8 MFMA waves + 8 LDS-DMA loader waves, 3-deep pixel ring, loaders two K
// steps ahead across tile boundaries, landing read-back before every publishing barrier, epilogue straight from the
// accumulators with its reads issued before the K loop.  Same K order and MFMA order as every other igemm tile: results
// are bit-identical.  Work: the workgroups with blockIdx % tiles_n == g form the group of column tile g; a workgroup takes
// pixel tile (blockIdx / tiles_n) first and every further one of ITS column tile from work_counter[g] (one int per column
// tile, zero at launch; NULL: dealt statically inside the group) -- three tiles ahead, so that the loaders, which run a whole
// tile ahead when K = 128, always find the next id published.
//
// Result (round 3, DESIGN.md sections 4 and 7).  Bit-identical on every test; chosen by the trace-based tuner for the
// detect leg's block17 "up" forward (256 patches: 2436 vs 2553 us per batch) and for six shapes of the 512 x 512
// configuration, for NO shape at BATCH_SIZE 64.  Halving the bytes did not halve the step: beside a 96 KB panel the ring
// holds 2 x 16 KB of pixel tiles in flight against the ~1.1 us an LDS-DMA takes to land (igemm5 keeps 3 x 32 KB), and the
// launches this was written for spend their time in the residual / accumulate traffic of the epilogue (2.9 - 4.3 TB/s),
// which no operand layout changes.
#include "conv_common.h"

namespace {

constexpr int k7BM = 128, k7BN = 128, k7NST = 3, k7MaxNk = 6;
constexpr int k7Stage = k7BM * 8;                                   // 16-byte slots of one pixel stage (16 KB)
constexpr int k7PanelTile = k7BN * 8;                               // slots of one 64-deep K tile of the panel (16 KB)

template <int EV>
__global__ void __launch_bounds__(1024)
conv_igemm7_kernel(const ConvK p) {
  constexpr int BM = k7BM, BN = k7BN, NST = k7NST, STAGE = k7Stage, MY = 2, NW = 2;
  constexpr int WM = 2, WN = 4, TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16, NA = NI / 2;
  static_assert(EV == 0 || EV == 2 || EV == 3 || EV == 4, "store / accumulate(+mask) / affine / residual");
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  const int nk = (p.Ktot + 63) >> 6;                                // <= k7MaxNk (host)
  u32x4* const panel = smem;                                        // [nk][BN rows][8 chunks], filter swizzle
  u32x4* const ring = smem + nk * k7PanelTile;                      // [NST][BM rows][8 chunks], pixel swizzle
  typedef __attribute__((address_space(3))) int* lds_int_ptr;
  const lds_int_ptr s_ids = (lds_int_ptr)(ring + NST * STAGE);      // eight pixel-tile ids of the queued assignment

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int g = (int)blockIdx.x % p.tiles_n, r = (int)blockIdx.x / p.tiles_n;      // column tile (fixed), rank inside its group
  const int gsize = ((int)gridDim.x - g + p.tiles_n - 1) / p.tiles_n;              // workgroups of this group
  const int none = p.tiles_m;                                       // sentinel: no further pixel tile
  const int first = r < p.tiles_m ? r : none;
  const bool queued = p.work_counter != nullptr;                    // (uniform)
  const int n0 = g * BN;
  // pixel tile j of this workgroup's sequence: s_ids[j & 7] (queued) or first + j gsize (static)
#define MBX7_SEQ(J, PREV) (queued ? s_ids[(J) & 7] : ((PREV) + gsize < none ? (PREV) + gsize : none))

  if (wave >= 8) {
    // -------------------------------------------------------------------------------------------- loader waves
    const int lw = wave - 8;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
    const int r8 = lane >> 3;
    const int chunk = (lane & 7) ^ r8;                              // pixel tile: key = row & 7
    const int chunkw = (lane & 7) ^ (((lw & 3) << 1) | ((r8 >> 1) & 1));     // filter rows: the key of the permuted fragment reads
    if (first == none) return;                                      // (whole workgroup: `first` is uniform)
    // ---- the filter panel of column tile g, once: rows 64 i + 8 lw + r8, all K tiles
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int n = n0 + 64 * i + 8 * lw + r8;
      const int wo = n < p.C_out ? n * p.Ktot * 2 : -1;
      for (int kt = 0; kt < nk; ++kt) {
        const int kb = (kt * 64 + chunkw * 8) * 2;
        glds16(wr, panel + kt * k7PanelTile + i * 512 + lw * 64, (wo >= 0 && kb < p.Ktot * 2) ? wo + kb : (int)kOOB);
      }
    }
    // ---- pixel tiles: issue cursor (tile, K step) NST - 1 steps ahead of the compute waves
    int t_i = first, j_i = 0, it_i = 0, st_issue = 0;
    int ro[MY];
#define MBX7_SETUP_TILE()                                                                                     \
  do {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MY; ++i) {                                                          \
      const int m = t_i * BM + 64 * i + 8 * lw + r8;                                                          \
      int img, oh, ow;                                                                                        \
      decode_pixel(p, m < p.M ? (unsigned)m : 0u, img, oh, ow);                                               \
      ro[i] = m < p.M ? (img * p.x_img_stride + (oh * p.W_in + ow) * p.ldx) * 2 : (int)kOOB;                  \
    }                                                                                                         \
  } while (0)
#define MBX7_ISSUE_PIECE(I)                                                                                   \
  glds16(xr, ring + st_issue * STAGE + (I) * 512 + lw * 64,                                                   \
         (ro[I] >= 0 && (it_i * 64 + chunk * 8) < p.C_in) ? ro[I] + (it_i * 64 + chunk * 8) * 2 : (int)kOOB)
#define MBX7_ADVANCE()                                                                                        \
  do {                                                                                                        \
    st_issue = st_issue == NST - 1 ? 0 : st_issue + 1;                                                        \
    if (++it_i == nk) {                                                                                       \
      it_i = 0; ++j_i;                                                                                        \
      t_i = MBX7_SEQ(j_i, t_i);                                                                               \
      if (t_i < none) MBX7_SETUP_TILE();                                                                      \
    }                                                                                                         \
  } while (0)
    MBX7_SETUP_TILE();
    MBX7_ISSUE_PIECE(0); MBX7_ISSUE_PIECE(1);                       // global step 0
    MBX7_ADVANCE();                                                 // (nk >= 2, host: no tile change here)
    MBX7_ISSUE_PIECE(0); MBX7_ISSUE_PIECE(1);                       // global step 1: still the first tile
    wait_vmcnt<MY>();                                               // panel + step 0 have retired (in order), step 1 in flight
    lds_readback_wait(lds_readback_issue(ring + 1 * 512 + lw * 64 + lane));
    raw_barrier();                                                  // panel and step 0 published (and s_ids[0..2] written)
    // the cursor advance of step 1 is done HERE, behind the barrier: with nk == 2 it moves to the next tile and reads
    // s_ids[1], which the compute waves wrote in front of that barrier
    MBX7_ADVANCE();
    int st_pub = 0, jc = 0;
    for (int t = first; t < none; ) {
      for (int it = 0; it < nk; ++it) {
        const bool more = t_i < none;
        st_pub = st_pub == NST - 1 ? 0 : st_pub + 1;                // the NEXT step's slot
        if (more) MBX7_ISSUE_PIECE(0);
        // the NEXT step (one step outstanding + the piece just issued) has retired: this wave's share
        if (more) wait_vmcnt<1>(); else wait_vmcnt<0>();
        const unsigned probe = lds_readback_issue(ring + st_pub * STAGE + 1 * 512 + lw * 64 + lane);
        if (more) { MBX7_ISSUE_PIECE(1); MBX7_ADVANCE(); }
        lds_readback_wait(probe);
        raw_barrier();
      }
      ++jc;
      t = MBX7_SEQ(jc, t);
    }
#undef MBX7_ADVANCE
#undef MBX7_ISSUE_PIECE
#undef MBX7_SETUP_TILE
    return;
  }

  // ---------------------------------------------------------------------------------------------- compute waves
  const int wn = wave % WN, wm = wave / WN;
  const int frow = lane & 15, fch = lane >> 4;
  const int fr0 = frow * 8 + (fch ^ (frow & 7));
  const int fr1 = frow * 8 + ((4 + fch) ^ (frow & 7));
  const int fwrow = 8 * (frow >> 2) + (frow & 3), fwkey = ((frow >> 2) << 1) | ((frow >> 1) & 1);
  const int fw0 = fwrow * 8 + (fch ^ fwkey);
  const int fw1 = fwrow * 8 + ((4 + fch) ^ fwkey);
  if (first == none) return;
  const bool fetcher = queued && tid == 0;
  auto fetch_tile = [&]() -> int {                                  // next pixel tile of column tile g (>= tiles_m: none left)
    const int v = gsize + atomicAdd(p.work_counter + g, 1);
    return v < none ? v : none;
  };
  if (fetcher) { s_ids[0] = first; s_ids[1] = fetch_tile(); s_ids[2] = fetch_tile(); }
  int st_comp = 0;
  raw_barrier();                                                    // panel and step 0 have landed
  int jt = 0;
  for (int t = first; t < none; ) {
    const int m0 = t * BM;
    const int cl0 = wn * TN + fch * 8, mlane = m0 + wm * TM + frow, clane = n0 + cl0;
    // the epilogue's reads, issued before the K loop (the compute waves never wait on vmcnt inside it)
    constexpr int PRE_RAW = (EV == 4 || EV == 2) ? 8 / (NA * (EV == 2 ? 2 : 1)) : 0;
    constexpr int PREB = PRE_RAW > MI ? MI : PRE_RAW;
    u32x4 pla[PREB > 0 ? PREB : 1][NA], plb[PREB > 0 ? PREB : 1][NA];
    if constexpr (PREB > 0) conv_epilogue_issue_reads<EV, false, NA, PREB>(p, mlane, clane, 0, pla, plb);
    // id of tile jt + 3: only the returning atomic is issued here (behind the reads above); its value is touched where the id is
    // published (as one expression the compiler waited for it right behind the issue, at the start of every tile: conv5.hip)
    int raw3 = 0;
    if (fetcher) raw3 = atomicAdd(p.work_counter + g, 1);
    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < nk; ++it) {
      const u32x4* cP = ring + st_comp * STAGE + (wm * TM) * 8;
      const u32x4* cW = panel + it * k7PanelTile + (wn * TN) * 8;
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
      st_comp = st_comp == NST - 1 ? 0 : st_comp + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's LDS reads are done before the stage is reused
      raw_barrier();
    }
    // ---------------------------------------------------------------- epilogue: straight from the accumulators
    if (fetcher) { const int v3 = gsize + raw3; s_ids[(jt + 3) & 7] = v3 < none ? v3 : none; }   // read by the loaders two tiles from now at the earliest
    float s1[NA][8], s2[NA][8], sc[NA][8], sh[NA][8];
    conv_epilogue_channels<EV, NA>(p, clane, sc, sh, s1, s2);
    if constexpr (PREB > 0) {
      conv_epilogue_finish<EV, false, NI, MI, PREB>(p, acc, mlane, clane, 0, pla, plb, sh, sc, s1, s2);
      asm volatile("" ::: "memory");
    }
    if constexpr (PREB < MI) {
      constexpr int REM = MI - PREB, CH = (EV == 0 || EV == 3) ? REM : (REM % 2 == 0 ? 2 : 1);
      conv_epilogue_range<EV, false, NI, MI, PREB, MI, CH>(p, acc, mlane, clane, sh, sc, s1, s2);
    }
    ++jt;
    t = MBX7_SEQ(jt, t);
  }
#undef MBX7_SEQ
}

}  // namespace

// mbx_conv_desc.tile_config = 65: the panel-resident pointwise launch.  Returns MBX_ERR_UNSUPPORTED for anything but a
// pointwise convolution with 64 < K <= 384 and a bf16 store / accumulate / affine / residual epilogue without statistics.
int mbx_launch_igemm7(void* convk, hipStream_t s) {
  ConvK& k = *reinterpret_cast<ConvK*>(convk);
  const int nk = (k.Ktot + 63) >> 6;
  if (!k.pw || k.shift || k.stats || k.bw_n || k.epi == MBX_EPI_STORE_F32 || nk < 2 || nk > k7MaxNk || (k.C_in % 8)) return MBX_ERR_UNSUPPORTED;
  k.tiles_m = (k.M + k7BM - 1) / k7BM;
  k.tiles_n = (k.C_out + k7BN - 1) / k7BN;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  const long ntiles = (long)k.tiles_m * k.tiles_n;
  int grid = ntiles < ncu ? (int)ntiles : ncu;                      // one persistent workgroup per CU, dealt round-robin to the column tiles
  if (k.max_wg > 0 && grid > k.max_wg) grid = k.max_wg;             // (mbx_conv_desc.max_workgroups)
  if (grid < k.tiles_n) grid = k.tiles_n;
  if (k.tiles_n > 32) return MBX_ERR_UNSUPPORTED;                   // (one work counter per column tile, 32 per launch)
  const int lds = nk * k7PanelTile * 16 + k7NST * k7Stage * 16 + 32;
  const int ev = k.epi == MBX_EPI_RESIDUAL ? 4 : k.epi == MBX_EPI_AFFINE ? 3 : (k.accumulate || k.skip || k.bits) ? 2 : 0;
  if (k.dry) return MBX_OK;                                         // mbx_conv_supported(): the checks above, no launch
  static bool attr[5] = {};
#define MBX7_LAUNCH(EV)                                                                                       \
  case EV:                                                                                                    \
    if (!attr[EV]) {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm7_kernel<EV>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, k7MaxNk * k7PanelTile * 16 + k7NST * k7Stage * 16 + 32); \
      attr[EV] = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_igemm7_kernel<EV>), dim3(grid), dim3(1024), lds, s, k);                          \
    break;
  switch (ev) { MBX7_LAUNCH(0) MBX7_LAUNCH(2) MBX7_LAUNCH(3) MBX7_LAUNCH(4) }
#undef MBX7_LAUNCH
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}
