// libmbx: conv_igemm5_kernel -- the persistent, loader / compute specialised implicit-GEMM convolution (round 2) -- and
// its launcher.  Separate translation unit so that it compiles beside conv.hip.
#include "conv_common.h"
#ifndef MBX_I5_NST4
#define MBX_I5_NST4 4
#endif

namespace {

// ------------------------------------------------------------------------------------------ igemm5
// The same implicit GEMM (same LDS image, same K order, same MFMA order: bit-identical results), restructured the way
// the grouped weight gradient was in round 2:
//  * SIXTEEN waves with fixed roles: waves 0-7 multiply, waves 8-15 issue the LDS-DMA.  In conv_igemm3_kernel every
//    wave does both behind one barrier, so the ~0.5 us of DMA issue and the LDS reads + MFMAs of a K step add up; it
//    gets its overlap from 2-3 small blocks per CU, which caps the tile at 128 x 64 / 128 x 128 (43 - 64 FLOP per byte
//    pulled through the L2 -> LDS path, and that path, ~45 GB/s per CU, is what bounds these launches).
//  * PERSISTENT: one block per CU walks tiles t = first, first + grid, ... (or pulls them from a counter); the loaders
//    run NST - 1 K steps ahead ACROSS tile boundaries, so the next tile's first filter / pixel tiles land while the
//    compute waves write the current tile out -- straight from the accumulators (conv_epilogue_direct: 16-byte stores, no
//    LDS staging, no barriers; round 2 staged it through an LDS window in 16-32 row passes behind two barriers each,
//    which the loaders had to attend).
//  * Tiles are (64 MY pixels) x (64 NW channels): 192 x 128 and 256 x 128 move 77 / 85 FLOP per byte.
// Stride-2 data gradients and the float32 head epilogue stay on conv_igemm3_kernel.
template <int MY, int NW>
struct Ig5 {
  static constexpr int BM = 64 * MY, BN = 64 * NW;
  static_assert(NW >= 1 && NW <= 4, "64 / 128 / 192 / 256 output channels per tile");
  // compute-wave grid (WM x WN = 8) and the 16 x 16 blocks per wave:
  //   128x64: 4x2 (2x2 blocks)   256x64: 8x1 (2x4)   192x64: 4x2 (3x2)   64x128: 2x4 (2x2)   128x128: 2x4 (4x2)
  //   192x128: 2x4 (6x2)         256x128: 4x2 (4x4)  128x256: 2x4 (4x4)  128x192: 4x2 (2x6: 96 channels per wave)
  // (128 x 192, round 4: ONE column tile for the 160 / 192-channel 1x7 / 7x1 layers of block17, whose launches are paced by
  // the L2 -> LDS operand feed -- two column tiles re-read every pixel row)
  static constexpr int WM = (NW == 1) ? (MY == 4 ? 8 : 4) : (NW == 3) ? 4 : (MY == 4 && NW == 2) ? 4 : 2;
  static constexpr int WN = 8 / WM;
  static constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  static constexpr int STAGE = (BM + BN) * 8;                       // 16-byte slots per ring stage
  static constexpr int RED_BYTES = WM * BN * 2 * 4;                 // batch-norm statistics: [WM waves][BN channels][2] floats
  // ring depth: the loaders run NST - 1 K steps ahead.  Four stages where they fit: the K loop is paced by the landing
  // of the operand tiles (0.5 us per 32 KB step measured with timestamps inside the kernel, whatever the compute waves
  // do), i.e. by the bytes in flight against the latency.
  static constexpr int NST = (4 * STAGE * 16 + RED_BYTES + 64 <= 160 * 1024) ? MBX_I5_NST4 : 3;
  static constexpr int RING_BYTES = NST * STAGE * 16;
  static constexpr int LDS_BYTES = RING_BYTES + RED_BYTES + 16;     // + four tile ids of the queued assignment
  static_assert(TM % 16 == 0 && TN % 32 == 0 && WM * WN == 8, "wave grid");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// (A TWELVE-wave form -- four loader waves carrying two row octets each, 168 registers per wave, the WHOLE tile's epilogue reads in
// flight across the K loop -- was built in round 5, bit-identical, and measured level: rows 3.7 -> 1.7 us, K loop 5.7 -> 7.2 us;
// removed in round 6, LAB_NOTES round 5.)
template <int MY, int NW, int EV, int MODE>
__global__ void __launch_bounds__(1024)
conv_igemm5_kernel(const ConvK p) {
  using G = Ig5<MY, NW>;
  constexpr bool PW = MODE == 1;
  constexpr int BM = G::BM, BN = G::BN, TM = G::TM, TN = G::TN, MI = G::MI, NI = G::NI, STAGE = G::STAGE, NST = G::NST;
  constexpr int NLW = 8, LV = 1;                                    // loader waves; row octets per loader wave
  constexpr int NL = LV * (MY + NW);                                // LDS-DMA instructions per loader wave and K step
  static_assert((NST - 2) * NL < 64, "vmcnt");
  constexpr int NBAR = (EV == 1 || EV == 6) ? 1 : 0;                             // barriers of one tile's epilogue (statistics reduce)
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  float* const red = reinterpret_cast<float*>(smem + NST * STAGE);  // [WM][BN][2]
  // QUEUED tile assignment (p.work_counter): tile sequence of this workgroup = its first tile by position, then
  // gridDim.x + (values of the counter).  Lane 0 of compute wave 0 fetches the id of tile j+2 when tile j starts (a
  // returning atomic, in flight during the K loop -- the compute waves issue no other vector-memory instruction there)
  // and publishes it through s_ids[(j + 2) & 3] at the end of tile j's K loop, barriers before anyone needs it: the loaders read
  // the id of tile j+1 while they are still inside tile j (they run NST - 1 K steps ahead, hence the nk > NST condition;
  // shorter K loops fall back to the static deal).  Same tiles, same arithmetic: results do not depend on the mode.
  // (an explicit LDS pointer: through a generic or volatile one hipcc emits FLAT accesses, which count on vmcnt too)
  typedef __attribute__((address_space(3))) int* lds_int_ptr;
  const lds_int_ptr s_ids = (lds_int_ptr)(red + G::WM * BN * 2);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int ntiles = p.tiles_m * p.tiles_n;
  const int first = xcd_remap(blockIdx.x, gridDim.x);               // tiles first, first + grid, ...
#ifdef MBX_I5_STAMPS
  // (probe, debug builds: do workgroups that start together stay in step -- equal K loops, then 256 epilogues sharing HBM -- and
  // would a start offset of (first % 4) phases help?  Measured: no; every launch is longer by the largest offset, at BATCH_SIZE 64
  // and 256 alike (tools/stagger_probe.sh, LAB_NOTES round 5): a tile's time is a property of its CU's memory path.)
  if (p.stagger) {
    const int n = (first & 3) * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);        // 512 cycles = a quarter of a microsecond
  }
#endif
  const int nk = (p.Ktot + 63) >> 6;
  // nk > NST: the loaders read s_ids[j + 1] when they have issued the last K step of tile j, i.e. in iteration nk - NST of
  // tile j's K loop; the compute waves publish that id after the last barrier of tile j - 1 -- with nk == NST the read
  // would sit in iteration 0, behind that same barrier and in front of any other (a race: ADVICE round 3).
  // ... and only when there ARE more tiles than workgroups: in a single-round launch (most of this network's) every
  // workgroup has its one tile by position and the counter could only say "none left" -- but the first fetch is a RETURNING
  // atomic on one address, 256 workgroups at once (~3 us), and its result used to be awaited in front of the first barrier:
  // 0.11 ms per step (MBX_I5_STATIC A/B, round 4).  Multi-round launches fetch it without waiting (below).
  // (a launch with the BN apply as its tail deals statically: every workgroup meets the others at the grid barrier anyway, and
  // the tail re-derives the workgroup's tiles from its position)
  const bool fused_tail = ((EV == 1) && p.fa.bar != nullptr) || ((EV == 0) && p.fb.bar != nullptr);         // (uniform)
  const bool queued = p.work_counter != nullptr && nk > NST && ntiles > (int)gridDim.x && !fused_tail;        // (uniform)

  if (wave >= 8) {
    // -------------------------------------------------------------------------------------------- loader waves
    const int lw0 = wave - 8;                                       // row octets lw0, lw0 + NLW, ... of every piece
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
    const int r8 = lane >> 3;
    const int chunk = (lane & 7) ^ r8;                              // source chunk of this lane's slot (row & 7 == r8)
    // filter rows (64 i + 8 lw + r8) carry the key of the permuted fragment reads (conv_igemm3_kernel): bits 3-4 and 1 of the row
    int chunkw[LV];
#pragma unroll
    for (int v = 0; v < LV; ++v) chunkw[v] = (lane & 7) ^ ((((lw0 + NLW * v) & 3) << 1) | ((r8 >> 1) & 1));
    const int ldx2 = p.ldx * 2;
    // issue cursor: (tile, K step) two steps ahead of the compute waves; row state of THAT tile
    int t_i = first, it_i = 0, st_issue = 0, j_i = 0;               // j_i: index of the issue tile in this workgroup's sequence
    int hb[LV][MY], wb[LV][MY], ro[LV][MY], wo[LV][NW];
    int kc = 0, kr = 0, ks = 0;
#define MBX5_SETUP_TILE()                                                                                     \
  do {                                                                                                        \
    const int tn_ = t_i % p.tiles_n, tm_ = t_i / p.tiles_n;                                                   \
    _Pragma("unroll") for (int v = 0; v < LV; ++v) {                                                          \
      const int lw = lw0 + NLW * v;                                                                           \
      _Pragma("unroll") for (int i = 0; i < MY; ++i) {                                                        \
        const int m = tm_ * BM + 64 * i + 8 * lw + r8;                                                        \
        const bool mv = m < p.M;                                                                              \
        int img, oh, ow;                                                                                      \
        decode_pixel(p, mv ? (unsigned)m : 0u, img, oh, ow);                                                  \
        hb[v][i] = mv ? oh * p.mul - p.pad_t : -(1 << 24);                                                    \
        wb[v][i] = ow * p.mul - p.pad_l;                                                                      \
        ro[v][i] = (img * p.x_img_stride + (hb[v][i] * p.W_in + wb[v][i]) * p.ldx) * 2;                       \
        if (PW && !mv) ro[v][i] = (int)kOOB;                                                                  \
      }                                                                                                       \
      _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                        \
        const int n = tn_ * BN + 64 * i + 8 * lw + r8;                                                        \
        wo[v][i] = n < p.C_out ? n * p.Ktot * 2 : -1;                                                         \
      }                                                                                                       \
    }                                                                                                         \
    kc = chunk * 8; kr = 0; ks = 0;                                                                           \
    while (kc >= p.C_in) { kc -= p.C_in; if (++ks >= p.S) { ks = 0; ++kr; } }                                 \
  } while (0)
// One K step's LDS-DMA in two halves (pixel rows, then filter rows + cursor advance): the loaders wait for and read
// back the NEXT step's tile between the halves (landing hand-off, see conv_igemm3_kernel), so that the read-back's LDS
// round trip is covered by the issue of the second half instead of sitting on the loaders' critical path (as a
// wait + read-back behind the whole issue it cost 1.5 ms per step in round 2).
#define MBX5_ISSUE_A()                                                                                        \
  do {                                                                                                        \
    const bool kv = kr < p.R;                                                                                 \
    _Pragma("unroll") for (int v = 0; v < LV; ++v) {                                                          \
      u32x4* sp = smem + st_issue * STAGE + (lw0 + NLW * v) * 64;                                             \
      if (PW) {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < MY; ++i)                                                        \
          glds16(xr, sp + i * 512, (kv && ro[v][i] >= 0) ? (ro[v][i] + kc * 2) : (int)kOOB);                  \
      } else {                                                                                                \
        const int toff = (kr * p.W_in + ks) * ldx2 + kc * 2;                                                  \
        _Pragma("unroll") for (int i = 0; i < MY; ++i) {                                                      \
          const bool ok = kv && ((unsigned)(hb[v][i] + kr) < (unsigned)p.H_in) &&                             \
                          ((unsigned)(wb[v][i] + ks) < (unsigned)p.W_in);                                     \
          glds16(xr, sp + i * 512, ok ? (ro[v][i] + toff) : (int)kOOB);                                       \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)
#define MBX5_ISSUE_B()                                                                                        \
  do {                                                                                                        \
    _Pragma("unroll") for (int v = 0; v < LV; ++v) {                                                          \
      u32x4* sp = smem + st_issue * STAGE + (lw0 + NLW * v) * 64;                                             \
      const int kb = (it_i * 64 + chunkw[v] * 8) * 2;                 /* the filter lane's own K position */    \
      const bool kvw = kb < p.Ktot * 2;                                                                       \
      _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                          \
        glds16(wr, sp + (MY + i) * 512, (kvw && wo[v][i] >= 0) ? (wo[v][i] + kb) : (int)kOOB);                \
    }                                                                                                         \
    kc += 64;                                                                                                 \
    while (kc >= p.C_in) { kc -= p.C_in; if (++ks >= p.S) { ks = 0; ++kr; } }                                 \
    st_issue = st_issue == NST - 1 ? 0 : st_issue + 1;                                                        \
    if (++it_i == nk) {                                                                                       \
      it_i = 0; ++j_i;                                                                                        \
      t_i = queued ? s_ids[j_i & 3] : t_i + (int)gridDim.x;                                                   \
      if (t_i < ntiles) MBX5_SETUP_TILE();                                                                    \
    }                                                                                                         \
  } while (0)
#define MBX5_ISSUE() do { MBX5_ISSUE_A(); MBX5_ISSUE_B(); } while (0)
    // read-back of this lane's slot of the wave's LAST piece (filter piece NW - 1) of ring slot `st_pub`, returned
#define MBX5_READBACK() lds_readback_wait(lds_readback_issue(smem + st_pub * STAGE + (MY + NW - 1) * 512 + (lw0 + NLW * (LV - 1)) * 64 + lane))

    int st_pub = 0;                                                 // ring slot of the step published next
    if (t_i < ntiles) {
      MBX5_SETUP_TILE();
      MBX5_ISSUE();                                                 // global step 0
      bool full = true;
#pragma unroll
      for (int a = 1; a < NST - 1; ++a) {
        if (t_i < ntiles) MBX5_ISSUE(); else full = false;
      }
      if (full) wait_vmcnt<(NST - 2) * NL>(); else wait_vmcnt<0>();
      MBX5_READBACK();
    }
    raw_barrier();                                                  // step 0 has landed (and s_ids[0..1] are written)
    int jc = 0;                                                     // index of the tile the compute waves are on
    for (int t = first; t < ntiles; t = queued ? s_ids[++jc & 3] : t + (int)gridDim.x) {
      for (int it = 0; it < nk; ++it) {
        const bool more = t_i < ntiles;                             // anything left to issue (this or a later tile)?
        st_pub = st_pub == NST - 1 ? 0 : st_pub + 1;                // the NEXT step's slot
#ifdef MBX_I5_STAMPS
        if (p.dbg & 4) { raw_barrier(); continue; }                 // timing probe: the loaders issue nothing (the multiplying waves alone; wrong results)
#endif
        if (more) MBX5_ISSUE_A();
#ifdef MBX_I5_STAMPS
        if (p.dbg & 2) { if (more) MBX5_ISSUE_B(); raw_barrier(); continue; }   // timing probe: do not wait for the landing (wrong results)
#endif
#ifndef MBX_NO_PROBE_I5
        // the NEXT step (steps s+1 .. s+NST-2 are outstanding, + the MY pieces just issued) has retired: this wave's share
        if (more) wait_vmcnt<(NST - 3) * NL + LV * MY>(); else wait_vmcnt<0>();
        const unsigned probe = lds_readback_issue(smem + st_pub * STAGE + (MY + NW - 1) * 512 + (lw0 + NLW * (LV - 1)) * 64 + lane);
        if (more) MBX5_ISSUE_B();
        lds_readback_wait(probe);                                   // read-back returned: publish
#else                                                               // (debug builds only: A/B of what the hand-off costs)
        if (more) { MBX5_ISSUE_B(); wait_vmcnt<(NST - 2) * NL>(); } else wait_vmcnt<0>();
#endif
        raw_barrier();
      }
#pragma unroll 1
      for (int b = 0; b < NBAR; ++b) raw_barrier();                 // the compute waves' statistics reduce
    }
#undef MBX5_READBACK
#undef MBX5_ISSUE_B
#undef MBX5_ISSUE_A
#undef MBX5_ISSUE
#undef MBX5_SETUP_TILE
  } else {
  // ---------------------------------------------------------------------------------------------- compute waves
  constexpr int WN = G::WN;
  const int wn = wave % WN, wm = wave / WN;
  const int frow = lane & 15, fch = lane >> 4;
  const int fr0 = frow * 8 + (fch ^ (frow & 7));
  const int fr1 = frow * 8 + ((4 + fch) ^ (frow & 7));
  const int fwrow = 8 * (frow >> 2) + (frow & 3), fwkey = ((frow >> 2) << 1) | ((frow >> 1) & 1);
  const int fw0 = fwrow * 8 + (fch ^ fwkey);                        // filter fragments: permuted rows, their own key
  const int fw1 = fwrow * 8 + ((4 + fch) ^ fwkey);
  int st_comp = 0;
  const bool fetcher = queued && tid == 0;
  // next id off the counter (>= ntiles: none left) in two halves: the RETURNING atomic is only issued here; the value is
  // touched where it is published, so that the wait for it sits there and not at the issue (as one expression the compiler
  // waited for the atomic -- one address, every workgroup at once: 1-3 us -- right behind it, at the start of every tile)
  auto fetch_raw = [&]() -> int { return atomicAdd(p.work_counter, 1); };
  auto tile_of = [&](const int raw) -> int { const int v = (int)gridDim.x + raw; return v < ntiles ? v : ntiles; };
  // the id of this workgroup's SECOND tile: fetched here, published behind the barrier (the loaders read it when they have
  // issued the first tile's last K step, nk - NST >= 1 iterations later)
  int raw1 = 0;
  if (fetcher) { s_ids[0] = first; raw1 = fetch_raw(); }
  raw_barrier();                                                    // step 0 has landed
  if (fetcher) s_ids[1] = tile_of(raw1);                            // (issued in front of the barrier; visible behind the first K step's)
  int jt = 0;
  for (int t = first; t < ntiles; t = queued ? s_ids[++jt & 3] : t + (int)gridDim.x) {
    const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#ifdef MBX_I5_STAMPS
    const int tnum = (t - first) / (int)gridDim.x;
    const bool stamp = p.stamps && tid == 0 && tnum < 8 && blockIdx.x < 64;
#define MBX5_STAMP(i) do { if (stamp) p.stamps[(blockIdx.x * 8 + tnum) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define MBX5_STAMP(i) do { } while (0)
#endif
    MBX5_STAMP(0);                                                  // tile start
#ifdef MBX_I5_STAMPS
    const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();   // (core-clock counter: the clock the chip holds in the K loop)
#endif
    // The epilogue's READS (residual skip / accumulate source / ReLU mask) are issued HERE, before the K loop: the compute
    // waves issue no other vector-memory instruction in the loop and never wait on vmcnt there, so the reads -- up to
    // 64 KB per tile, the HBM-bound half of a short-K tile's life -- land while the tile is multiplied; the epilogue then
    // is arithmetic + fire-and-forget 16-byte stores.  (As many pixel blocks as 16 registers hold: the whole tile for
    // 128x64 / 128x128 / 256x64 residual tiles, half of it for 256x128; the rest is read in the epilogue.)
    constexpr int NA = NI / 2;
    // pixel blocks whose reads are issued ahead: 16 registers' worth (four 16-byte reads per lane)
    // (EV = 7: the accumulate + mask epilogue with the mask read from the relu SIGN BITS, ConvK::bits -- to the shared
    // epilogue it is EV 2 with the bf16-mask path compiled out; plain EV 2 here has the bits path compiled out: these
    // kernels have no registers for both.  The 256x128 / 128x256 residual tiles, which also write the bits: one block.)
    constexpr int EVC = EV == 7 ? 2 : EV, BMODE = EV == 7 ? 2 : EV == 2 ? 0 : 1;
    constexpr int PRE_RAW = (EV == 4 && G::BM * G::BN >= 256 * 128) ? 1
                            : (EV == 4 || EVC == 2 || EV == 6) ? 4 / (NA * (EVC == 2 ? 2 : 1)) : 0;
    constexpr int PREB = PRE_RAW > MI ? MI : PRE_RAW;
    const int cl0 = wn * TN + fch * 8, mlane = m0 + wm * TM + frow, clane = n0 + cl0;
    u32x4 pla[PREB > 0 ? PREB : 1][NA], plb[PREB > 0 ? PREB : 1][NA];
    if constexpr (PREB > 0) conv_epilogue_issue_reads<EVC, false, NA, PREB, BMODE>(p, mlane, clane, 0, pla, plb);
    // id of tile jt + 2: the returning atomic is issued here (BEHIND the reads above: the compiler pairs its result register
    // with an address register of theirs otherwise and waits for it at once) and returns during the K loop
    int raw2 = 0;
    if (fetcher) raw2 = fetch_raw();
    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < nk; ++it) {
#ifdef MBX_I5_STAMPS
      if (p.dbg & 1) { st_comp = st_comp == NST - 1 ? 0 : st_comp + 1; raw_barrier(); continue; }   // timing probe: compute waves idle
#endif
      const u32x4* cP = smem + st_comp * STAGE + (wm * TM) * 8;
      const u32x4* cW = smem + st_comp * STAGE + BM * 8 + (wn * TN) * 8;
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
    MBX5_STAMP(1);                                                  // K loop done
#ifdef MBX_I5_STAMPS
    if (stamp) p.stamps[(blockIdx.x * 8 + tnum) * 4 + 3] = __builtin_amdgcn_s_memtime() - cyc0;
#endif
    // ---------------------------------------------------------------- epilogue: straight from the accumulators
    if (fetcher) s_ids[(jt + 2) & 3] = tile_of(raw2);               // visible behind the next tile's K-loop barriers
    float s1[NA][8], s2[NA][8], sc[NA][8], sh[NA][8];
    conv_epilogue_channels<EVC, NA>(p, clane, sc, sh, s1, s2);
    if constexpr (PREB > 0) {
      conv_epilogue_finish<EVC, false, NI, MI, PREB, BMODE>(p, acc, mlane, clane, 0, pla, plb, sh, sc, s1, s2);
      asm volatile("" ::: "memory");
    }
    if constexpr (PREB < MI) {
      // the rest of the tile: reads issued here, one or two pixel blocks at a time (128 registers per lane in a 16-wave block)
      constexpr int REM = MI - PREB, CH = (EV == 0 || EV == 1 || EV == 3) ? REM : (REM % 2 == 0 ? 2 : 1);
      conv_epilogue_range<EVC, false, NI, MI, PREB, MI, CH, BMODE>(p, acc, mlane, clane, sh, sc, s1, s2);
    }
    MBX5_STAMP(2);                                                  // rows written
    if constexpr (EV == 1 || EV == 6) {
      // batch-norm statistics partials of this tile (EV = 6: the backward sums of the layers this data gradient feeds): lane's pixels -> the 16 lanes sharing its channels (DPP row sums)
      // -> the WM waves along the pixel dimension through `red` (its own LDS area: the ring is being refilled)
#pragma unroll
      for (int A = 0; A < NA; ++A) { row_sum16_x8(s1[A]); row_sum16_x8(s2[A]); }
#pragma unroll
      for (int A = 0; A < NA; ++A)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x1 = s1[A][j], x2 = s2[A][j];
          if (frow == 0) {
            red[(wm * BN + cl0 + 32 * A + j) * 2] = x1;
            red[(wm * BN + cl0 + 32 * A + j) * 2 + 1] = x2;
          }
        }
      lds_barrier();                                                // (the loaders attend: NBAR)
      if (tid < BN && n0 + tid < p.C_out) {
        float x1 = 0.f, x2 = 0.f;
#pragma unroll
        for (int w = 0; w < G::WM; ++w) { x1 += red[(w * BN + tid) * 2]; x2 += red[(w * BN + tid) * 2 + 1]; }
        if constexpr (EV == 6) bw_stats_write(p, tile_m, n0 + tid, x1, x2); else stats_write(p, tile_m, n0 + tid, x1, x2);
      }
      // (no second barrier: `red` is rewritten only after the next tile's K loop, nk >= 1 barriers away)
    }
#undef MBX5_STAMP
  }
  }  // compute waves
  if constexpr (EV == 1) {
    // ---------------------------------------------------------------------------- the layer's BN apply as the launch's tail
    // (fused_bn.h): all sixteen waves; the ring is dead -- its first 12 C_out bytes hold the per-channel parameters
    if (fused_tail) {
      constexpr int NT = 512 + 64 * NLW;
      fused_apply_tail<NT>(p, smem, [&](auto&& fn) {
        for (int t = first; t < ntiles; t += (int)gridDim.x) fn((t / p.tiles_n) * BM, BM, (t % p.tiles_n) * BN, BN);
      });
    }
  }
  if constexpr (EV == 0 && !(MY == 2 && NW == 3)) {       // (128 x 192: 96 accumulator registers leave no room for the tail: spills)
    // -------------------------------- data gradients: the BN backward of the layers whose activation gradient this launch wrote
    if (fused_tail) {
      constexpr int NT = 512 + 64 * NLW;
      static_assert(NT * 68 + sizeof(FbShared) <= G::RING_BYTES, "the tail's reduce area fits the ring");
      static_assert(BM * BN <= 8 * kFbChunk * NT, "a lane's share of a tile is one chunk");
      fused_bwd_tail<NT, true>(p, smem, [&](auto&& fn) {
        for (int t = first; t < ntiles; t += (int)gridDim.x) fn((t / p.tiles_n) * BM, BM, (t % p.tiles_n) * BN, BN);
      });
    }
  }
}


template <int MY, int NW>
int launch5(ConvK& k, hipStream_t s) {
  using G = Ig5<MY, NW>;
  k.tiles_m = (k.M + G::BM - 1) / G::BM;
  k.tiles_n = (k.C_out + G::BN - 1) / G::BN;
  k.stats_cap = stats_cap_for(k.tiles_m);
  const int ntiles = k.tiles_m * k.tiles_n;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  int grid = ntiles < ncu ? ntiles : ncu;                           // one persistent block per CU
  if (k.max_wg > 0 && grid > k.max_wg) grid = k.max_wg;             // ... or fewer: CUs left to a kernel on another stream
  const int ev = k.epi == MBX_EPI_RESIDUAL ? 4 : k.epi == MBX_EPI_AFFINE ? 3 : k.bw_n ? 6 : k.stats ? 1 : k.bits ? 7 : (k.accumulate || k.skip) ? 2 : 0;
  // the BN-backward statistics epilogue (EV = 6) on the 256 x 128 tile: 64 accumulator + 48 sum / threshold registers + the
  // reads in flight do not fit the 128-register budget of a 16-wave block (33 spilled): not built, the caller takes 192 x 128
  constexpr bool kNo6 = (MY == 4 && NW == 2) || NW >= 3;
  // (128 x 192: 96 channels per wave -- the accumulate + mask epilogue's reads in flight spill as well: store / statistics /
  // affine / residual only)
  constexpr bool kNo2 = NW == 3;
  if ((ev == 6 && kNo6) || ((ev == 2 || ev == 7) && kNo2)) return MBX_ERR_UNSUPPORTED;
  if (k.fb.bar && (ev != 0 || (MY == 2 && NW == 3))) return MBX_ERR_UNSUPPORTED;     // the BN-backward tail: plain store launches
  if (k.fa.bar && ev != 1) return MBX_ERR_UNSUPPORTED;
  if (k.dry) return MBX_OK;                                         // mbx_conv_supported(): the checks above, no launch
  static bool attr[8][2] = {};
#define MBX5_LAUNCH(EV, MODE)                                                                                 \
  do {                                                                                                        \
    if (!attr[EV][MODE]) {                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm5_kernel<MY, NW, EV, MODE>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);                    \
      attr[EV][MODE] = true;                                                                                  \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_igemm5_kernel<MY, NW, EV, MODE>), dim3(grid), dim3(1024), G::LDS_BYTES, s, k);   \
  } while (0)
#define MBX5_EV(EV) case EV: if (k.pw) MBX5_LAUNCH(EV, 1); else MBX5_LAUNCH(EV, 0); break;
  switch (ev) {
    MBX5_EV(0) MBX5_EV(1) MBX5_EV(3) MBX5_EV(4)
    case 2:
      if constexpr (!kNo2) { if (k.pw) MBX5_LAUNCH(2, 1); else MBX5_LAUNCH(2, 0); }
      break;
    case 6:
      if constexpr (!kNo6) { if (k.pw) MBX5_LAUNCH(6, 1); else MBX5_LAUNCH(6, 0); }
      break;
    case 7:
      if constexpr (!kNo2) { if (k.pw) MBX5_LAUNCH(7, 1); else MBX5_LAUNCH(7, 0); }
      break;
  }
#undef MBX5_EV
#undef MBX5_LAUNCH
  MBX_LAUNCH_CHECK();
  return MBX_OK;
}

}  // namespace

// tile shapes of the igemm5 launch: mbx_conv_desc.tile_config = 32 + index + 1
extern const int mbx_i5_tiles[][2] = {{128, 64}, {128, 128}, {192, 128}, {256, 128}, {256, 64}, {128, 192}, {128, 256}};
extern const int mbx_i5_num_tiles = 7;

int mbx_launch_igemm5(void* convk, int index, hipStream_t s) {
  ConvK& k = *reinterpret_cast<ConvK*>(convk);
  if (k.shift || k.epi == MBX_EPI_STORE_F32) return MBX_ERR_UNSUPPORTED;   // stride-2 data gradient / float32 heads: igemm3
  switch (index) {
    case 0: return launch5<2, 1>(k, s);
    case 1: return launch5<2, 2>(k, s);
    case 2: return launch5<3, 2>(k, s);
    case 3: return launch5<4, 2>(k, s);
    case 4: return launch5<4, 1>(k, s);
    case 5: return launch5<2, 3>(k, s);
    case 6: return launch5<2, 4>(k, s);
    default: return MBX_ERR_UNSUPPORTED;
  }
}
