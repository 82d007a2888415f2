// Device helpers and the kernel parameter block shared by the convolution translation units (conv.hip: igemm3 +
// weight gradients + host entry points; conv5.hip: the persistent igemm5 kernel).  gfx950 only.
#pragma once
#include "common.h"
#include "grid_barrier.h"

namespace {


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// Two floats to one packed bf16 pair (lo in bits 0..15) in ONE v_cvt_pk_bf16_f32: the same rounding as f2bf, which the
// compiler emits as one conversion PER VALUE plus a shift and an or (4 instructions a pair instead of 1).
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{lo, hi}, pk_bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// Lanes l, l + 16, l + 32, l + 48 (the four rows of a wave) each hold one byte (bits 0..7 of b, the rest zero): all four
// get the dword [row 0 | row 1 << 8 | row 2 << 16 | row 3 << 24].  EXEC must be all ones.
__device__ __forceinline__ unsigned gather4_rows(unsigned b) {
  typedef unsigned g4_u32x2 __attribute__((ext_vector_type(2)));
  const g4_u32x2 r = __builtin_amdgcn_permlane16_swap(b, b, false, false);     // (even row, odd row) of the lane's row pair
  const unsigned t = r.x | (r.y << 8);
  const g4_u32x2 q = __builtin_amdgcn_permlane32_swap(t, t, false, false);     // (lower half, upper half)
  return q.x | (q.y << 16);
}
// max(x, 0) in one v_max_f32 (fmaxf costs a second one: the compiler canonicalises its operand first).  Inline asm is
// invisible to the compiler's hazard checks: apply it ONLY to the result of a VALU instruction, never straight to an MFMA
// accumulator (the affine / residual arithmetic always comes first).  The packed conversion above is the vector form of
// the plain cast for the same reason: it IS the first reader of the accumulators in the store epilogues.
__device__ __forceinline__ float relu_f(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

struct ConvK {
  const unsigned short* x; int x_img_stride, ldx, H_in, W_in, C_in;
  const unsigned short* w; int C_out, R, S, Ktot;
  unsigned x_bytes, w_bytes;
  unsigned y_bytes, skip_bytes, acc_bytes;  // buffer-descriptor ranges of the epilogue's tensors (bf16 outputs; < 2 GB each)
  int mul, shift, pad_t, pad_l, W_out, HW_out, M;
  void* y; int y_img_stride, ldy;
  int epi, relu, accumulate;
  const float* scale; const float* shiftv;
  const unsigned short* skip; int skip_img_stride, ld_skip; float rscale;
  const unsigned short* acc_src; int acc_img_stride, ld_acc;      // accumulate: OLD value read from here (may alias y)
  // ReLU sign bits, one per output element: byte [m * bits_ld + c / 8] bit (c & 7) = (stored y[m][c] > 0).  WRITTEN by the
  // residual epilogue (EV = 4, relu); READ by the accumulate epilogue (EV = 2) as the relu-backward mask in place of the
  // 16-times larger bf16 tensor (skip is NULL then).  m = raster pixel index (never the parity walk: host check).
  unsigned char* bits; int bits_ld; unsigned bits_bytes;
  float* stats;
  float stats_cap;                       // fixed-point rows: bound of ONE adder's |sums| (stats_cap_for(adders per channel), below)
  int stats_mod, stats_ld;               // stats_mod = R > 0: tile sums ADDED atomically into row (tile % R) of [R][stats_ld][2] (zero at launch); 0: a plain row per tile
  // EV = 6 (data gradients): BATCH-NORM BACKWARD statistics of the layers whose activation gradient this launch writes.
  // Output channels [bw_cb[i], bw_cb[i + 1]) belong to segment i: y = that layer's pre-BN output [M][bw_ldy] (pointer at the
  // segment's first channel), thr = its relu threshold on y, sums {sum g, sum g y} with g = (y > thr) ? stored gradient : 0
  // ADDED into row (tile % bw_mod) of bw_stats[i] = [bw_mod][bw_sld][2].  Segment boundaries are multiples of 32 channels.
  int bw_n, bw_mod;
  int bw_cb[4], bw_ldy[4], bw_sld[4];
  const unsigned short* bw_y[4];
  const float* bw_thr[4];
  float* bw_stats[4];
  FusedApply fa;                       // fa.bar != NULL: the layer's BN apply runs as the tail of this launch (fused_bn.h)
  FusedBwd fb;                         // fb.bar != NULL (data gradients): the BN backward of the layers it feeds runs as its tail
  int tiles_m, tiles_n;
  unsigned mg_hw, sh_hw, mg_w, sh_w;   // magic-number division by HW_out / W_out
  int pw;                              // pointwise: R = S = 1, no padding, unit stride
  // stride-2 data gradient (shift = 1): output pixels are walked PARITY-CLASS-major (class = (oh & 1) * 2 + (ow & 1)),
  // so a pixel tile lies in one class and only the filter taps of that class are multiplied (a quarter of them)
  int cls_m0[4], cls_hw[4], cls_w[4];  // first pixel index, pixels per image (Hc * Wc) and row length Wc per class
  int skip_taps;                       // C_in % 64 == 0: K tiles never straddle taps, whole taps can be skipped
  int parity;                          // pixels walked parity-class-major (0: raster order, MBX_NO_TAP_SKIP=1)
  int* work_counter;                   // igemm5: tiles after a workgroup's first come from this counter (NULL: static)
  int max_wg;                          // persistent launches: grid cap (0: one workgroup per CU); host side only
  int dry;                             // host side only: mbx_conv_supported() -- every check, no launch
  // split-K (float32 partial tiles, igemm3 EV = 5 only): slice `lid / (tiles_m tiles_n)` multiplies K steps
  // [slice * kps, ...) and stores to y + slice * y_split_stride floats; ksplit = 1: the whole K range
  int ksplit, kps; long long y_split_stride;
#ifdef MBX_I5_STAMPS
  unsigned long long* stamps;          // debug build: wall_clock64() per tile phase of the first 8 tiles of 64 blocks
  int dbg;                             // debug build: MBX_I5_DBG timing probes (bit 0: compute waves idle, bit 1: loaders do not wait)
  int stagger;                         // debug build: MBX_STAGGER -- workgroup w of a persistent launch starts (w % 4) * stagger quarter-microseconds late
#endif
};

constexpr int kThreads = 256;

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4): four v_add_f32_dpp row_ror, every lane ends up with the
// row sum -- no LDS traffic (ds_bpermute shuffles made the epilogue VALU/LDS-bound).  Written as ONE asm block of fused
// DPP adds: from the builtin (update_dpp + add) the compiler SLP-packs the adds into v_pk_add_f32, which cannot take a DPP
// operand, and emits v_mov_b32_dpp + hazard nops + packed add: 2.5x the instructions (the statistics epilogues do 32 of these
// sums on the tail of a tile).  A DPP source written by the preceding VALU instruction needs two wait states: the leading
// s_nop covers the caller's last write, and inside the block each add's input is three instructions old.
__device__ __forceinline__ float row_sum16(float v) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
               : "+v"(v));
  return v;
}
// ... eight independent sums at once: the adds of one rotation step back to back (no wait states needed between them)
__device__ __forceinline__ void row_sum16_x8(float (&v)[8]) {
#define MBX_RS_STEP(R)                                                                                                     \
  "v_add_f32_dpp %0, %0, %0 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %2, %2, %2 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %4, %4, %4 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %6, %6, %6 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 row_ror:" #R " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" MBX_RS_STEP(8) MBX_RS_STEP(4) MBX_RS_STEP(2) MBX_RS_STEP(1)
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#undef MBX_RS_STEP
}
// exact m / d for m < 2^31: q = (m * magic) >> shift, magic = floor(2^shift / d) + 1, shift = 31 + ceil(log2 d)
__device__ __forceinline__ unsigned fast_div(unsigned m, unsigned magic, unsigned shift) {
  return (unsigned)(((unsigned long long)m * magic) >> shift);
}
// one channel's statistics sums of one tile (or of one workgroup) into the partial table: a plain float32 row per tile, or --
// stats_mod rows whatever the tile count -- ADDED into row (tile mod stats_mod) of a table that is zero at launch.  The adds
// are 64-bit INTEGER atomics on the sums in fixed point (2^-20 units: |sum| < 4.4e12 IN TOTAL, resolution 1e-6 -- the tile sums are
// float32 values of magnitude 1 .. 1e6): integer addition is associative, so the table does not depend on the order in which
// the tiles arrive and the forward pass stays bit-reproducible (float atomics made it differ from run to run in the last
// bits of every mean, which a chaotic network amplifies).  Fire and forget, performed at the memory side.
constexpr float kStatsFix = 1048576.f;                  // 2^20
// OUT OF RANGE (ADVICE rounds 4 and 5).  The conversion is defined for |x| 2^20 < 2^63 only -- and it is the TOTAL that must stay
// in range, not the tile: a table word takes one add per adder of its row (tile rows / workgroups / images: 9 .. 5 400 of them),
// and the consumer adds the rows once more in a bare int64.  So every adder's sums are held below
//   ConvK::stats_cap = kStatsTotal / (adders of ONE channel over ALL rows)      (stats_cap_for(), set by every launcher)
// which bounds the grand total of a channel by kStatsTotal = 2^42 (2^62 in fixed point: |sum y| and sum y^2 < 4.4e12 -- the
// documented range of include/mbx.h) whatever the order of arrival, with no wrap anywhere: not in a row, not in the consumer.
// A sum that is not finite or not below the cap POISONS its channel instead of wrapping into finite garbage: the sum-of-squares
// word -- otherwise a sum of NON-NEGATIVE adds that stays below 2^62 -- is forced to the most negative integer by a signed
// atomic MIN; the legitimate adds that arrive before or after it move it by less than 2^62 in all, so the word (and the
// consumer's sum over the rows) stays NEGATIVE: sticky, and order-independent as a verdict.  The consumer
// (bn_apply_rows_kernel) turns a negative word or a negative total into NaN mean / rstd: the activations, the loss and the
// matching status go non-finite exactly as with the float32 rows -- loud, not silent.  (reachable: `--fine_tune` straight
// after a fresh start feeds activations of ~1e16 into the training-mode head batch norms)
constexpr float kStatsTotal = 4398046511104.f;          // 2^42
__host__ __device__ __forceinline__ float stats_cap_for(long long adders) { return kStatsTotal / (float)(adders > 0 ? adders : 1); }
__device__ __forceinline__ void stats_add_fixed(unsigned long long* o, const float x1, const float x2, const float cap) {
  if (__builtin_fabsf(x1) < cap && __builtin_fabsf(x2) < cap) {                      // (false for NaN / inf as well)
    atomicAdd(o, (unsigned long long)__float2ll_rn(x1 * kStatsFix));
    atomicAdd(o + 1, (unsigned long long)__float2ll_rn(x2 * kStatsFix));
  } else {
    atomicMin(reinterpret_cast<long long*>(o + 1), (long long)0x8000000000000000ull);
  }
}
__device__ __forceinline__ void stats_write(const ConvK& p, const int tile_row, const int ch, const float x1, const float x2) {
  if (p.stats_mod) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stats) + ((size_t)(tile_row % p.stats_mod) * p.stats_ld + ch) * 2;
    stats_add_fixed(o, x1, x2, p.stats_cap);
  } else {
    float* o = p.stats + ((size_t)tile_row * p.stats_ld + ch) * 2;
    o[0] = x1;
    o[1] = x2;
  }
}
// segment of output channel c in the BN-backward statistics table (EV = 6)
__device__ __forceinline__ int bw_seg(const ConvK& p, const int c) {
  int s = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < p.bw_n && c >= p.bw_cb[i]) s = i;
  return s;
}
__device__ __forceinline__ void bw_stats_write(const ConvK& p, const int tile_row, const int ch, const float x1, const float x2) {
  const int s = bw_seg(p, ch);
  float* o = p.bw_stats[s] + ((size_t)(tile_row % p.bw_mod) * p.bw_sld[s] + (ch - p.bw_cb[s])) * 2;
  unsafeAtomicAdd(o, x1);
  unsafeAtomicAdd(o + 1, x2);
}
// pixel index m -> (image, output row, output column)
__device__ __forceinline__ void decode_pixel(const ConvK& p, unsigned m, int& img, int& oh, int& ow) {
  img = (int)fast_div(m, p.mg_hw, p.sh_hw);
  const int rem = (int)m - img * p.HW_out;
  oh = (int)fast_div((unsigned)rem, p.mg_w, p.sh_w);
  ow = rem - oh * p.W_out;
}
// ... in the parity-class-major order of the stride-2 data gradient
__device__ __forceinline__ void decode_pixel_parity(const ConvK& p, unsigned m, int& img, int& oh, int& ow) {
  const int c = ((int)m >= p.cls_m0[1]) + ((int)m >= p.cls_m0[2]) + ((int)m >= p.cls_m0[3]);
  const int m0 = c == 0 ? p.cls_m0[0] : c == 1 ? p.cls_m0[1] : c == 2 ? p.cls_m0[2] : p.cls_m0[3];
  const int hw = c == 0 ? p.cls_hw[0] : c == 1 ? p.cls_hw[1] : c == 2 ? p.cls_hw[2] : p.cls_hw[3];
  const int wc = c == 0 ? p.cls_w[0] : c == 1 ? p.cls_w[1] : c == 2 ? p.cls_w[2] : p.cls_w[3];
  unsigned r = m - (unsigned)m0;
  img = (int)(r / (unsigned)hw);
  r -= (unsigned)img * (unsigned)hw;
  const int a = (int)(r / (unsigned)wc), b = (int)r - a * wc;
  oh = 2 * a + (c >> 1);
  ow = 2 * b + (c & 1);
}
__device__ __forceinline__ int pixel_class(const ConvK& p, int m) {
  return (m >= p.cls_m0[1]) + (m >= p.cls_m0[2]) + (m >= p.cls_m0[3]);
}
constexpr unsigned kOOB = 0x80000000u;      // byte offset beyond every tensor: buffer loads return 0 there

// Buffer loads: 32-bit byte offsets off an SGPR descriptor; an out-of-range offset yields zeros,
// so zero padding / tile edges need no branch and no exec masking.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}

// XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch); give each
// XCD a contiguous run of logical tiles so that tiles sharing a pixel panel share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------------------------------
// The GEMM loop: tiles staged by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip) into a
// 3-deep ring of 64-deep K tiles: tile t+2 is in flight while tile t is multiplied, so two tiles of
// global latency are covered per block.  The LDS image is the XOR-swizzled [row][8 chunks] layout;
// because an LDS-DMA wave-instruction writes 64 consecutive 16-B slots (8 rows x 8 chunks), the
// swizzle is applied to the SOURCE chunk each lane fetches (chunk = slot ^ (row & 7)).
// Counted s_waitcnt vmcnt + raw s_barrier: nothing in the loop drains the DMA queue.
// LDS-DMA: 16 B per lane from buffer offset `off` to lds_dst + lane (lds_dst wave-uniform); offsets
// past the buffer write zeros.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, u32x4* lds_dst, int off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, off, 0, 0, 0);
}
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
// Bare s_barrier (no vmcnt / lgkmcnt drain: LDS-DMA stays in flight across it), fenced for the COMPILER on both sides:
// llvm.amdgcn.s.barrier is IntrNoMem, so without the fences nothing in the IR keeps the LDS accesses that follow it
// from being scheduled above it.  (Not the cause of the round-2 side-stream corruption -- that was the 2-deep ring's
// landing, see conv_igemm3_kernel -- but a hole of the same kind.)
__device__ __forceinline__ void raw_barrier() {
#ifndef MBX_UNFENCED_BARRIER                       // (debug builds only: the regression test's "does it catch the old form" check)
  asm volatile("" ::: "memory");
#endif
  __builtin_amdgcn_s_barrier();
#ifndef MBX_UNFENCED_BARRIER
  asm volatile("" ::: "memory");
#endif
}
// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for this wave's outstanding
// vector-memory operations (fire-and-forget float atomics of the previous work item keep draining behind it)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// debug builds only (tools/ab_builds.sh): what the landing hand-off costs, per kernel family
#ifdef MBX_NO_LANDING_PROBE
#define MBX_NO_PROBE_I3 1
#define MBX_NO_PROBE_I5 1
#define MBX_NO_PROBE_WG 1
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LANDING READ-BACK of an LDS-DMA destination.  Deliberately a FLAT load through the LDS aperture, not a ds_read: it
// travels the vector-memory path (TA -> LDS) that the wave's `buffer_load ... lds` writes took, so it cannot overtake
// them; its return (lgkmcnt) therefore says that every LDS-DMA this wave had retired by `s_waitcnt vmcnt` before
// issuing it has reached the LDS array.  Round 2's fix was this instruction without knowing it: hipcc compiled the
// `volatile unsigned*` read of the ring to `flat_load_dword ... sc0 sc1` followed by `s_waitcnt vmcnt(0) lgkmcnt(0)`
// (a FLAT access counts on both counters) -- correct, but the vmcnt(0) drained the DMA queue, which is what made the
// same probe cost 1.5 ms per step in the loader waves of the deep rings.  Written as inline asm the compiler adds no
// wait: the issue is free-standing and lds_readback_wait() waits on lgkmcnt only.  (The flat load also holds a vmcnt
// slot until it returns; a counted vmcnt wait issued after it can only become more conservative by that one.)
// lds_slot: the calling lane's 16-byte slot of the piece (piece base + lane); the lanes read CONSECUTIVE dwords of the
// piece's first 256 bytes instead (base + 4 lane): with dwords at a 16-byte stride the 64 lanes met 8 banks four ways and
// the read-back alone was 11 % of the convolution kernels' LDS cycles (SQ_LDS_BANK_CONFLICT, profiles/r03_conv_lds_counters.txt)
__device__ __forceinline__ unsigned lds_readback_issue(const void* lds_slot) {
  unsigned v;
  const unsigned long long a = reinterpret_cast<unsigned long long>(lds_slot) - 12ull * (threadIdx.x & 63);   // generic address: shared aperture | offset
  asm volatile("flat_load_dword %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void lds_readback_wait(unsigned v) { asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(v) : "memory"); }

// ------------------------------------------------------------------------------------------
// The epilogue of the implicit-GEMM kernels, straight from the accumulators (conv_igemm3_kernel and conv_igemm5_kernel).
// acc[a][b][r] holds output channel  32 (a >> 1) + 8 fch + 4 (a & 1) + r  (relative to the wave tile) of pixel
// 16 b + frow: blocks 2A and 2A + 1 give a lane EIGHT CONSECUTIVE channels of one pixel = one 16-byte bf16 store;
// the four lanes of a pixel cover 64 contiguous bytes.  mlane = the lane's pixel for b = 0, clane = its first channel for
// A = 0.  EV: 0 store, 1 store + batch-norm statistics sums (s1 / s2: per-lane sums of the STORED, bf16-rounded values
// over its MI pixels), 2 accumulate (+ ReLU mask), 3 affine (+ReLU), 4 residual (+ReLU).
// BRANCH-FREE: every global access goes through a buffer descriptor with a range-checked 32-bit offset (out-of-tile
// lanes read zeros and their stores are dropped), so that ALL the 16-byte reads of a chunk of BCH pixel blocks (residual
// skip, accumulate source, mask) are issued back to back before the first one is consumed -- with a branch around each
// group, as first written, every group exposed its own memory latency (the data-gradient launches with two reads per
// group got 6-10 % SLOWER than the LDS-staged epilogue of round 2).  BCH bounds the registers the reads in flight take.
// One pixel block's (image, pixel) pair for lane pixel m (clamped when out of range)
template <bool SH>
__device__ __forceinline__ void epi_pixel(const ConvK& p, const int m, int& img, int& pix) {
  const unsigned mm = m < p.M ? (unsigned)m : 0u;
  if (SH && p.parity) {
    int oh_, ow_;
    decode_pixel_parity(p, mm, img, oh_, ow_);
    pix = oh_ * p.W_out + ow_;
  } else {
    img = (int)fast_div(mm, p.mg_hw, p.sh_hw);
    pix = (int)mm - img * p.HW_out;
  }
}

// Phase 1 for pixel blocks [b0, b0 + BCH): issue every 16-byte read of the epilogue (nothing waits here).
// BMODE: the relu sign bits (ConvK::bits) -- 1: used when the pointer is set (one uniform branch), 0: compiled out,
// 2: EV = 2 masks by them unconditionally and the bf16 mask path is compiled out (igemm5's registers do not hold both).
template <int EV, bool SH, int NA, int BCH, int BMODE = 1>
__device__ __forceinline__ void conv_epilogue_issue_reads(const ConvK& p, const int mlane, const int clane, const int b0,
                                                          u32x4 (&la)[BCH][NA], u32x4 (&lb)[BCH][NA]) {
  if constexpr (EV == 6) {
    // y of the layer that owns the lane's 8 channels at the lane's pixel (linear pixel index: never the stride-2 walk).
    // Out-of-tile lanes read the segment's first row (valid, finite): their gradient is an exact zero and adds nothing.
#pragma unroll
    for (int A = 0; A < NA; ++A) {
      const int c0 = clane + 32 * A;
      const int sg = bw_seg(p, c0 < p.C_out ? c0 : 0);
      const unsigned short* yb = p.bw_y[sg] + (c0 < p.C_out ? c0 - p.bw_cb[sg] : 0);
      const long long ld = p.bw_ldy[sg];
#pragma unroll
      for (int bb = 0; bb < BCH; ++bb) {
        const int m = mlane + (b0 + bb) * 16;
        la[bb][A] = *reinterpret_cast<const u32x4*>(yb + ((m < p.M && c0 < p.C_out) ? m * ld : 0ll));
      }
    }
  }
  if constexpr (EV == 2 || EV == 4) {
    const __amdgpu_buffer_rsrc_t kr_ = make_rsrc(p.skip, p.skip ? p.skip_bytes : 0u);
    const __amdgpu_buffer_rsrc_t ar = make_rsrc(p.acc_src, p.acc_bytes);
    const bool do_acc = EV == 2 && p.accumulate, do_mask = EV == 2 && BMODE != 2 && p.skip != nullptr;     // (uniform)
    const bool do_bits = EV == 2 && (BMODE == 2 || (BMODE == 1 && p.bits != nullptr));        // (uniform; never with do_mask)
    const __amdgpu_buffer_rsrc_t br = make_rsrc(p.bits, p.bits ? p.bits_bytes : 0u);
#pragma unroll
    for (int bb = 0; bb < BCH; ++bb) {
      const int m = mlane + (b0 + bb) * 16;
      int img, pix;
      epi_pixel<SH>(p, m, img, pix);
#pragma unroll
      for (int A = 0; A < NA; ++A) {
        const int c0 = clane + 32 * A;
        const bool ok = m < p.M && c0 < p.C_out;
        const unsigned so = ok ? (unsigned)((img * p.skip_img_stride + pix * p.ld_skip + c0) * 2) : kOOB;
        if constexpr (EV == 4) {
          la[bb][A] = buf_load16(kr_, so);
        } else {
          const unsigned ao = ok ? (unsigned)((img * p.acc_img_stride + pix * p.ld_acc + c0) * 2) : kOOB;
          if (do_acc) la[bb][A] = buf_load16(ar, ao);
          if (do_mask) lb[bb][A] = buf_load16(kr_, so);
          // one byte = the lane's eight mask bits (out-of-range lanes read 0: their values are dropped anyway)
          if (do_bits) lb[bb][A].x = __builtin_amdgcn_raw_buffer_load_b8(br, ok ? m * p.bits_ld + (c0 >> 3) : (int)kOOB, 0, 0);
        }
      }
    }
  }
}

// Phase 2 for pixel blocks [b0, b0 + BCH): arithmetic and the 16-byte stores (fire and forget).
template <int EV, bool SH, int NI, int MI, int BCH, int BMODE = 1>
__device__ __forceinline__ void conv_epilogue_finish(const ConvK& p, const f32x4 (&acc)[NI][MI], const int mlane, const int clane,
                                                     const int b0, const u32x4 (&la)[BCH][NI / 2], const u32x4 (&lb)[BCH][NI / 2],
                                                     const float (&sh)[NI / 2][8], const float (&sc)[NI / 2][8],
                                                     float (&s1)[NI / 2][8], float (&s2)[NI / 2][8]) {
  constexpr int NA = NI / 2;
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(p.y, p.y_bytes);
  const bool do_acc = EV == 2 && p.accumulate, do_mask = EV == 2 && BMODE != 2 && p.skip != nullptr;     // (uniform)
  const bool do_bits = (EV == 2 && BMODE == 2) || ((EV == 2 || EV == 4) && BMODE == 1 && p.bits != nullptr);   // (uniform)
  const __amdgpu_buffer_rsrc_t br = make_rsrc(p.bits, p.bits ? p.bits_bytes : 0u);
  const int fch_ = (threadIdx.x >> 4) & 3;              // the lane's channel octet within a 32-channel block (clane = ... + 8 fch)
#pragma unroll
  for (int bb = 0; bb < BCH; ++bb) {
    const int b = b0 + bb;
    const int m = mlane + b * 16;
    int img, pix;
    epi_pixel<SH>(p, m, img, pix);
    unsigned bits_dw = 0;
#pragma unroll
    for (int A = 0; A < NA; ++A) {
      const int c0 = clane + 32 * A;
      const unsigned yo = (m < p.M && c0 < p.C_out) ? (unsigned)((img * p.y_img_stride + pix * p.ldy + c0) * 2) : kOOB;
      float v[8] = {acc[2 * A][b][0], acc[2 * A][b][1], acc[2 * A][b][2], acc[2 * A][b][3],
                    acc[2 * A + 1][b][0], acc[2 * A + 1][b][1], acc[2 * A + 1][b][2], acc[2 * A + 1][b][3]};
      if constexpr (EV == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[A][j] + sh[A][j];
      } else if constexpr (EV == 4) {
        const unsigned w[4] = {la[bb][A].x, la[bb][A].y, la[bb][A].z, la[bb][A].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * j] = bf2f(w[j] & 0xffffu) + p.rscale * (v[2 * j] + sh[A][2 * j]);
          v[2 * j + 1] = bf2f(w[j] >> 16) + p.rscale * (v[2 * j + 1] + sh[A][2 * j + 1]);
        }
      } else if constexpr (EV == 2) {
#pragma clang fp contract(off)          // the product is rounded on its own, as it was under the branch (no fma with the add below)
        const float rs = p.rscale != 0.f ? p.rscale : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= rs;
      } else {
        // (x * 1.0f is exact: one multiply for every value instead of a select per value around it)
        const float rs = p.rscale != 0.f ? p.rscale : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= rs;
      }
      if constexpr (EV == 2) {
        if (do_acc) {
          const unsigned w[4] = {la[bb][A].x, la[bb][A].y, la[bb][A].z, la[bb][A].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[2 * j] += bf2f(w[j] & 0xffffu); v[2 * j + 1] += bf2f(w[j] >> 16); }
        }
        if (do_mask) {                      // relu backward of the tensor this gradient belongs to
          const unsigned w[4] = {lb[bb][A].x, lb[bb][A].y, lb[bb][A].z, lb[bb][A].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!(bf2f(w[j] & 0xffffu) > 0.f)) v[2 * j] = 0.f;
            if (!(bf2f(w[j] >> 16) > 0.f)) v[2 * j + 1] = 0.f;
          }
        }
      }
      if constexpr (EV == 3 || EV == 4) {
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = relu_f(v[j]);
        }
      }
      unsigned q4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[j] = pack2bf(v[2 * j], v[2 * j + 1]);
      if constexpr (EV == 2) {
        if (do_bits) {                      // relu backward from the sign bits: bit 2j / 2j + 1 -> the halves of pair j
          const int byte = (int)lb[bb][A].x;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            q4[j] &= ((unsigned)__builtin_amdgcn_sbfe(byte, 2 * j, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(byte, 2 * j + 1, 1) << 16);
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{q4[0], q4[1], q4[2], q4[3]}, yr, (int)yo, 0, 0);
      if constexpr (EV == 4) {
        if (do_bits) {                      // (relu: the stored halves are >= 0 or -0; > 0 <=> non-zero below the sign bit)
          // min(half, 1) of both halves of a pair in one packed instruction: bits 0 / 16 of t[j]; the four pairs side by side
          // (bits 0, 2, 4, 6 and 16, 18, 20, 22), then the upper halves folded down beside the lower ones
          // (inline asm -- on values the VALU has just produced: the compiler turns the vector min into a compare and a select
          // per half)
          unsigned sidx = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned h;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(h) : "v"(q4[j] & 0x7fff7fffu), "s"(0x00010001u));
            sidx |= h << (2 * j);
          }
          const unsigned byte = (sidx | (sidx >> 15)) & 0xffu;
          // the four lanes of a pixel (lane, +16, +32, +48: fch 0..3) hold four CONSECUTIVE bytes of its row: gathered into
          // one dword in all of them (two row swaps, no LDS), and the lane with fch == A keeps the dword of channel block A
          // -- ONE 4-byte store per pixel block and 32 channels below, instead of a byte store per lane (64 single bytes
          // per instruction cost the residual launches of block17 3.9 us each, more than the data gradients gained)
          const unsigned dw = gather4_rows(byte);
          if (A == 0 || fch_ == A) bits_dw = dw;
        }
      }
      if constexpr (EV == 1) {
        // (out-of-tile lanes hold exact zeros -- zero-filled pixel rows / filter rows -- and add nothing)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (j & 1) ? bf_hi(q4[j >> 1]) : bf_lo(q4[j >> 1]); s1[A][j] += f; s2[A][j] += f * f; }
      }
      if constexpr (EV == 6) {
        // batch-norm backward sums of the STORED gradient: g = relu mask (y > thr; sh holds thr) ? da : 0; sum g, sum g y
        const unsigned w[4] = {la[bb][A].x, la[bb][A].y, la[bb][A].z, la[bb][A].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yv = bf2f((unsigned short)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu)));
          const float g = yv > sh[A][j] ? ((j & 1) ? bf_hi(q4[j >> 1]) : bf_lo(q4[j >> 1])) : 0.f;
          s1[A][j] += g;
          s2[A][j] += g * yv;
        }
      }
    }
    if constexpr (EV == 4) {
      if (do_bits) {
        // lane (pixel, fch) stores the dword of channel block A = fch: channels [32 fch, 32 fch + 32) from the wave's first
        const int cw = clane - 8 * fch_ + 32 * fch_;
        __builtin_amdgcn_raw_buffer_store_b32(bits_dw, br, (fch_ < NA && m < p.M && cw < p.C_out) ? m * p.bits_ld + (cw >> 3) : (int)kOOB, 0, 0);
      }
    }
  }
}

// per-channel scale / shift of the affine and residual epilogues, statistics sums cleared
template <int EV, int NA>
__device__ __forceinline__ void conv_epilogue_channels(const ConvK& p, const int clane, float (&sc)[NA][8], float (&sh)[NA][8],
                                                       float (&s1)[NA][8], float (&s2)[NA][8]) {
#pragma unroll
  for (int A = 0; A < NA; ++A)
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[A][j] = 1.f; sh[A][j] = 0.f; s1[A][j] = 0.f; s2[A][j] = 0.f; }
  if constexpr (EV == 6) {
#pragma unroll
    for (int A = 0; A < NA; ++A) {
      const int c0 = clane + 32 * A;
      if (c0 < p.C_out) {
        const int sg = bw_seg(p, c0);
        const float* t = p.bw_thr[sg] + (c0 - p.bw_cb[sg]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[A][j] = t[j];
      }
    }
  }
  if constexpr (EV == 3 || EV == 4) {
#pragma unroll
    for (int A = 0; A < NA; ++A) {
      const int c0 = clane + 32 * A;
      if (c0 < p.C_out) {                                    // C_out % 8 == 0 for bf16 outputs
        // two 16-byte reads per table (per-element reads compiled to a branch and an address computation each); the
        // tables are only 4-byte aligned (slices of the parameter buffer), which global memory reads allow
        if (p.scale) {
          const f32x4u u0 = *reinterpret_cast<const f32x4u*>(p.scale + c0), u1 = *reinterpret_cast<const f32x4u*>(p.scale + c0 + 4);
          sc[A][0] = u0.x; sc[A][1] = u0.y; sc[A][2] = u0.z; sc[A][3] = u0.w; sc[A][4] = u1.x; sc[A][5] = u1.y; sc[A][6] = u1.z; sc[A][7] = u1.w;
        }
        if (p.shiftv) {
          const f32x4u u0 = *reinterpret_cast<const f32x4u*>(p.shiftv + c0), u1 = *reinterpret_cast<const f32x4u*>(p.shiftv + c0 + 4);
          sh[A][0] = u0.x; sh[A][1] = u0.y; sh[A][2] = u0.z; sh[A][3] = u0.w; sh[A][4] = u1.x; sh[A][5] = u1.y; sh[A][6] = u1.z; sh[A][7] = u1.w;
        }
      }
    }
  }
}

// Pixel blocks [B0, B1) in chunks of BCH: reads of a chunk issued back to back, then its arithmetic and stores.  The
// empty asm keeps the compiler from hoisting the NEXT chunk's reads above this chunk (BCH bounds the registers the
// reads in flight take: hoisted all together they spilled).
template <int EV, bool SH, int NI, int MI, int B0, int B1, int BCH, int BMODE = 1>
__device__ __forceinline__ void conv_epilogue_range(const ConvK& p, const f32x4 (&acc)[NI][MI], const int mlane, const int clane,
                                                    const float (&sh)[NI / 2][8], const float (&sc)[NI / 2][8],
                                                    float (&s1)[NI / 2][8], float (&s2)[NI / 2][8]) {
  static_assert((B1 - B0) % BCH == 0 && B0 >= 0 && B1 <= MI, "chunks");
  constexpr int NA = NI / 2;
#pragma unroll
  for (int b0 = B0; b0 < B1; b0 += BCH) {
    u32x4 la[BCH][NA], lb[BCH][NA];
    conv_epilogue_issue_reads<EV, SH, NA, BCH, BMODE>(p, mlane, clane, b0, la, lb);
    conv_epilogue_finish<EV, SH, NI, MI, BCH, BMODE>(p, acc, mlane, clane, b0, la, lb, sh, sc, s1, s2);
    if (b0 + BCH < B1) asm volatile("" ::: "memory");
  }
}

template <int EV, bool SH, int NI, int MI, int BCH = MI>
__device__ __forceinline__ void conv_epilogue_direct(const ConvK& p, const f32x4 (&acc)[NI][MI], const int mlane, const int clane,
                                                     float (&s1)[NI / 2][8], float (&s2)[NI / 2][8]) {
  static_assert(NI % 2 == 0 && EV != 5 && MI % BCH == 0, "wave tile: a multiple of 32 output channels; float32 heads are handled by the caller");
  constexpr int NA = NI / 2;
  float sc[NA][8], sh[NA][8];
  conv_epilogue_channels<EV, NA>(p, clane, sc, sh, s1, s2);
  conv_epilogue_range<EV, SH, NI, MI, 0, MI, BCH>(p, acc, mlane, clane, sh, sc, s1, s2);
}

}  // namespace

#include "fused_bn.h"
