// L2 -> LDS operand-feed micro-benchmark (gfx950).  The convolution kernels are bound by how fast a CU can pull operand
// tiles that sit in L2 into LDS (~45 GB/s per CU measured inside the kernels).  This program asks
// WHY: is it the bytes in flight (Little's law against the L2 latency, capped by how much LDS the ring can spend) or a
// throughput limit of the path?  Every wave streams a region that stays L2-resident through one of two paths:
//   mode 0  buffer_load ... lds (LDS-DMA, 16 B per lane) into a ring of DEPTH stages of G KB per wave;
//   mode 1  global_load_dwordx4 into REGISTERS (DEPTH x G x 16 B per lane in flight), ds_write_b128 into a 2-stage ring
//           once landed -- the bytes in flight live in the register file instead of LDS.
// Prints GB/s per CU for a sweep of waves per workgroup, stages in flight and stage size.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/feed_bench tools/feed_bench.hip ; run: tools/feed_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// region: bytes every XCD's workgroups cycle through (L2 resident); all sizes in bytes
template <int MODE, int DEPTH, int G>
__global__ void __launch_bounds__(1024)
feed_kernel(const unsigned char* __restrict__ src, unsigned region, int iters, unsigned* __restrict__ sink, int pat, int row_stride) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const unsigned xcd = blockIdx.x & 7;
  const unsigned char* base = src + (size_t)xcd * region;
  // every wave walks the region in 1 KB steps from its own start, wrapping
  unsigned pos = (unsigned)(((unsigned long long)((blockIdx.x >> 3) * waves + wave) * (region / 512u)) % region) & ~1023u;
  unsigned acc = 0;
  // lane -> byte offset inside one wave-level load: 8 rows x 128 B (the conv tiles' shape); pat bit 0 = the XOR chunk
  // swizzle of the LDS image applied on the source, row_stride = bytes between rows (128 = contiguous 1 KB)
  const int lrow = lane >> 3, lchunk = (pat & 1) ? ((lane & 7) ^ (lrow & 7)) : (lane & 7);
  const unsigned loff = (unsigned)(lrow * row_stride + lchunk * 16);
  const unsigned adv = (unsigned)(8 * row_stride);
  constexpr int kStage = G * 64;                               // u32x4 per wave per stage
  if constexpr (MODE == 0) {
    u32x4* ring = lds + (size_t)wave * DEPTH * kStage;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(base, region);
    auto issue = [&](int stage) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(ring + stage * kStage + g * 64), 16,
                                                 pos + loff, 0, 0, 0);
        pos += adv; if (pos + adv >= region) pos = 0;
      }
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) issue(s);
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) {
        issue((s + DEPTH - 1) % DEPTH);
        wait_vmcnt<(DEPTH - 1) * G>();
        const u32x4 v = ring[s * kStage + lane];               // touch the landed stage
        acc ^= v.x;
      }
    }
  } else {
    u32x4* ring = lds + (size_t)wave * 2 * kStage;
    u32x4 regs[DEPTH][G];
    auto issue = [&](int stage) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        regs[stage][g] = *reinterpret_cast<const u32x4*>(base + pos + loff);
        pos += adv; if (pos + adv >= region) pos = 0;
      }
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) issue(s);
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) {
        issue((s + DEPTH - 1) % DEPTH);
        wait_vmcnt<(DEPTH - 1) * G>();
#pragma unroll
        for (int g = 0; g < G; ++g) ring[(s & 1) * kStage + g * 64 + lane] = regs[s][g];
        acc ^= regs[s][0].x;
      }
    }
  }
  wait_vmcnt<0>();
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int DEPTH, int G>
static void run(const unsigned char* src, unsigned region, unsigned* sink, int waves, int blocks, int pat = 0, int row_stride = 128) {
  const int iters = 4096 / G / DEPTH * DEPTH;
  const size_t lds = (size_t)waves * (MODE == 0 ? DEPTH : 2) * G * 1024;
  if (lds > 160 * 1024) return;
  hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<MODE, DEPTH, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((feed_kernel<MODE, DEPTH, G>), dim3(blocks), dim3(waves * 64), lds, 0, src, region, iters, sink, pat, row_stride);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
  }
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * waves * iters * G * 1024.0;
  const double inflight_kb = (double)waves * (DEPTH - 1) * G * (blocks > 256 ? 2 : 1);
  printf("%-9s %s rows %4d B apart  waves %2d x %d blk/CU  depth %d  stage %2d KB/wave  in flight %5.0f KB/CU  LDS %3zu KB/blk : %7.1f GB/s per CU  %6.2f TB/s chip\n",
         MODE == 0 ? "LDS-DMA" : "registers", (pat & 1) ? "swizzled" : "linear  ", row_stride, waves, blocks / 256, DEPTH, G, inflight_kb, lds / 1024, bytes / (ms * 1e-3) / 1e9 / 256, bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  // region per XCD: 2 MB stays in the 4 MB L2; 16 MB (128 MB in all) streams from the Infinity Cache; 96 MB from HBM
  const unsigned region = (unsigned)(argc > 1 ? atoi(argv[1]) : 2) << 20;
  printf("region %u MB per XCD\n", region >> 20);
  unsigned char* src; unsigned* sink;
  hipMalloc(&src, (size_t)region * 8);
  hipMalloc(&sink, 64);
  hipMemset(src, 1, (size_t)region * 8);
  if (argc > 2 && atoi(argv[2]) == 2) {                        // access-pattern sweep: 8 rows x 128 B per wave-level load
    for (int stride : {128, 256, 768, 2176, 4160})
      for (int pat : {0, 1}) {
        run<0, 3, 4>(src, region, sink, 8, 256, pat, stride);
        run<1, 3, 4>(src, region, sink, 8, 256, pat, stride);
      }
    return 0;
  }
  for (int blocks : {256, 512}) {
    for (int waves : {8, 16}) {
      if (blocks == 512 && waves == 16) continue;
      if (argc > 2 && blocks == 512) continue;
      run<0, 2, 1>(src, region, sink, waves, blocks);
      run<0, 2, 2>(src, region, sink, waves, blocks);
      run<0, 2, 4>(src, region, sink, waves, blocks);
      run<0, 3, 2>(src, region, sink, waves, blocks);
      run<0, 3, 4>(src, region, sink, waves, blocks);
      run<0, 4, 2>(src, region, sink, waves, blocks);
      run<0, 5, 2>(src, region, sink, waves, blocks);
      run<0, 5, 1>(src, region, sink, waves, blocks);
      run<0, 9, 1>(src, region, sink, waves, blocks);
      run<1, 2, 1>(src, region, sink, waves, blocks);
      run<1, 2, 2>(src, region, sink, waves, blocks);
      run<1, 2, 4>(src, region, sink, waves, blocks);
      run<1, 3, 2>(src, region, sink, waves, blocks);
      run<1, 3, 4>(src, region, sink, waves, blocks);
      run<1, 5, 2>(src, region, sink, waves, blocks);
      run<1, 5, 4>(src, region, sink, waves, blocks);
      run<1, 9, 2>(src, region, sink, waves, blocks);
      run<1, 9, 4>(src, region, sink, waves, blocks);
      run<1, 13, 2>(src, region, sink, waves, blocks);
    }
  }
  return 0;
}
