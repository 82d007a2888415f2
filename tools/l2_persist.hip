// Does an XCD's L2 keep what a kernel wrote for the NEXT kernel (gfx950, 8 XCDs, 4 MB of L2 each)?
// Kernel W: workgroup b writes chunk b (64 KB) of a 16 MB buffer.  Kernel R: workgroup b reads chunk (b + shift) % nblk:
// shift = 0 -> the chunk its own XCD wrote (workgroups are dealt to XCDs round-robin: XCD = b % 8), shift = 1 -> a chunk
// another XCD wrote, shift = 8 -> another workgroup's chunk of the SAME XCD.  If the L2 survives the kernel boundary, shift 0 / 8
// read at the L2 rate and shift 1 at the fabric (Infinity Cache) rate.
// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/l2_persist tools/l2_persist.hip && tools/_bin/l2_persist
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int kChunk = 64 * 1024;

__global__ void __launch_bounds__(256) wkernel(u32x4* buf, unsigned v) {
  u32x4* p = buf + (size_t)blockIdx.x * (kChunk / 16);
  for (int i = threadIdx.x; i < kChunk / 16; i += 256) p[i] = u32x4{v, v + i, v, v};
}
__global__ void __launch_bounds__(256) rkernel(const u32x4* buf, int shift, int nblk, unsigned* sink) {
  const u32x4* p = buf + (size_t)((blockIdx.x + shift) % nblk) * (kChunk / 16);
  unsigned a = 0;
#pragma unroll 4
  for (int i = threadIdx.x; i < kChunk / 16; i += 256) { const u32x4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (a == 0x12345678u) sink[0] = a;
}
int main() {
  for (int mb : {8, 16, 32, 64}) {
    const int nblk = mb * 1024 * 1024 / kChunk;
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, (size_t)nblk * kChunk); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shift : {0, 1, 8, 4, 0, 1}) {
      float best = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(wkernel, dim3(nblk), dim3(256), 0, 0, buf, (unsigned)rep);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rkernel, dim3(nblk), dim3(256), 0, 0, buf, shift, nblk, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%3d MB  shift %d: read %.1f us  %.2f TB/s\n", mb, shift, best * 1e3, mb * 1.048576 / best * 1e-3 * 1e0);
    }
    hipFree(buf); hipFree(sink);
  }
  return 0;
}
