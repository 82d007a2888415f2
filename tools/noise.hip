// Memory-system noise for tools/side_stream_stress.py (test infrastructure, not part of libmbx): kernels that keep
// HBM, the L2 -> LDS path and the memory-side atomic units saturated from a SECOND stream while the training step runs,
// the way a 240 MB RCCL all-reduce does in data-parallel runs.  Built on demand: hipcc --offload-arch=gfx950 -shared.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// (1) streaming copy: every block walks its share of a buffer far larger than the Infinity Cache with 16-byte
// accesses, `iters` times -- HBM read + write bandwidth
__global__ void __launch_bounds__(256) noise_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int iters) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (int it = 0; it < iters; ++it)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
      u32x4 v = __builtin_nontemporal_load(src + i);
      v.x += (unsigned)it;
      __builtin_nontemporal_store(v, dst + i);
    }
}

// (2) atomics storm: float adds scattered over a table (one address per lane, a different line per lane), device scope
__global__ void __launch_bounds__(256) noise_atomics(float* __restrict__ table, unsigned mask, int iters) {
  unsigned x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    atomicAdd(table + ((x >> 7) & mask), 1.0f);
  }
}

// (3) L2 -> LDS hammer: LDS-DMA (buffer_load ... lds, 16 B per lane) of an L2-resident table into 32 KB of LDS per
// block, over and over -- the operand path of the convolution kernels, and LDS capacity next to their blocks
__global__ void __launch_bounds__(256) noise_lds_dma(const void* __restrict__ table, unsigned bytes, int iters, unsigned* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2048];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(table), (short)0, (int)bytes, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned off = ((blockIdx.x * 256u + threadIdx.x) * 16u) % bytes;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (p * 4 + wave) * 64), 16,
                                               (int)((off + p * 4096u) % bytes), 0, 0, 0);
    }
    off = (off + 32768u + 16u * lane) % bytes;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && lds[5].x == 0xdeadbeefu) sink[0] = 1;       // keep the LDS image alive
}

extern "C" int noise_launch(void* src, void* dst, size_t bytes, float* table, unsigned table_elems_pow2, const void* l2_table,
                            unsigned l2_bytes, unsigned* sink, int what, int iters, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (what & 1) hipLaunchKernelGGL(noise_copy, dim3(1024), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, bytes / 16, iters);
  if (what & 2) hipLaunchKernelGGL(noise_atomics, dim3(1024), dim3(256), 0, s, table, table_elems_pow2 - 1, iters * 64);
  if (what & 4) hipLaunchKernelGGL(noise_lds_dma, dim3(1024), dim3(256), 0, s, l2_table, l2_bytes, iters * 64, sink);
  return (int)hipGetLastError();
}
