// Throughput of device-scope fp32 atomics to DISTINCT addresses for two lane layouts of one wave instruction:
//   A: 4 rows x 16 consecutive floats (the MFMA D-fragment layout the weight-gradient epilogue uses today)
//   B: 64 consecutive floats (two full 128-byte lines)
// Every block adds a 128 x 128 fp32 tile into a [rows][ld] matrix; `share` blocks hit the same tile.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/atomic_bench2.hip -o tools/_bin/atomic_bench2
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LAYOUT>
__global__ void __launch_bounds__(256) tile_atomics(float* dw, int ld, int tiles_k, int share) {
  const int tile = blockIdx.x / share;
  const int tn = tile / tiles_k, tk = tile % tiles_k;
  float* base = dw + (size_t)tn * 128 * ld + tk * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (LAYOUT == 0) {
    const int wn = wave & 1, wk = wave >> 1;
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b)
        for (int r = 0; r < 4; ++r) {
          const int n = wn * 64 + a * 16 + (lane >> 4) * 4 + r, k = wk * 64 + b * 16 + (lane & 15);
          atomicAdd(base + (size_t)n * ld + k, 1.0f);
        }
  } else {
    // 64 instructions per wave, each 64 consecutive floats of one row: wave w owns rows w*32 .. w*32+31
    for (int i = 0; i < 64; ++i) {
      const int n = wave * 32 + (i >> 1), k = (i & 1) * 64 + lane;
      atomicAdd(base + (size_t)n * ld + k, 1.0f);
    }
  }
}

template <int LAYOUT>
static void run(int tiles_n, int tiles_k, int share) {
  const int ld = tiles_k * 128;
  float* dw;
  hipMalloc(&dw, sizeof(float) * (size_t)tiles_n * 128 * ld);
  hipMemset(dw, 0, sizeof(float) * (size_t)tiles_n * 128 * ld);
  const int blocks = tiles_n * tiles_k * share;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(tile_atomics<LAYOUT>, dim3(blocks), dim3(256), 0, 0, dw, ld, tiles_k, share);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(tile_atomics<LAYOUT>, dim3(blocks), dim3(256), 0, 0, dw, ld, tiles_k, share);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  float h = 0;
  hipMemcpy(&h, dw + 5 * ld + 77, 4, hipMemcpyDeviceToHost);
  const double bytes = (double)blocks * 128 * 128 * 4;
  printf("layout %c tiles %dx%d share %d (%d blocks): %7.1f us/launch  %6.2f TB/s of atomics  check=%g (expect %d)\n",
         LAYOUT ? 'B' : 'A', tiles_n, tiles_k, share, blocks, ms * 1000 / reps, bytes / (ms / reps * 1e-3) * 1e-12, h, share * 12);
  hipFree(dw);
}

int main() {
  run<0>(3, 9, 9); run<1>(3, 9, 9);
  run<0>(9, 3, 9); run<1>(9, 3, 9);
  run<0>(2, 7, 18); run<1>(2, 7, 18);
  run<0>(16, 16, 1); run<1>(16, 16, 1);
  return 0;
}
