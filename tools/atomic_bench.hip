// Micro-benchmark: cost of device-scope float/double atomics when `blocks` workgroups all add into the
// SAME `addrs` addresses (the batch-norm statistics pattern: many pixel tiles -> C channel sums).
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/atomic_bench.hip -o gpurun_out/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T>
__global__ void __launch_bounds__(256) contend(T* acc, int addrs, int per_thread) {
  // 256 threads: thread t adds to address (t + i*256) % addrs
  for (int i = 0; i < per_thread; ++i) atomicAdd(acc + (threadIdx.x + i * 256) % addrs, (T)1);
}

template <typename T>
__global__ void __launch_bounds__(256) baseline(T* acc, int addrs, int per_thread) {
  if (blockIdx.x == 0x7fffffff) acc[0] = 0;
}

template <typename T>
static void run(const char* name, int blocks, int addrs, int per_thread) {
  T* acc;
  hipMalloc(&acc, sizeof(T) * addrs);
  hipMemset(acc, 0, sizeof(T) * addrs);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(contend<T>, dim3(blocks), dim3(256), 0, 0, acc, addrs, per_thread);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(contend<T>, dim3(blocks), dim3(256), 0, 0, acc, addrs, per_thread);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(baseline<T>, dim3(blocks), dim3(256), 0, 0, acc, addrs, per_thread);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms0;
  hipEventElapsedTime(&ms0, e0, e1);
  std::vector<T> h(addrs);
  hipMemcpy(h.data(), acc, sizeof(T) * addrs, hipMemcpyDeviceToHost);
  const double total = (double)blocks * 256 * per_thread;
  const double expect = total * 12 / addrs;
  printf("%s blocks=%6d addrs=%5d per_thread=%d : %8.1f us/launch (empty %5.1f us)  %7.2f G atomics/s  per-address %6.0f  sum0=%g (expect %g)\n",
         name, blocks, addrs, per_thread, ms * 1000 / reps, ms0 * 1000 / reps, total / (ms / reps * 1e-3) * 1e-9,
         total / addrs, (double)h[0], expect);
  hipFree(acc);
}

int main() {
  const int blocks[] = {64, 512, 4096, 16384};
  const int addrs[] = {64, 256, 640, 4096};
  for (int b : blocks)
    for (int a : addrs) {
      run<float>("f32", b, a, 1);
      run<double>("f64", b, a, 1);
    }
  return 0;
}
