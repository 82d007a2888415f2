// Which XCD does block b run on?  Reads HW_REG_XCC_ID in every block of a 1024-block grid and prints the histogram
// of (blockIdx % 8, xcc id) -- checks the round-robin placement the XCD-aware remaps assume.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
}
int main() {
  const int n = 1024;
  int* d; hipMalloc(&d, n * 4);
  int h[n];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(n), dim3(512), 100 * 1024, 0, d);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int hist[8][16] = {};
    for (int b = 0; b < n; ++b) hist[b & 7][h[b] & 15]++;
    for (int r = 0; r < 8; ++r) { printf("b%%8=%d:", r); for (int x = 0; x < 16; ++x) if (hist[r][x]) printf(" xcc%d:%d", x, hist[r][x]); printf("\n"); }
  }
  return 0;
}
