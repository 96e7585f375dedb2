// What v_permlane32_swap does on gfx950: prints both results per lane for a = lane, b = 100 + lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  p[threadIdx.x] = r[0];
  p[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 = %3u  r1 = %3u\n", l, h[l], h[64 + l]);
  return 0;
}
