// Measured read-only HBM bandwidth of this GPU: every workgroup streams a contiguous slice with
// 16-byte loads (8 in flight per lane, default cache policy and nontemporal) and folds it into one word.  Build:
//   hipcc -O3 --offload-arch=gfx950 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, unsigned* out,
                                                   long vecs_per_wg) {
  const u32x4* p = src + (long)blockIdx.x * vecs_per_wg;
  u32x4 acc = {0, 0, 0, 0};
  for (long i = threadIdx.x; i < vecs_per_wg; i += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(p + i + j * 256) : p[i + j * 256];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
  }
  unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u) out[blockIdx.x] = r;  // keeps the loads alive, practically never taken
}

int main() {
  const long bytes = 8L << 30;
  u32x4* src; unsigned* out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(src, 1, bytes));
  for (int nt = 0; nt < 2; ++nt)
    for (int wgs : {2048, 8192, 32768, 131072}) {
      const long vecs_per_wg = bytes / 16 / wgs;
      auto fn = nt ? read_kernel<true> : read_kernel<false>;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fn, dim3(wgs), dim3(256), 0, 0, src, out, vecs_per_wg);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fn, dim3(wgs), dim3(256), 0, 0, src, out, vecs_per_wg);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("read-only 8 GiB, %s loads, %6d workgroups: %.1f GB/s\n", nt ? "nontemporal" : "default", wgs,
             bytes * 5.0 / (ms * 1e-3) / 1e9);
    }
  return 0;
}
