// What read rate does HBM deliver for once-read chunks in random order (the access pattern of
// paged attention: a (block, head) of an fp8 KV cache is a 2 KiB run)?  Every wave walks its own
// list of chunk numbers (a random permutation, read from memory like a block table) and loads each
// chunk with 16-byte nontemporal loads, 16 KiB in flight per wave, 8 waves per CU-slot.  Build:
//   hipcc -O3 --offload-arch=gfx950 read_bw_chunks.hip -o read_bw_chunks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// CH: chunk bytes (1024 .. 16384).  A wave-instruction loads 1 KiB; a wave keeps 16 of them in flight.
template <int CH>
__global__ __launch_bounds__(256, 2) void chunk_kernel(const unsigned char* __restrict__ src,
                                                      const int* __restrict__ perm, unsigned* out,
                                                      int chunks_per_wave) {
  constexpr int PIECES = CH / 1024;                 // wave-instructions per chunk
  constexpr int CPI = PIECES >= 16 ? 1 : 16 / PIECES;   // chunks per iteration (16 KiB in flight)
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int* mine = perm + wave * chunks_per_wave;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < chunks_per_wave; i += CPI) {
    u32x4 v[CPI * PIECES];
#pragma unroll
    for (int c = 0; c < CPI; ++c) {
      const long id = mine[i + c];
      const u32x4* p = reinterpret_cast<const u32x4*>(src + id * CH) + lane;
#pragma unroll
      for (int j = 0; j < PIECES; ++j) v[c * PIECES + j] = __builtin_nontemporal_load(p + j * 64);
    }
#pragma unroll
    for (int j = 0; j < CPI * PIECES; ++j) acc ^= v[j];
  }
  unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u) out[blockIdx.x] = r;
}

template <int CH>
void run(const unsigned char* src, int* perm_d, unsigned* out, long bytes, bool shuffled) {
  const long nchunks = bytes / CH;
  std::vector<int> perm(nchunks);
  for (long i = 0; i < nchunks; ++i) perm[i] = (int)i;
  if (shuffled) { std::mt19937 g(7); std::shuffle(perm.begin(), perm.end(), g); }
  CK(hipMemcpy(perm_d, perm.data(), nchunks * 4, hipMemcpyHostToDevice));
  const int wgs = 2048 * 4;
  const int cpw = (int)(nchunks / (wgs * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(chunk_kernel<CH>, dim3(wgs), dim3(256), 0, 0, src, perm_d, out, cpw);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(chunk_kernel<CH>, dim3(wgs), dim3(256), 0, 0, src, perm_d, out, cpw);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("chunks of %5d B, %s order: %.1f GB/s\n", CH, shuffled ? "random" : "linear",
         (double)cpw * wgs * 4 * CH * 5.0 / (ms * 1e-3) / 1e9);
}

int main() {
  const long bytes = 8L << 30;
  unsigned char* src; unsigned* out; int* perm_d;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&perm_d, bytes / 1024 * 4));
  CK(hipMemset(src, 1, bytes));
  for (int sh = 0; sh < 2; ++sh) {
    run<1024>(src, perm_d, out, bytes, sh);
    run<2048>(src, perm_d, out, bytes, sh);
    run<4096>(src, perm_d, out, bytes, sh);
    run<16384>(src, perm_d, out, bytes, sh);
  }
  return 0;
}
