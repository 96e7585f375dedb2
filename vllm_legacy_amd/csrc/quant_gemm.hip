// int4 weight-only quantisation for gfx950: AWQ and GPTQ (4 bit) dequantisation kernels, the
// exllama weight shuffle, and the dequant-GEMM on the matrix cores.
//
// Weight formulas (bit-exact with the reference's fp16 arithmetic, because q - z is an exact
// small integer and the single fp16 multiply equals the fp32 product rounded once):
//   AWQ : W[k, 8j+i] = (nib(qweight[k, j], P[i]) - nib(qzeros[k/G, j], P[i])) * scales[k/G, 8j+i],
//         P = {0,4,1,5,2,6,3,7}   (csrc/quantization/awq/dequantize.cuh:41-61,
//                                  csrc/quantization/awq/gemm_kernels.cu:395-424)
//   GPTQ: W[k, n] = (nib(qweight[k/8, n], k%8) - (nib(qzeros[g, n/8], n%8) + 1)) * scales[g, n]
//                                 (csrc/quantization/gptq/q_gemm.cu:1386-1420); after
//         gptq_shuffle the nibble of row k%8 = i sits at position (i odd ? 4 + i/2 : i/2)
//                                 (csrc/quantization/gptq/qdq_4.cuh:16-29)
#include "vla_common.h"

namespace vla {
namespace {

__device__ __forceinline__ uint16_t f16_bits(float v) {
  return __builtin_bit_cast(uint16_t, (_Float16)v);
}

// ------------------------------------------------------------------ AWQ dequantise
// one lane per packed word: 4 + 4 + 16 bytes in, 16 bytes out, all coalesced along out_c
__global__ __launch_bounds__(256) void awq_dequantize_kernel(
    const uint32_t* __restrict__ qweight, const _Float16* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, _Float16* __restrict__ out, int in_c, int qoc,
    int group) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= qoc) return;
  const int g = k / group;
  const uint32_t qw = qweight[(int64_t)k * qoc + j];
  const uint32_t qz = qzeros[(int64_t)g * qoc + j];
  const u32x4 sv = *reinterpret_cast<const u32x4*>(scales + ((int64_t)g * qoc + j) * 8);
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pos = 4 * ((i >> 1) + 4 * (i & 1));  // P[i] * 4
    const float w = (float)((int)((qw >> pos) & 15u) - (int)((qz >> pos) & 15u));
    const uint32_t sw = sv[i >> 1];
    const float s = (float)__builtin_bit_cast(_Float16, (uint16_t)((i & 1) ? (sw >> 16) : (sw & 0xffff)));
    const uint32_t h = f16_bits(w * s);
    if (i & 1) o[i >> 1] |= h << 16;
    else o[i >> 1] = h;
  }
  *reinterpret_cast<u32x4*>(out + ((int64_t)k * qoc + j) * 8) = o;
}

// ------------------------------------------------------------------ GPTQ dequantise
// one lane per packed word (8 rows of one column); stores are 128-byte rows across the wave
template <bool SHUFFLED>
__global__ __launch_bounds__(256) void gptq_dequantize_kernel(
    const uint32_t* __restrict__ qweight, const uint32_t* __restrict__ qzeros,
    const _Float16* __restrict__ scales, const int* __restrict__ g_idx,
    _Float16* __restrict__ out, int size_k, int size_n, int group_size) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int k8 = blockIdx.y;
  if (n >= size_n) return;
  const uint32_t qw = qweight[(int64_t)k8 * size_n + n];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k8 * 8 + i;
    const int g = g_idx ? g_idx[k] : k / group_size;
    const int pos = SHUFFLED ? ((i & 1) ? 4 + (i >> 1) : (i >> 1)) : i;
    const uint32_t qz = qzeros[(int64_t)g * (size_n >> 3) + (n >> 3)];
    const int w = (int)((qw >> (4 * pos)) & 15u) - ((int)((qz >> (4 * (n & 7))) & 15u) + 1);
    const float s = (float)scales[(int64_t)g * size_n + n];
    out[(int64_t)k * size_n + n] = (_Float16)((float)w * s);
  }
}

// ------------------------------------------------------------------ GPTQ shuffle
__global__ void gptq_make_sequential_kernel(const uint32_t* __restrict__ w,
                                            uint32_t* __restrict__ w_new,
                                            const int* __restrict__ q_perm, int size_n) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (n >= size_n) return;
  uint32_t dst = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int src_row = q_perm[r * 8 + i];
    const uint32_t v = w[(int64_t)(src_row >> 3) * size_n + n];
    dst |= ((v >> (4 * (src_row & 7))) & 15u) << (4 * i);
  }
  w_new[(int64_t)r * size_n + n] = dst;
}

__global__ void gptq_shuffle_kernel(uint32_t* __restrict__ w, int64_t words) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  const uint32_t qa = w[i];
  uint32_t qb = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qb |= ((qa >> (8 * j)) & 15u) << (4 * j);
    qb |= ((qa >> (8 * j + 4)) & 15u) << (4 * j + 16);
  }
  w[i] = qb;
}

// ------------------------------------------------------------------ dequant-GEMM
// out[M, N] (fp16) = x[M, K] (fp16) . W[K, N], W dequantised on the fly into LDS as W^T tiles so
// that every MFMA operand is one 16-byte read; fp32 accumulation, one rounding at the end.
// Workgroup tile BM x BN, 4 waves; each wave owns BM/4 rows.
enum { kAwq = 0, kGptq = 1, kGptqShuffled = 2 };

constexpr int BN = 64;   // output columns per workgroup
constexpr int BK = 32;   // contraction depth per step = one MFMA
constexpr int WT_PITCH = BK + 8;  // fp16 elements per W^T row in LDS (80 bytes: 16-byte aligned)

template <int MODE, int MT>  // MT = 16-row MFMA tiles per wave
__global__ __launch_bounds__(256) void q4_gemm_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qweight,
    const uint32_t* __restrict__ qzeros, const _Float16* __restrict__ scales,
    const int* __restrict__ g_idx_or_perm, _Float16* __restrict__ out, int m, int size_n,
    int size_k, int group_size) {
  __shared__ __attribute__((aligned(16))) _Float16 wt[BN * WT_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * (4 * MT * 16) + wave * (MT * 16);

  f32x4 acc[MT][BN / 16];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < BN / 16; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < size_k; k0 += BK) {
    // ---- dequantise W[k0 .. k0+31][n0 .. n0+63] into wt[n][k]
    if constexpr (MODE == kAwq) {
      // 32 rows x 8 words = 256 words, one per thread
      const int kk = threadIdx.x >> 3, j = threadIdx.x & 7;
      const int k = k0 + kk, grp = k / group_size, qoc = size_n >> 3, jw = (n0 >> 3) + j;
      const uint32_t qw = qweight[(int64_t)k * qoc + jw];
      const uint32_t qz = qzeros[(int64_t)grp * qoc + jw];
      const _Float16* sp = scales + (int64_t)grp * size_n + n0 + 8 * j;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pos = 4 * ((i >> 1) + 4 * (i & 1));
        const float w = (float)((int)((qw >> pos) & 15u) - (int)((qz >> pos) & 15u));
        wt[(8 * j + i) * WT_PITCH + kk] = (_Float16)(w * (float)sp[i]);
      }
    } else {
      // 4 packed rows x 64 columns = 256 words, one per thread: 8 k of one column
      const int r = threadIdx.x >> 6, nn = threadIdx.x & 63;
      const int n = n0 + nn, k8 = (k0 >> 3) + r;
      const uint32_t qw = qweight[(int64_t)k8 * size_n + n];
      u32x4 packed;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k8 * 8 + i;
        int grp;
        if (MODE == kGptq) grp = g_idx_or_perm ? g_idx_or_perm[k] : k / group_size;
        else grp = k / group_size;
        const int pos = (MODE == kGptqShuffled) ? ((i & 1) ? 4 + (i >> 1) : (i >> 1)) : i;
        const uint32_t qz = qzeros[(int64_t)grp * (size_n >> 3) + (n >> 3)];
        const int w = (int)((qw >> (4 * pos)) & 15u) - ((int)((qz >> (4 * (n & 7))) & 15u) + 1);
        const uint32_t h = f16_bits((float)w * (float)scales[(int64_t)grp * size_n + n]);
        if (i & 1) packed[i >> 1] |= h << 16;
        else packed[i >> 1] = h;
      }
      *reinterpret_cast<u32x4*>(&wt[nn * WT_PITCH + 8 * r]) = packed;
    }
    __syncthreads();

    // ---- operands: A = x rows (global, L2 resident), B = W^T rows (LDS)
    u32x4 bf[BN / 16];
#pragma unroll
    for (int b = 0; b < BN / 16; ++b)
      bf[b] = *reinterpret_cast<const u32x4*>(&wt[(16 * b + c) * WT_PITCH + 8 * g]);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int row = m0 + 16 * a + c;
      u32x4 af = {0, 0, 0, 0};
      if (row < m) {
        if (MODE == kGptqShuffled && g_idx_or_perm) {
          // act-order: x columns are gathered through the permutation (q_gemm.cu:224-227)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const uint32_t lo = __builtin_bit_cast(
                uint16_t, x[(int64_t)row * size_k + g_idx_or_perm[k0 + 8 * g + e]]);
            const uint32_t hi = __builtin_bit_cast(
                uint16_t, x[(int64_t)row * size_k + g_idx_or_perm[k0 + 8 * g + e + 1]]);
            af[e >> 1] = lo | (hi << 16);
          }
        } else {
          af = *reinterpret_cast<const u32x4*>(x + (int64_t)row * size_k + k0 + 8 * g);
        }
      }
#pragma unroll
      for (int b = 0; b < BN / 16; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af),
                                                           __builtin_bit_cast(f16x8, bf[b]),
                                                           acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- C: lane (g, c) holds column 16b + c, rows 4g .. 4g+3 of each tile
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < BN / 16; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + 16 * a + 4 * g + i;
        if (row < m) out[(int64_t)row * size_n + n0 + 16 * b + c] = (_Float16)acc[a][b][i];
      }
}

template <int MODE>
int launch_gemm(const void* x, const int32_t* qweight, const int32_t* qzeros, const void* scales,
                const int32_t* aux, void* out, int m, int n, int k, int group_size,
                hipStream_t s) {
  if (m <= 64) {
    dim3 grid(n / BN, (m + 63) / 64);
    hipLaunchKernelGGL((q4_gemm_kernel<MODE, 1>), grid, dim3(256), 0, s, (const _Float16*)x,
                       (const uint32_t*)qweight, (const uint32_t*)qzeros, (const _Float16*)scales,
                       aux, (_Float16*)out, m, n, k, group_size);
  } else {
    dim3 grid(n / BN, (m + 255) / 256);
    hipLaunchKernelGGL((q4_gemm_kernel<MODE, 4>), grid, dim3(256), 0, s, (const _Float16*)x,
                       (const uint32_t*)qweight, (const uint32_t*)qzeros, (const _Float16*)scales,
                       aux, (_Float16*)out, m, n, k, group_size);
  }
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

}  // namespace
}  // namespace vla

using namespace vla;

extern "C" int vla_awq_dequantize(const int32_t* qweight, const void* scales,
                                  const int32_t* qzeros, void* out, int in_c, int out_c, int group,
                                  void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (!qweight || !scales || !qzeros || !out) return VLA_ERR_ARG;
  if (in_c <= 0 || out_c <= 0 || out_c % 8 || group <= 0 || in_c % group) return VLA_ERR_QUANT_SHAPE;
  const int qoc = out_c / 8;
  dim3 grid((qoc + 255) / 256, in_c);
  hipLaunchKernelGGL(awq_dequantize_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)qweight, (const _Float16*)scales, (const uint32_t*)qzeros,
                     (_Float16*)out, in_c, qoc, group);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_awq_gemm(const void* x, const int32_t* qweight, const void* scales,
                            const int32_t* qzeros, void* out, int m, int in_c, int out_c,
                            int group, void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (!x || !qweight || !scales || !qzeros || !out || m < 0) return VLA_ERR_ARG;
  // the reference's constraints (gemm_kernels.cu:515-522)
  if (out_c % 64 || out_c % 8 || group <= 0 || group % 32 || out_c % group || in_c % group)
    return VLA_ERR_QUANT_SHAPE;
  if (m == 0) return VLA_OK;
  return launch_gemm<kAwq>(x, qweight, qzeros, scales, nullptr, out, m, out_c, in_c, group,
                           (hipStream_t)stream);
}

extern "C" int vla_gptq_dequantize(const int32_t* qweight, const int32_t* qzeros,
                                   const void* scales, const int32_t* g_idx, void* out,
                                   int size_k, int size_n, int groups, int shuffled,
                                   void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (!qweight || !qzeros || !scales || !out) return VLA_ERR_ARG;
  if (size_k <= 0 || size_n <= 0 || size_k % 8 || size_n % 8 || groups <= 0 || size_k % groups)
    return VLA_ERR_QUANT_SHAPE;
  dim3 grid((size_n + 255) / 256, size_k / 8);
  hipStream_t s = (hipStream_t)stream;
  if (shuffled)
    hipLaunchKernelGGL(gptq_dequantize_kernel<true>, grid, dim3(256), 0, s,
                       (const uint32_t*)qweight, (const uint32_t*)qzeros, (const _Float16*)scales,
                       g_idx, (_Float16*)out, size_k, size_n, size_k / groups);
  else
    hipLaunchKernelGGL(gptq_dequantize_kernel<false>, grid, dim3(256), 0, s,
                       (const uint32_t*)qweight, (const uint32_t*)qzeros, (const _Float16*)scales,
                       g_idx, (_Float16*)out, size_k, size_n, size_k / groups);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_gptq_gemm(const void* a, const int32_t* qweight, const int32_t* qzeros,
                             const void* scales, const int32_t* b_g_idx, void* out, int m,
                             int size_n, int size_k, int groups, int use_exllama, int bit,
                             void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (!a || !qweight || !qzeros || !scales || !out || m < 0) return VLA_ERR_ARG;
  if (bit != 4) return VLA_ERR_QUANT_SHAPE;  // 2/3/8-bit GPTQ are outside this backend
  if (size_n % 64 || size_k % 32 || groups <= 0 || size_k % groups || (size_k / groups) % 8)
    return VLA_ERR_QUANT_SHAPE;
  if (m == 0) return VLA_OK;
  const int gs = size_k / groups;
  hipStream_t s = (hipStream_t)stream;
  if (use_exllama && !b_g_idx && size_k % 64 == 0 && gs % 32 == 0)
    return vla_q4_gemm_packed(a, qweight, qzeros, scales, out, m, size_n, size_k, gs, 1, nullptr,
                              0, stream);
  if (use_exllama)
    return launch_gemm<kGptqShuffled>(a, qweight, qzeros, scales, b_g_idx, out, m, size_n, size_k,
                                      gs, s);
  return launch_gemm<kGptq>(a, qweight, qzeros, scales, b_g_idx, out, m, size_n, size_k, gs, s);
}

extern "C" int vla_gptq_shuffle(int32_t* qweight, const int32_t* q_perm, int32_t* scratch,
                                int size_k, int size_n, int bit, void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (!qweight) return VLA_ERR_ARG;
  if (bit != 4 || size_k % 8) return VLA_ERR_QUANT_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int rows = size_k / 8;
  const int64_t words = (int64_t)rows * size_n;
  if (q_perm) {
    if (!scratch) return VLA_ERR_ARG;
    dim3 grid((size_n + 255) / 256, rows);
    hipLaunchKernelGGL(gptq_make_sequential_kernel, grid, dim3(256), 0, s,
                       (const uint32_t*)qweight, (uint32_t*)scratch, q_perm, size_n);
    VLA_HIP_CHECK(hipGetLastError());
    VLA_HIP_CHECK(hipMemcpyAsync(qweight, scratch, (size_t)words * 4, hipMemcpyDeviceToDevice, s));
  }
  hipLaunchKernelGGL(gptq_shuffle_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s,
                     (uint32_t*)qweight, words);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}
