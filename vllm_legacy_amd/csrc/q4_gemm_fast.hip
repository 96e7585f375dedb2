// int4 dequant-GEMM for gfx950, throughput version: out[M, N] = x[M, K] (fp16) . W[K, N].
//
// Weight format ("k-packed, pair order"): qw[K/8][N] uint32, one word = 8 consecutive k of one
// column, nibbles ordered k0,k2,k4,k6,k1,k3,k5,k7 -- exactly what the reference's gptq_shuffle
// writes (csrc/quantization/gptq/qdq_4.cuh:12-29).  In that order four mask-and-or operations
// turn one word into the 8 fp16 values (k0..k7) of a column, i.e. into one 16-byte MFMA B operand
// held by the lane that loaded the word: the weights never pass through LDS.  AWQ checkpoints are
// brought into the same format once by vla_awq_repack (AWQ packs along N, dequantize.cuh:41-61).
// zeros: qz[K/G][N/8] uint32, natural nibble order; effective zero = nibble + ZOFF (GPTQ stores
// zero - 1, q_gemm.cu:1413; AWQ stores the zero itself).  scales: [K/G][N] fp16.
//
// Each weight is reconstructed bit-exactly as the reference does it -- (q - z) exactly in fp16,
// then one fp16 multiply by the scale (gemm_kernels.cu:156-179, q_gemm.cu:939-947) -- and the
// contraction runs on 16x16x32 fp16 MFMAs with fp32 accumulation, rounded to fp16 once.
//
// Workgroup: 512 threads = 8 waves as 4 (M) x 2 (N), tile 256 x 128, K step 64.  x is staged
// through LDS in full 128-byte rows (XOR-swizzled 16-byte chunks, conflict-free ds_read_b128),
// double buffered, one barrier per K step; weight words, zeros and scales go global -> registers.
// Small-N layers are split along K into fp32 slabs that a second kernel sums.
#include "vla_common.h"

namespace vla {
namespace {

constexpr int TM = 256, TN = 128, TK = 64;
constexpr int kThreadsG = 512;
constexpr int X_TILE_BYTES = TM * TK * 2;  // 32 KiB per buffer

__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t as_u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }

template <int ZOFF, bool SPLIT>
__global__ __launch_bounds__(kThreadsG, 2) void q4_gemm_fast_kernel(
    const _Float16* __restrict__ x, const uint32_t* __restrict__ qw,
    const uint32_t* __restrict__ qz, const _Float16* __restrict__ scales, void* __restrict__ outp,
    int m, int n, int k, int group_size, int k_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * X_TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM;
  const int k_begin = blockIdx.z * k_per_split;
  const int k_end = min(k, k_begin + k_per_split);

  // ---- x staging: thread -> (row = 64 i + tid / 8, 16-byte chunk tid % 8)
  const int srow = tid >> 3, schunk = tid & 7;
  const _Float16* xsrc = x + (int64_t)(m0 + srow) * k + 8 * schunk;
  const int sdst = srow * 128 + ((schunk ^ (srow & 7)) << 4);
  u32x4 xr[4];
  auto load_x = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xr[i] = (m0 + srow + 64 * i < m)
                  ? *reinterpret_cast<const u32x4*>(xsrc + (int64_t)(64 * i) * k + k0)
                  : u32x4{0, 0, 0, 0};
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4*>(xs + buf * X_TILE_BYTES + sdst + i * (64 * 128)) = xr[i];
  };

  // ---- weight words: lane (g, c) owns column ncol[b] and rows k8 = .. + g of each 32-deep step
  int ncol[4];
  bool nok[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    ncol[b] = n0 + 64 * wn + 16 * b + c;
    nok[b] = ncol[b] < n;
  }
  uint32_t wr[2][4];
  auto load_w = [&](int k0, uint32_t (&w)[2][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int64_t row = (int64_t)((k0 + 32 * kk) / 8 + g) * n;
#pragma unroll
      for (int b = 0; b < 4; ++b) w[kk][b] = nok[b] ? qw[row + ncol[b]] : 0u;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-group dequantisation constants of this lane's 4 columns
  f16x2 zc1[4], zc2[4], sc2[4];
  const f16x2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};

  load_x(k_begin);
  load_w(k_begin, wr);
  store_x(0);
  __syncthreads();

  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += TK) {
    const bool has_next = k0 + TK < k_end;
    uint32_t wnx[2][4];
    if (has_next) {
      load_x(k0 + TK);
      load_w(k0 + TK, wnx);
    }
    const unsigned char* xb = xs + buf * X_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ks = k0 + 32 * kk;
      if (ks == k_begin || ks % group_size == 0) {  // wave-uniform: new quantisation group
        const int grp = ks / group_size;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          uint32_t zw = 0;
          _Float16 s = (_Float16)0.f;
          if (nok[b]) {
            zw = qz[(int64_t)grp * (n >> 3) + (ncol[b] >> 3)];
            s = scales[(int64_t)grp * n + ncol[b]];
          }
          const int z = (int)((zw >> (4 * (ncol[b] & 7))) & 15u) + ZOFF;
          const _Float16 z1 = (_Float16)(float)(-(1024 + z)), z2 = (_Float16)(float)(-(64 + z));
          zc1[b] = f16x2{z1, z1};
          zc2[b] = f16x2{z2, z2};
          sc2[b] = f16x2{s, s};
        }
      }
      // B operands: dequantise this lane's words
      u32x4 bf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t w = wr[kk][b], w8 = w >> 8;
        const f16x2 d0 = as_h2((w & 0x000f000fu) | 0x64006400u) + zc1[b];
        const f16x2 d1 = __builtin_elementwise_fma(as_h2((w & 0x00f000f0u) | 0x64006400u),
                                                   sixteenth, zc2[b]);
        const f16x2 d2 = as_h2((w8 & 0x000f000fu) | 0x64006400u) + zc1[b];
        const f16x2 d3 = __builtin_elementwise_fma(as_h2((w8 & 0x00f000f0u) | 0x64006400u),
                                                   sixteenth, zc2[b]);
        bf[b][0] = as_u32(d0 * sc2[b]);
        bf[b][1] = as_u32(d1 * sc2[b]);
        bf[b][2] = as_u32(d2 * sc2[b]);
        bf[b][3] = as_u32(d3 * sc2[b]);
      }
      // A operands from LDS, 16 MFMAs
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int row = 64 * wm + 16 * a + c;
        const u32x4 af = *reinterpret_cast<const u32x4*>(
            xb + row * 128 + (((4 * kk + g) ^ (row & 7)) << 4));
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af),
                                                             __builtin_bit_cast(f16x8, bf[b]),
                                                             acc[a][b], 0, 0, 0);
      }
    }
    if (has_next) {
      store_x(buf ^ 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int b = 0; b < 4; ++b) wr[kk][b] = wnx[kk][b];
    }
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: lane (g, c) holds column ncol[b], rows 4g .. 4g+3 of each 16-row tile
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 64 * wm + 16 * a + 4 * g + i;
      if (row >= m) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (!nok[b]) continue;
        if constexpr (SPLIT)
          reinterpret_cast<float*>(outp)[((int64_t)blockIdx.z * m + row) * n + ncol[b]] = acc[a][b][i];
        else
          reinterpret_cast<_Float16*>(outp)[(int64_t)row * n + ncol[b]] = (_Float16)acc[a][b][i];
      }
    }
}

// out[i] = fp16(sum over splits of slab[s][i]); 4 elements per thread
__global__ __launch_bounds__(256) void q4_splitk_reduce_kernel(const float* __restrict__ slabs,
                                                               _Float16* __restrict__ out,
                                                               int64_t elems, int splits) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= elems) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
  for (int p = 1; p < splits; ++p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (int64_t)p * elems + i);
    s += v;
  }
  u32x2 o;
  o[0] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[0]) |
         ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[1]) << 16);
  o[1] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[2]) |
         ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)s[3]) << 16);
  *reinterpret_cast<u32x2*>(out + i) = o;
}

// AWQ [K][N/8] (8 columns per word, nibble P[i] = column 8j+i, P = {0,4,1,5,2,6,3,7}) ->
// k-packed pair order [K/8][N]; one thread per output word
__global__ __launch_bounds__(256) void awq_repack_weight_kernel(const uint32_t* __restrict__ qweight,
                                                                uint32_t* __restrict__ out, int k,
                                                                int n) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int k8 = blockIdx.y;
  if (col >= n) return;
  const int i = col & 7, pos = 4 * ((i >> 1) + 4 * (i & 1));
  uint32_t w = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t q = (qweight[(int64_t)(k8 * 8 + e) * (n >> 3) + (col >> 3)] >> pos) & 15u;
    const int dst = (e & 1) ? 4 + (e >> 1) : (e >> 1);
    w |= q << (4 * dst);
  }
  out[(int64_t)k8 * n + col] = w;
}

// AWQ zeros [K/G][N/8] in AWQ nibble order -> natural nibble order (nibble i = column 8j+i)
__global__ __launch_bounds__(256) void awq_repack_zeros_kernel(const uint32_t* __restrict__ qzeros,
                                                               uint32_t* __restrict__ out,
                                                               int64_t words) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= words) return;
  const uint32_t z = qzeros[idx];
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= ((z >> (4 * ((i >> 1) + 4 * (i & 1)))) & 15u) << (4 * i);
  out[idx] = o;
}

}  // namespace
}  // namespace vla

using namespace vla;

static int pick_splits(int m, int n, int k, int group) {
  const int wgs = ((n + TN - 1) / TN) * ((m + TM - 1) / TM);
  if (wgs >= 160) return 1;
  int want = (256 + wgs - 1) / wgs;
  if (want > 8) want = 8;
  const int unit = group > TK ? group : TK;  // a split must hold whole groups and whole K steps
  while (want > 1 && (k % (want * unit)) != 0) --want;
  return want;
}

extern "C" int64_t vla_q4_gemm_workspace_bytes(int m, int n, int k, int group) {
  if (m <= 0 || n <= 0 || k <= 0 || group <= 0) return 0;
  const int s = pick_splits(m, n, k, group);
  return s > 1 ? (int64_t)s * m * n * 4 : 0;
}

extern "C" int vla_q4_gemm_packed(const void* x, const int32_t* qweight, const int32_t* qzeros,
                                  const void* scales, void* out, int m, int n, int k, int group,
                                  int zero_offset, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  (void)hipGetLastError();
  if (!x || !qweight || !qzeros || !scales || !out || m < 0) return VLA_ERR_ARG;
  if (n % 8 || k % TK || group <= 0 || group % 32 || k % group ||
      (zero_offset != 0 && zero_offset != 1))
    return VLA_ERR_QUANT_SHAPE;
  if (((uintptr_t)x % 16) || ((uintptr_t)out % 8)) return VLA_ERR_ARG;
  if (m == 0) return VLA_OK;
  hipStream_t s = (hipStream_t)stream;
  int splits = pick_splits(m, n, k, group);
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * m * n * 4 ||
                     ((uintptr_t)workspace % 16) || ((int64_t)m * n) % 4))
    splits = 1;
  dim3 grid((n + TN - 1) / TN, (m + TM - 1) / TM, splits);
  const int kps = k / splits;
  void* dst = splits > 1 ? workspace : out;
#define VLA_Q4_LAUNCH(Z, SP)                                                                     \
  hipLaunchKernelGGL((q4_gemm_fast_kernel<Z, SP>), grid, dim3(kThreadsG), 0, s,                  \
                     (const _Float16*)x, (const uint32_t*)qweight, (const uint32_t*)qzeros,      \
                     (const _Float16*)scales, dst, m, n, k, group, kps)
  if (zero_offset == 0) {
    if (splits > 1) VLA_Q4_LAUNCH(0, true); else VLA_Q4_LAUNCH(0, false);
  } else {
    if (splits > 1) VLA_Q4_LAUNCH(1, true); else VLA_Q4_LAUNCH(1, false);
  }
#undef VLA_Q4_LAUNCH
  VLA_HIP_CHECK(hipGetLastError());
  if (splits > 1) {
    const int64_t elems = (int64_t)m * n;
    hipLaunchKernelGGL(q4_splitk_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)),
                       dim3(256), 0, s, (const float*)workspace, (_Float16*)out, elems, splits);
    VLA_HIP_CHECK(hipGetLastError());
  }
  return VLA_OK;
}

extern "C" int vla_awq_repack(const int32_t* qweight, const int32_t* qzeros, int32_t* out_qweight,
                              int32_t* out_qzeros, int in_c, int out_c, int group, void* stream) {
  (void)hipGetLastError();
  if (!qweight || !qzeros || !out_qweight || !out_qzeros) return VLA_ERR_ARG;
  if (in_c <= 0 || out_c <= 0 || in_c % 8 || out_c % 8 || group <= 0 || in_c % group)
    return VLA_ERR_QUANT_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(awq_repack_weight_kernel, dim3((out_c + 255) / 256, in_c / 8), dim3(256), 0, s,
                     (const uint32_t*)qweight, (uint32_t*)out_qweight, in_c, out_c);
  const int64_t zw = (int64_t)(in_c / group) * (out_c / 8);
  hipLaunchKernelGGL(awq_repack_zeros_kernel, dim3((unsigned)((zw + 255) / 256)), dim3(256), 0, s,
                     (const uint32_t*)qzeros, (uint32_t*)out_qzeros, zw);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}
