// KV-cache maintenance for gfx950: reshape_and_cache (scatter new tokens into the paged
// layout, optional fp8 quantisation), copy_blocks, swap_blocks, convert_fp8.
// All of it is byte/index work bounded by HBM (or PCIe for swaps): 16-byte accesses wherever
// the layout allows, no LDS, no matrix cores.
#include "vla_common.h"

#include <type_traits>

namespace vla {
namespace {

// ------------------------------------------------------------------ reshape_and_cache
// Index math of the reference kernel (csrc/cache_kernels.cu:169-189):
//   K[block, h, d / x, off, d % x] = key[t, h, d];   V[block, h, d, off] = value[t, h, d]
// A K chunk (x consecutive d of one token) is 16 contiguous bytes in the cache, so one lane
// moves one chunk with a single 16-byte store; V is a stride-block_size scatter by layout.
template <class S, bool FP8, bool VEC16>
__global__ __launch_bounds__(256) void reshape_and_cache_kernel(
    const typename S::T* __restrict__ key, const typename S::T* __restrict__ value,
    void* __restrict__ key_cache, void* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, int64_t key_stride, int64_t value_stride,
    int num_heads, int head_size, int block_size, int x, float kv_scale) {
  using T = typename S::T;
  using CT = typename std::conditional<FP8, uint8_t, T>::type;
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padding token (ref :164-167)
  const int64_t block_idx = slot / block_size;
  const int64_t block_off = slot % block_size;
  const int n = num_heads * head_size;
  CT* kc = reinterpret_cast<CT*>(key_cache);
  CT* vc = reinterpret_cast<CT*>(value_cache);
  const T* ksrc = key + token * key_stride;
  const T* vsrc = value + token * value_stride;

  if constexpr (VEC16) {
    // one lane per K chunk of x elements
    const int chunks = n / x;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
      const int i = c * x;
      const int h = i / head_size, d = i - h * head_size;
      const int64_t tgt = block_idx * num_heads * head_size * block_size +
                          (int64_t)h * head_size * block_size + (int64_t)(d / x) * block_size * x +
                          block_off * x;
      if constexpr (FP8 && sizeof(T) == 2) {  // x == 16: 32 source bytes -> 16 cache bytes
        u32x4 packed;
        const u32x4* src = reinterpret_cast<const u32x4*>(ksrc + i);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32x4 v = src[half];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            uint32_t o = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              uint32_t word = v[2 * w + (e >> 1)];
              T el = __builtin_bit_cast(T, (uint16_t)((e & 1) ? (word >> 16) : (word & 0xffff)));
              o |= (uint32_t)fp8_e4m3_encode(S::to_f32(el) / kv_scale) << (8 * e);
            }
            packed[2 * half + w] = o;
          }
        }
        *reinterpret_cast<u32x4*>(kc + tgt) = packed;
      } else if constexpr (!FP8) {
        *reinterpret_cast<u32x4*>(kc + tgt) = *reinterpret_cast<const u32x4*>(ksrc + i);
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int h = i / head_size, d = i - h * head_size;
      const int64_t tgt = block_idx * num_heads * head_size * block_size +
                          (int64_t)h * head_size * block_size + (int64_t)(d / x) * block_size * x +
                          block_off * x + d % x;
      if constexpr (FP8) kc[tgt] = fp8_e4m3_encode(S::to_f32(ksrc[i]) / kv_scale);
      else kc[tgt] = ksrc[i];
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int h = i / head_size, d = i - h * head_size;
    const int64_t tgt = block_idx * num_heads * head_size * block_size +
                        (int64_t)h * head_size * block_size + (int64_t)d * block_size + block_off;
    if constexpr (FP8) vc[tgt] = fp8_e4m3_encode(S::to_f32(vsrc[i]) / kv_scale);
    else vc[tgt] = vsrc[i];
  }
}

template <class S, bool FP8>
int launch_reshape(const void* key, const void* value, void* kc, void* vc, const int64_t* slots,
                   int num_tokens, int num_heads, int head_size, int block_size, int x,
                   int64_t ks, int64_t vs, float kv_scale, hipStream_t s) {
  using T = typename S::T;
  const int n = num_heads * head_size;
  const int expect_x = FP8 ? 16 : 16 / (int)sizeof(T);
  // 16-byte path: standard x, 16-bit source for the fp8 packer, aligned rows
  bool vec = (x == expect_x) && (head_size % x == 0) &&
             ((uintptr_t)key % 16 == 0) && ((ks * sizeof(T)) % 16 == 0) &&
             ((uintptr_t)kc % 16 == 0) && (!FP8 || sizeof(T) == 2);
  dim3 grid(num_tokens), block(min(n, 256) < 64 ? 64 : min(n, 256));
  if (vec)
    hipLaunchKernelGGL((reshape_and_cache_kernel<S, FP8, true>), grid, block, 0, s, (const T*)key,
                       (const T*)value, kc, vc, slots, ks, vs, num_heads, head_size, block_size, x,
                       kv_scale);
  else
    hipLaunchKernelGGL((reshape_and_cache_kernel<S, FP8, false>), grid, block, 0, s, (const T*)key,
                       (const T*)value, kc, vc, slots, ks, vs, num_heads, head_size, block_size, x,
                       kv_scale);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

// ------------------------------------------------------------------ copy_blocks
constexpr int kMaxLayersPerLaunch = 64;
struct CopyPtrs {
  void* k[kMaxLayersPerLaunch];
  void* v[kMaxLayersPerLaunch];
};

// grid (pair, layer, {K,V}); the pointer tables ride in the kernel arguments
template <class VecT>
__global__ __launch_bounds__(256) void copy_blocks_kernel(CopyPtrs ptrs,
                                                          const int64_t* __restrict__ mapping,
                                                          int64_t vecs_per_block) {
  const int64_t src = mapping[2 * blockIdx.x], dst = mapping[2 * blockIdx.x + 1];
  VecT* base = reinterpret_cast<VecT*>(blockIdx.z == 0 ? ptrs.k[blockIdx.y] : ptrs.v[blockIdx.y]);
  const VecT* s = base + src * vecs_per_block;
  VecT* d = base + dst * vecs_per_block;
  for (int64_t i = threadIdx.x; i < vecs_per_block; i += blockDim.x) d[i] = s[i];
}

// ------------------------------------------------------------------ swap_blocks (scattered)
// Many short copies: one launch moves up to kSwapPairs blocks, the pair list riding in the kernel
// arguments.  Used when the mapping does not collapse into a few long runs and both sides are
// device-addressable (device memory, or pinned host memory through its device alias).
constexpr int kSwapPairs = 120;
struct SwapPairs {
  int64_t src[kSwapPairs];
  int64_t dst[kSwapPairs];
};

__global__ __launch_bounds__(256) void swap_blocks_kernel(const char* __restrict__ src,
                                                          char* __restrict__ dst, SwapPairs p,
                                                          int64_t block_bytes) {
  const u32x4* s = reinterpret_cast<const u32x4*>(src + p.src[blockIdx.x] * block_bytes);
  u32x4* d = reinterpret_cast<u32x4*>(dst + p.dst[blockIdx.x] * block_bytes);
  const int64_t vecs = block_bytes / 16;
  for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < vecs;
       i += (int64_t)gridDim.y * blockDim.x)
    d[i] = s[i];
}

// ------------------------------------------------------------------ convert_fp8
// eight elements per lane: 16-byte accesses on the 16-bit side, 8-byte on the fp8 side
template <class S, bool TO_FP8>
__global__ __launch_bounds__(256) void convert_fp8_vec8_kernel(void* __restrict__ dst,
                                                               const void* __restrict__ src,
                                                               int64_t n8, float scale) {
  using T = typename S::T;
  static_assert(sizeof(T) == 2, "16-bit model dtypes only");
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    if constexpr (TO_FP8) {
      const u32x4 v = reinterpret_cast<const u32x4*>(src)[i];
      u32x2 o = {0, 0};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const T el = __builtin_bit_cast(T, (uint16_t)(v[e >> 1] >> (16 * (e & 1))));
        o[e >> 2] |= (uint32_t)fp8_e4m3_encode(S::to_f32(el) / scale) << (8 * (e & 3));
      }
      reinterpret_cast<u32x2*>(dst)[i] = o;
    } else {
      const u32x2 v = reinterpret_cast<const u32x2*>(src)[i];
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        uint32_t w = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint8_t b = (uint8_t)(v[(e + h) >> 2] >> (8 * ((e + h) & 3)));
          T r = S::from_f32(fp8_e4m3_decode(b) * scale);
          if (b == 0x80) r = S::from_f32(-0.0f);   // see the scalar kernel below
          w |= (uint32_t)__builtin_bit_cast(uint16_t, r) << (16 * h);
        }
        o[e >> 1] = w;
      }
      reinterpret_cast<u32x4*>(dst)[i] = o;
    }
  }
}

template <class S, bool TO_FP8>
__global__ __launch_bounds__(256) void convert_fp8_kernel(void* __restrict__ dst,
                                                          const void* __restrict__ src, int64_t n,
                                                          float scale) {
  using T = typename S::T;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if constexpr (TO_FP8)
      reinterpret_cast<uint8_t*>(dst)[i] =
          fp8_e4m3_encode(S::to_f32(reinterpret_cast<const T*>(src)[i]) / scale);
    else {
      const uint8_t b = reinterpret_cast<const uint8_t*>(src)[i];
      T r = S::from_f32(fp8_e4m3_decode(b) * scale);
      // keep the sign of the -0 code: the multiply is fused into a mixed-precision fma with a
      // +0 addend, which turns -0 into +0
      if (b == 0x80) r = S::from_f32(-0.0f);
      reinterpret_cast<T*>(dst)[i] = r;
    }
  }
}

template <class S>
int launch_convert(void* dst, const void* src, int64_t n, float scale, int to_fp8, hipStream_t s) {
  if (n == 0) return VLA_OK;
  if constexpr (sizeof(typename S::T) == 2) {
    if (n % 8 == 0 && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)src % 16 == 0)) {
      const int64_t n8 = n / 8;
      int64_t vb = (n8 + 255) / 256;
      if (vb > 16384) vb = 16384;
      if (to_fp8)
        hipLaunchKernelGGL((convert_fp8_vec8_kernel<S, true>), dim3((unsigned)vb), dim3(256), 0, s,
                           dst, src, n8, scale);
      else
        hipLaunchKernelGGL((convert_fp8_vec8_kernel<S, false>), dim3((unsigned)vb), dim3(256), 0, s,
                           dst, src, n8, scale);
      VLA_HIP_CHECK(hipGetLastError());
      return VLA_OK;
    }
  }
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (to_fp8)
    hipLaunchKernelGGL((convert_fp8_kernel<S, true>), dim3((unsigned)blocks), dim3(256), 0, s, dst,
                       src, n, scale);
  else
    hipLaunchKernelGGL((convert_fp8_kernel<S, false>), dim3((unsigned)blocks), dim3(256), 0, s, dst,
                       src, n, scale);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

}  // namespace
}  // namespace vla

using namespace vla;

extern "C" int vla_reshape_and_cache(const void* key, const void* value, void* key_cache,
                                     void* value_cache, const int64_t* slot_mapping,
                                     int num_tokens, int num_heads, int head_size, int block_size,
                                     int x, int64_t key_stride, int64_t value_stride, int dtype,
                                     int kv_dtype, float kv_scale, void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (num_tokens == 0) return VLA_OK;
  if (!key || !value || !key_cache || !value_cache || !slot_mapping) return VLA_ERR_ARG;
  if (num_tokens < 0 || num_heads <= 0 || head_size <= 0 || block_size <= 0 || x <= 0 ||
      head_size % x != 0)
    return VLA_ERR_ARG;
  if (kv_dtype != VLA_KV_AUTO && kv_dtype != VLA_KV_FP8_E4M3) return VLA_ERR_KV_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  const bool fp8 = kv_dtype == VLA_KV_FP8_E4M3;
#define VLA_RC(SC)                                                                              \
  return fp8 ? launch_reshape<SC, true>(key, value, key_cache, value_cache, slot_mapping,        \
                                        num_tokens, num_heads, head_size, block_size, x,         \
                                        key_stride, value_stride, kv_scale, s)                   \
             : launch_reshape<SC, false>(key, value, key_cache, value_cache, slot_mapping,       \
                                         num_tokens, num_heads, head_size, block_size, x,        \
                                         key_stride, value_stride, kv_scale, s)
  switch (dtype) {
    case VLA_F16: VLA_RC(F16);
    case VLA_BF16: VLA_RC(BF16);
    case VLA_F32: VLA_RC(F32);
    default: return VLA_ERR_DTYPE;
  }
#undef VLA_RC
}

extern "C" int vla_copy_blocks(void* const* key_caches, void* const* value_caches, int num_layers,
                               const int64_t* block_mapping, int num_pairs, int64_t block_bytes,
                               void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (num_layers == 0 || num_pairs == 0) return VLA_OK;
  if (!key_caches || !value_caches || !block_mapping || num_layers < 0 || num_pairs < 0 ||
      block_bytes <= 0)
    return VLA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  for (int l0 = 0; l0 < num_layers; l0 += kMaxLayersPerLaunch) {
    const int nl = min(kMaxLayersPerLaunch, num_layers - l0);
    CopyPtrs ptrs;
    bool aligned = block_bytes % 16 == 0;
    for (int i = 0; i < nl; ++i) {
      ptrs.k[i] = key_caches[l0 + i];
      ptrs.v[i] = value_caches[l0 + i];
      if (!ptrs.k[i] || !ptrs.v[i]) return VLA_ERR_ARG;
      aligned = aligned && ((uintptr_t)ptrs.k[i] % 16 == 0) && ((uintptr_t)ptrs.v[i] % 16 == 0);
    }
    dim3 grid(num_pairs, nl, 2);
    if (aligned)
      hipLaunchKernelGGL(copy_blocks_kernel<u32x4>, grid, dim3(256), 0, s, ptrs, block_mapping,
                         block_bytes / 16);
    else
      hipLaunchKernelGGL(copy_blocks_kernel<uint8_t>, grid, dim3(256), 0, s, ptrs, block_mapping,
                         block_bytes);
    VLA_HIP_CHECK(hipGetLastError());
  }
  return VLA_OK;
}

extern "C" int vla_swap_blocks(const void* src, void* dst, const int64_t* block_mapping,
                               int num_pairs, int64_t block_bytes, int direction, void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (num_pairs == 0) return VLA_OK;
  if (!src || !dst || !block_mapping || num_pairs < 0 || block_bytes <= 0) return VLA_ERR_ARG;
  hipMemcpyKind kind;
  switch (direction) {
    case VLA_SWAP_D2D: kind = hipMemcpyDeviceToDevice; break;
    case VLA_SWAP_D2H: kind = hipMemcpyDeviceToHost; break;
    case VLA_SWAP_H2D: kind = hipMemcpyHostToDevice; break;
    default: return VLA_ERR_ARG;  // "Invalid device combination" (ref :38-40)
  }
  hipStream_t s = (hipStream_t)stream;
  const char* sp = static_cast<const char*>(src);
  char* dp = static_cast<char*>(dst);
  int runs = 0;
  for (int i = 0; i < num_pairs; ++i) {
    if (block_mapping[2 * i] < 0 || block_mapping[2 * i + 1] < 0) return VLA_ERR_ARG;
    if (i == 0 || block_mapping[2 * i] != block_mapping[2 * i - 2] + 1 ||
        block_mapping[2 * i + 1] != block_mapping[2 * i - 1] + 1)
      ++runs;
  }
  // A scattered mapping is moved by one kernel per 120 blocks instead of one copy call per block,
  // if the host side (when there is one) is pinned memory the device can address.
  if (runs > 4 && block_bytes % 16 == 0) {
    const char* ksrc = sp;
    char* kdst = dp;
    bool ok = true;
    auto device_alias = [&](const void* host) -> void* {
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, host) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
      return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
    };
    if (direction == VLA_SWAP_D2H) {
      kdst = static_cast<char*>(device_alias(dst));
      ok = kdst != nullptr;
    } else if (direction == VLA_SWAP_H2D) {
      ksrc = static_cast<const char*>(device_alias(src));
      ok = ksrc != nullptr;
    }
    if (ok && ((uintptr_t)ksrc % 16 == 0) && ((uintptr_t)kdst % 16 == 0)) {
      int64_t ysplit = (block_bytes / 16 + 256 * 4 - 1) / (256 * 4);
      if (ysplit > 64) ysplit = 64;
      for (int i0 = 0; i0 < num_pairs; i0 += kSwapPairs) {
        const int np = min(kSwapPairs, num_pairs - i0);
        SwapPairs p;
        for (int j = 0; j < np; ++j) {
          p.src[j] = block_mapping[2 * (i0 + j)];
          p.dst[j] = block_mapping[2 * (i0 + j) + 1];
        }
        hipLaunchKernelGGL(swap_blocks_kernel, dim3(np, (unsigned)ysplit), dim3(256), 0, s, ksrc,
                           kdst, p, block_bytes);
      }
      VLA_HIP_CHECK(hipGetLastError());
      return VLA_OK;
    }
  }
  int i = 0;
  while (i < num_pairs) {
    const int64_t s0 = block_mapping[2 * i], d0 = block_mapping[2 * i + 1];
    int run = 1;  // consecutive (src+1, dst+1) pairs collapse into one copy
    while (i + run < num_pairs && block_mapping[2 * (i + run)] == s0 + run &&
           block_mapping[2 * (i + run) + 1] == d0 + run)
      ++run;
    VLA_HIP_CHECK(hipMemcpyAsync(dp + d0 * block_bytes, sp + s0 * block_bytes,
                                 (size_t)run * block_bytes, kind, s));
    i += run;
  }
  return VLA_OK;
}

extern "C" int vla_convert_fp8(void* dst, const void* src, int64_t n, int dtype, int kv_dtype,
                               float scale, int to_fp8, void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (n == 0) return VLA_OK;
  if (!dst || !src || n < 0) return VLA_ERR_ARG;
  // "auto" has no fp8 codec behind it (the reference asserts, amd/quant_utils.cuh:521-539)
  if (kv_dtype != VLA_KV_FP8_E4M3) return VLA_ERR_KV_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case VLA_F16: return launch_convert<F16>(dst, src, n, scale, to_fp8, s);
    case VLA_BF16: return launch_convert<BF16>(dst, src, n, scale, to_fp8, s);
    case VLA_F32: return launch_convert<F32>(dst, src, n, scale, to_fp8, s);
    default: return VLA_ERR_DTYPE;
  }
}
