// Fused forms of the small steps of a decode layer, for gfx950.  They compute exactly what the
// separate entry points compute, in the same rounding order, from fewer launches:
//
//   rotary_cache       rotary_embedding on q and k (csrc/pos_encoding_kernels.cu:10-93) followed by
//                      reshape_and_cache of the rotated k and of v (csrc/cache_kernels.cu:151-202)
//   add_rms_norm       fused_add_rms_norm (csrc/layernorm_kernels.cu:254-289)
//
// Both can take their input either as a model-dtype tensor or as the fp32 split-K slabs of the
// int4 GEMM that produced it: the slabs are summed in slab order and rounded to the model dtype
// first, which is what the GEMM's own reduction kernel does, so results are bit-identical to
// running the reduction as a launch of its own.
#include "vla_common.h"

namespace vla {
namespace {

template <class S>
__device__ __forceinline__ float rnd(float v) { return S::to_f32(S::from_f32(v)); }

struct RotCacheArgs {
  const int64_t* positions;
  const int64_t* slots;
  void* q;
  void* k;
  void* v;
  int64_t q_stride, k_stride, v_stride;
  const void* cos_sin;
  void* key_cache;
  void* value_cache;
  int rot_dim, num_heads, num_kv_heads, head_size, block_size, x;
  float kv_scale;
  const float* slabs;       // null: q/k/v hold the input; else [num_slabs][tokens][slab_row] fp32
  int num_slabs;
  int64_t slab_stride, slab_row;
  int q_col, k_col, v_col;  // first column of q, k, v inside a slab row
};

template <class S, bool FP8, bool NEOX>
__global__ __launch_bounds__(256) void rotary_cache_kernel(RotCacheArgs a) {
  using T = typename S::T;
  using CT = typename std::conditional<FP8, uint8_t, T>::type;
  const int64_t tok = blockIdx.x;
  const int D = a.head_size, embed = a.rot_dim / 2;
  const T* cache = reinterpret_cast<const T*>(a.cos_sin) + a.positions[tok] * a.rot_dim;
  T* q = reinterpret_cast<T*>(a.q) + tok * a.q_stride;
  T* k = reinterpret_cast<T*>(a.k) + tok * a.k_stride;
  T* v = reinterpret_cast<T*>(a.v) + tok * a.v_stride;
  const float* srow = a.slabs ? a.slabs + tok * a.slab_row : nullptr;
  // value of element `col + i` of this token's row, as the model dtype would hold it
  auto in = [&](const T* t, int col, int i) -> float {
    if (srow) {
      float s = srow[col + i];
      for (int p = 1; p < a.num_slabs; ++p) s += srow[(int64_t)p * a.slab_stride + col + i];
      return rnd<S>(s);
    }
    return S::to_f32(t[i]);
  };
  const int64_t slot = a.slots[tok];
  const bool cache_it = slot >= 0;  // padding tokens are rotated but not cached (ref :164-167)
  const int64_t blk = cache_it ? slot / a.block_size : 0, off = cache_it ? slot % a.block_size : 0;
  CT* kc = reinterpret_cast<CT*>(a.key_cache) + blk * a.num_kv_heads * D * a.block_size;
  CT* vc = reinterpret_cast<CT*>(a.value_cache) + blk * a.num_kv_heads * D * a.block_size;
  auto put_k = [&](int h, int d, float val) {
    if (!cache_it) return;
    const int64_t tgt = (int64_t)h * D * a.block_size + (int64_t)(d / a.x) * a.block_size * a.x +
                        off * a.x + d % a.x;
    if constexpr (FP8) kc[tgt] = fp8_e4m3_encode(val / a.kv_scale);
    else kc[tgt] = S::from_f32(val);
  };

  const int nq = a.num_heads * embed, nk = a.num_kv_heads * embed;
  for (int i = threadIdx.x; i < nq + nk; i += blockDim.x) {
#pragma clang fp contract(off)
    const bool is_q = i < nq;
    const int j = is_q ? i : i - nq;
    const int head = j / embed, o = j - head * embed;
    const int xi = head * D + (NEOX ? o : 2 * o), yi = head * D + (NEOX ? embed + o : 2 * o + 1);
    const float c = S::to_f32(cache[o]), s = S::to_f32(cache[embed + o]);
    const float x = in(is_q ? q : k, is_q ? a.q_col : a.k_col, xi);
    const float y = in(is_q ? q : k, is_q ? a.q_col : a.k_col, yi);
    // every product and every sum is rounded to the model dtype, as scalar_t arithmetic does
    const float xr = rnd<S>(rnd<S>(x * c) - rnd<S>(y * s));
    const float yr = rnd<S>(rnd<S>(y * c) + rnd<S>(x * s));
    T* dst = is_q ? q : k;
    dst[xi] = S::from_f32(xr);
    dst[yi] = S::from_f32(yr);
    if (!is_q) {
      put_k(head, xi - head * D, xr);
      put_k(head, yi - head * D, yr);
    }
  }
  // dimensions past rot_dim are not rotated
  const int tail = D - a.rot_dim;
  if (tail > 0) {
    const int tq = a.num_heads * tail, tk = a.num_kv_heads * tail;
    for (int i = threadIdx.x; i < tq + tk; i += blockDim.x) {
      const bool is_q = i < tq;
      const int j = is_q ? i : i - tq;
      const int head = j / tail, d = a.rot_dim + (j - head * tail);
      const float val = in(is_q ? q : k, is_q ? a.q_col : a.k_col, head * D + d);
      if (srow) (is_q ? q : k)[head * D + d] = S::from_f32(val);
      if (!is_q) put_k(head, d, val);
    }
  }
  for (int i = threadIdx.x; i < a.num_kv_heads * D; i += blockDim.x) {
    const float val = in(v, a.v_col, i);
    if (srow) v[i] = S::from_f32(val);
    if (cache_it) {
      const int h = i / D, d = i - h * D;
      const int64_t tgt = (int64_t)h * D * a.block_size + (int64_t)d * a.block_size + off;
      if constexpr (FP8) vc[tgt] = fp8_e4m3_encode(val / a.kv_scale);
      else vc[tgt] = S::from_f32(val);
    }
  }
}

// Vector form for the common layout: NeoX pairing, the whole head rotated, 16-bit model dtype,
// head_size and x multiples of 4.  One task = 4 consecutive dimensions of one head (for q and k:
// 4 rotation pairs, i.e. dimensions o..o+3 and embed+o..embed+o+3); 16-byte slab loads, 8-byte
// tensor loads/stores, 4- or 8-byte K-cache stores.  Arithmetic is the scalar kernel's.
template <class S, bool FP8>
__global__ __launch_bounds__(1024) void rotary_cache_vec_kernel(RotCacheArgs a) {
  using T = typename S::T;
  using CT = typename std::conditional<FP8, uint8_t, T>::type;
  const int64_t tok = blockIdx.x;
  const int D = a.head_size, embed = D / 2;
  const T* cache = reinterpret_cast<const T*>(a.cos_sin) + a.positions[tok] * a.rot_dim;
  T* q = reinterpret_cast<T*>(a.q) + tok * a.q_stride;
  T* k = reinterpret_cast<T*>(a.k) + tok * a.k_stride;
  T* v = reinterpret_cast<T*>(a.v) + tok * a.v_stride;
  const float* srow = a.slabs ? a.slabs + tok * a.slab_row : nullptr;
  auto load4 = [&](const T* t, int col, int i, float (&f)[4]) {
    if (srow) {
      f32x4 s = *reinterpret_cast<const f32x4*>(srow + col + i);
#pragma unroll
      for (int p = 1; p < 8; ++p)
        if (p < a.num_slabs)
          s += *reinterpret_cast<const f32x4*>(srow + (int64_t)p * a.slab_stride + col + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = rnd<S>(s[e]);
    } else {
      const u32x2 w = *reinterpret_cast<const u32x2*>(t + i);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        f[e] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(w[e >> 1] >> (16 * (e & 1)))));
    }
  };
  auto pack4 = [&](const float (&f)[4]) {
    u32x2 w;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      w[e] = (uint32_t)__builtin_bit_cast(uint16_t, S::from_f32(f[2 * e])) |
             ((uint32_t)__builtin_bit_cast(uint16_t, S::from_f32(f[2 * e + 1])) << 16);
    return w;
  };
  const int64_t slot = a.slots[tok];
  const bool cache_it = slot >= 0;
  const int64_t blk = cache_it ? slot / a.block_size : 0, off = cache_it ? slot % a.block_size : 0;
  CT* kc = reinterpret_cast<CT*>(a.key_cache) + blk * a.num_kv_heads * D * a.block_size;
  CT* vc = reinterpret_cast<CT*>(a.value_cache) + blk * a.num_kv_heads * D * a.block_size;
  auto put_k4 = [&](int h, int d, const float (&f)[4]) {   // d % 4 == 0, so inside one x group
    if (!cache_it) return;
    const int64_t tgt = (int64_t)h * D * a.block_size + (int64_t)(d / a.x) * a.block_size * a.x +
                        off * a.x + d % a.x;
    if constexpr (FP8) {
      uint32_t w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) w |= (uint32_t)fp8_e4m3_encode(f[e] / a.kv_scale) << (8 * e);
      *reinterpret_cast<uint32_t*>(kc + tgt) = w;
    } else {
      *reinterpret_cast<u32x2*>(kc + tgt) = pack4(f);
    }
  };
  const int tq = a.num_heads * (embed / 4), tk = a.num_kv_heads * (embed / 4);
  const int tv = a.num_kv_heads * (D / 4);
  for (int i = threadIdx.x; i < tq + tk + tv; i += blockDim.x) {
#pragma clang fp contract(off)
    if (i < tq + tk) {
      const bool is_q = i < tq;
      const int j = is_q ? i : i - tq;
      const int head = j / (embed / 4), o = 4 * (j - head * (embed / 4));
      T* t = is_q ? q : k;
      const int col = is_q ? a.q_col : a.k_col;
      float x[4], y[4], c[4], s[4], xr[4], yr[4];
      load4(t, col, head * D + o, x);
      load4(t, col, head * D + embed + o, y);
      const u32x2 cw = *reinterpret_cast<const u32x2*>(cache + o);
      const u32x2 sw = *reinterpret_cast<const u32x2*>(cache + embed + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c[e] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(cw[e >> 1] >> (16 * (e & 1)))));
        s[e] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(sw[e >> 1] >> (16 * (e & 1)))));
        xr[e] = rnd<S>(rnd<S>(x[e] * c[e]) - rnd<S>(y[e] * s[e]));
        yr[e] = rnd<S>(rnd<S>(y[e] * c[e]) + rnd<S>(x[e] * s[e]));
      }
      *reinterpret_cast<u32x2*>(t + head * D + o) = pack4(xr);
      *reinterpret_cast<u32x2*>(t + head * D + embed + o) = pack4(yr);
      if (!is_q) {
        put_k4(head, o, xr);
        put_k4(head, embed + o, yr);
      }
    } else {
      const int j = 4 * (i - tq - tk);
      float f[4];
      load4(v, a.v_col, j, f);
      if (srow) *reinterpret_cast<u32x2*>(v + j) = pack4(f);
      if (cache_it) {
        const int h = j / D, d = j - h * D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t tgt = (int64_t)h * D * a.block_size + (int64_t)(d + e) * a.block_size + off;
          if constexpr (FP8) vc[tgt] = fp8_e4m3_encode(f[e] / a.kv_scale);
          else vc[tgt] = S::from_f32(f[e]);
        }
      }
    }
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  __syncthreads();
  return r;
}

// x = model_dtype(sum of slabs); residual = x + residual; out = rms_norm(residual) * weight.
// Four elements per lane and pass (16-byte slab loads).
// KEEP: passes of the row a thread holds in registers between the two loops (0: re-read the row
// from `residual`); with it a row of up to KEEP * 4096 elements is read once and the second loop
// does not wait for the first one's stores to come back through L2.
template <class S, int KEEP>
__global__ __launch_bounds__(1024) void add_rms_norm_slabs_kernel(
    const float* __restrict__ slabs, int num_slabs, int64_t slab_stride,
    typename S::T* __restrict__ out, typename S::T* __restrict__ residual,
    const typename S::T* __restrict__ weight, float eps, int hidden) {
  __shared__ float red[16];
  const int64_t base = (int64_t)blockIdx.x * hidden;
  float var = 0.f;
  float xk[KEEP > 0 ? KEEP : 1][4];
  int pass = 0;
  for (int i = threadIdx.x * 4; i < hidden; i += blockDim.x * 4, ++pass) {
    // all slab loads of this chunk are issued before the first one is consumed
    f32x4 part[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (p < num_slabs)
        part[p] = *reinterpret_cast<const f32x4*>(slabs + (int64_t)p * slab_stride + base + i);
    f32x4 s = part[0];
#pragma unroll
    for (int p = 1; p < 8; ++p)
      if (p < num_slabs) s += part[p];
    for (int p = 8; p < num_slabs; ++p)
      s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)p * slab_stride + base + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = rnd<S>(rnd<S>(s[e]) + S::to_f32(residual[base + i + e]));
      residual[base + i + e] = S::from_f32(x);
      var += x * x;
      if constexpr (KEEP > 0) {
#pragma unroll
        for (int q = 0; q < KEEP; ++q)
          if (q == pass) xk[q][e] = x;
      }
    }
  }
  var = block_sum(var, red);
  const float inv = rsqrtf(var / (float)hidden + eps);
  pass = 0;
  for (int i = threadIdx.x * 4; i < hidden; i += blockDim.x * 4, ++pass) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x;
      if constexpr (KEEP > 0) {
        x = xk[0][e];
#pragma unroll
        for (int q = 1; q < KEEP; ++q)
          if (q == pass) x = xk[q][e];
      } else {
        x = S::to_f32(residual[base + i + e]);
      }
      out[base + i + e] = S::from_f32(rnd<S>(x * inv) * S::to_f32(weight[i + e]));
    }
  }
}

// silu_and_mul whose input [tokens][2 d] arrives as split-K slabs: each half is summed over the
// slabs (fp32, slab order) and rounded to the model dtype as the GEMM's own reduction would,
// then out = silu(gate) * up with the activation kernel's arithmetic
// (csrc/activation_kernels.cu:14-33).  Four outputs per thread; grid (tokens, column parts): a
// decode batch of a few tokens would otherwise leave one workgroup per token to walk the whole
// row (14 336 outputs of Llama-3-8B at batch 1: 10 us).
template <class S>
__global__ __launch_bounds__(256) void silu_and_mul_slabs_kernel(
    const float* __restrict__ slabs, int num_slabs, int64_t slab_stride,
    typename S::T* __restrict__ out, int d) {
  const int64_t row = blockIdx.x;
  const float* src = slabs + row * 2 * d;
  for (int i = (blockIdx.y * blockDim.x + threadIdx.x) * 4; i < d; i += gridDim.y * blockDim.x * 4) {
    f32x4 g = *reinterpret_cast<const f32x4*>(src + i);
    f32x4 u = *reinterpret_cast<const f32x4*>(src + d + i);
    for (int p = 1; p < num_slabs; ++p) {
      g += *reinterpret_cast<const f32x4*>(src + (int64_t)p * slab_stride + i);
      u += *reinterpret_cast<const f32x4*>(src + (int64_t)p * slab_stride + d + i);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = rnd<S>(g[e]), y = rnd<S>(u[e]);
      out[row * d + i + e] = S::from_f32(rnd<S>(x / (1.0f + expf(-x))) * y);
    }
  }
}

}  // namespace
}  // namespace vla

using namespace vla;

#define VLA_BY_DTYPE(dtype, CALL) \
  switch (dtype) {                \
    case VLA_F16: { using S = F16; CALL; } break;   \
    case VLA_BF16: { using S = BF16; CALL; } break; \
    case VLA_F32: { using S = F32; CALL; } break;   \
    default: return VLA_ERR_DTYPE;                  \
  }

extern "C" int vla_rotary_reshape_and_cache(
    const int64_t* positions, void* query, void* key, void* value, const void* cos_sin_cache,
    void* key_cache, void* value_cache, const int64_t* slot_mapping, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, int block_size, int x, int is_neox, int dtype, int kv_dtype,
    float kv_scale, const float* slabs, int num_slabs, int64_t slab_stride, int64_t slab_row,
    int q_col, int k_col, int v_col, void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!positions || !query || !key || !value || !cos_sin_cache || !key_cache || !value_cache ||
      !slot_mapping || num_tokens < 0 || rot_dim <= 0 || rot_dim % 2 || rot_dim > head_size ||
      block_size <= 0 || x <= 0 || head_size % x)
    return VLA_ERR_ARG;
  if (slabs && (num_slabs < 1 || slab_row <= 0)) return VLA_ERR_ARG;
  if (kv_dtype != VLA_KV_AUTO && kv_dtype != VLA_KV_FP8_E4M3) return VLA_ERR_KV_DTYPE;
  RotCacheArgs a{positions, slot_mapping, query, key, value, query_stride, key_stride, value_stride,
                 cos_sin_cache, key_cache, value_cache, rot_dim, num_heads, num_kv_heads, head_size,
                 block_size, x, kv_scale, slabs, num_slabs, slab_stride, slab_row, q_col, k_col,
                 v_col};
  hipStream_t s = (hipStream_t)stream;
  const int es = dtype == VLA_F32 ? 4 : 2;
  const bool vec = is_neox && rot_dim == head_size && dtype != VLA_F32 && head_size % 8 == 0 &&
                   x % 4 == 0 && query_stride % 4 == 0 && key_stride % 4 == 0 &&
                   value_stride % 4 == 0 && ((uintptr_t)query % 8 == 0) &&
                   ((uintptr_t)key % 8 == 0) && ((uintptr_t)value % 8 == 0) &&
                   ((uintptr_t)cos_sin_cache % 8 == 0) && ((uintptr_t)key_cache % 8 == 0) &&
                   (!slabs || (num_slabs <= 8 && ((uintptr_t)slabs % 16 == 0) &&
                               slab_stride % 4 == 0 && slab_row % 4 == 0 && q_col % 4 == 0 &&
                               k_col % 4 == 0 && v_col % 4 == 0));
  (void)es;
  if (vec) {
    const int tasks = (num_heads + num_kv_heads) * (head_size / 8) + num_kv_heads * (head_size / 4);
    const int threads = tasks >= 1024 ? 1024 : ((tasks + 63) / 64) * 64;
    if (kv_dtype == VLA_KV_FP8_E4M3) {
      if (dtype == VLA_F16)
        hipLaunchKernelGGL((rotary_cache_vec_kernel<F16, true>), dim3(num_tokens), dim3(threads), 0, s, a);
      else
        hipLaunchKernelGGL((rotary_cache_vec_kernel<BF16, true>), dim3(num_tokens), dim3(threads), 0, s, a);
    } else {
      if (dtype == VLA_F16)
        hipLaunchKernelGGL((rotary_cache_vec_kernel<F16, false>), dim3(num_tokens), dim3(threads), 0, s, a);
      else
        hipLaunchKernelGGL((rotary_cache_vec_kernel<BF16, false>), dim3(num_tokens), dim3(threads), 0, s, a);
    }
    VLA_HIP_CHECK(hipGetLastError());
    return VLA_OK;
  }
#define VLA_RC(FP8)                                                                              \
  do {                                                                                           \
    if (is_neox)                                                                                 \
      hipLaunchKernelGGL((rotary_cache_kernel<S, FP8, true>), dim3(num_tokens), dim3(256), 0, s, a); \
    else                                                                                         \
      hipLaunchKernelGGL((rotary_cache_kernel<S, FP8, false>), dim3(num_tokens), dim3(256), 0, s, a); \
  } while (0)
  if (kv_dtype == VLA_KV_FP8_E4M3) {
    VLA_BY_DTYPE(dtype, VLA_RC(true));
  } else {
    VLA_BY_DTYPE(dtype, VLA_RC(false));
  }
#undef VLA_RC
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_fused_add_rms_norm_splitk(const float* slabs, int num_slabs, int64_t slab_stride,
                                             void* out, void* residual, const void* weight,
                                             float epsilon, int num_tokens, int hidden_size,
                                             int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!slabs || !out || !residual || !weight || num_slabs < 1 || num_tokens < 0 ||
      hidden_size <= 0 || hidden_size % 4 || ((uintptr_t)slabs % 16) || (slab_stride % 4))
    return VLA_ERR_ARG;
  const int chunks = hidden_size / 4;
  const int threads = chunks >= 1024 ? 1024 : ((chunks + 63) / 64) * 64;
#define VLA_ARS(KEEP_)                                                                          \
  hipLaunchKernelGGL((add_rms_norm_slabs_kernel<S, KEEP_>), dim3(num_tokens), dim3(threads), 0,  \
                     (hipStream_t)stream, slabs, num_slabs, slab_stride, (typename S::T*)out,    \
                     (typename S::T*)residual, (const typename S::T*)weight, epsilon, hidden_size)
  if (chunks <= 2 * threads) {
    VLA_BY_DTYPE(dtype, VLA_ARS(2));
  } else {
    VLA_BY_DTYPE(dtype, VLA_ARS(0));
  }
#undef VLA_ARS
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_silu_and_mul_splitk(const float* slabs, int num_slabs, int64_t slab_stride,
                                       void* out, int num_tokens, int d, int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!slabs || !out || num_slabs < 1 || num_tokens < 0 || d <= 0 || d % 4 ||
      ((uintptr_t)slabs % 16) || (slab_stride % 4))
    return VLA_ERR_ARG;
  // column parts until the launch has about two workgroups per CU (one pass of 1024 outputs each
  // at most)
  int gy = (512 + num_tokens - 1) / num_tokens;
  const int passes = (d / 4 + 255) / 256;
  if (gy > passes) gy = passes;
  if (gy < 1) gy = 1;
  VLA_BY_DTYPE(dtype, hipLaunchKernelGGL((silu_and_mul_slabs_kernel<S>), dim3(num_tokens, gy), dim3(256),
                                         0, (hipStream_t)stream, slabs, num_slabs, slab_stride,
                                         (typename S::T*)out, d));
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}
