// The elementwise steps on either side of attention in a decoder layer -- RMSNorm (plain and
// fused with the residual add), rotary position embedding, SwiGLU gating -- for gfx950.
// All are HBM/L2-bound byte movers: one workgroup per token, 16-byte accesses, wave64 shuffle
// reductions, no matrix cores.  Rounding points follow the reference kernels so that results can
// be compared bit for bit where no transcendental is involved:
//   rms_norm            out = model_dtype(x * rsqrt(mean(x^2) + eps)) * weight
//                       (csrc/layernorm_kernels.cu:22-46)
//   fused_add_rms_norm  residual = input + residual (model dtype); input = norm(residual) * weight
//                       (csrc/layernorm_kernels.cu:254-289)
//   rotary_embedding    x' = x*cos - y*sin, y' = y*cos + x*sin, every product and sum rounded to
//                       the model dtype (csrc/pos_encoding_kernels.cu:10-35), NeoX or GPT-J pairing
//   silu_and_mul        out = model_dtype(x / (1 + exp(-x))) * y   (csrc/activation_kernels.cu:14-33)
#include "vla_common.h"

namespace vla {
namespace {

template <class S>
__device__ __forceinline__ float rnd(float v) { return S::to_f32(S::from_f32(v)); }

__device__ __forceinline__ float block_sum_f32(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  __syncthreads();
  return r;
}

template <class S>
struct Vec16 {  // 16 bytes of model-dtype elements
  using T = typename S::T;
  static constexpr int N = 16 / sizeof(T);
  static __device__ __forceinline__ void load(const T* p, float (&f)[16 / sizeof(T)]) {
    if constexpr (sizeof(T) == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = v[i];
    } else {
      u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f[2 * i] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(v[i] & 0xffff)));
        f[2 * i + 1] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(v[i] >> 16)));
      }
    }
  }
  // f[] already hold model-dtype-representable values
  static __device__ __forceinline__ void store(T* p, const float (&f)[16 / sizeof(T)]) {
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
    } else {
      u32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = (uint32_t)__builtin_bit_cast(uint16_t, S::from_f32(f[2 * i])) |
               ((uint32_t)__builtin_bit_cast(uint16_t, S::from_f32(f[2 * i + 1])) << 16);
      *reinterpret_cast<u32x4*>(p) = v;
    }
  }
};

// FUSED: input += residual first (result also written to residual), output goes to `input`.
// KEEP: passes of the row a thread holds in registers between the two loops (0: re-read the row).
// A row of up to KEEP * 256 * N elements is then read once -- the second loop of the generic form
// loads what the first has just stored, a round trip through L2 that a decode batch of a few
// tokens (one workgroup each, nothing else in flight) waits out in full.
template <class S, bool FUSED, int KEEP>
__global__ __launch_bounds__(1024) void rms_norm_kernel(typename S::T* __restrict__ out,
                                                       typename S::T* __restrict__ input,
                                                       typename S::T* __restrict__ residual,
                                                       const typename S::T* __restrict__ weight,
                                                       float eps, int hidden) {
  using V = Vec16<S>;
  constexpr int N = V::N;
  __shared__ float red[16];
  const int64_t base = (int64_t)blockIdx.x * hidden;
  float var = 0.f;
  if constexpr (KEEP > 0) {
    float xk[KEEP][N], wk[KEEP][N];
#pragma unroll
    for (int p = 0; p < KEEP; ++p) {
      const int i = (p * blockDim.x + threadIdx.x) * N;
      if (i < hidden) {
        V::load(input + base + i, xk[p]);
        V::load(weight + i, wk[p]);
        if constexpr (FUSED) {
          float r[N];
          V::load(residual + base + i, r);
#pragma unroll
          for (int e = 0; e < N; ++e) xk[p][e] = rnd<S>(xk[p][e] + r[e]);
          V::store(residual + base + i, xk[p]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) var += xk[p][e] * xk[p][e];
      }
    }
    var = block_sum_f32(var, red);
    const float inv = rsqrtf(var / (float)hidden + eps);
    typename S::T* dst = FUSED ? input : out;
#pragma unroll
    for (int p = 0; p < KEEP; ++p) {
      const int i = (p * blockDim.x + threadIdx.x) * N;
      if (i < hidden) {
#pragma unroll
        for (int e = 0; e < N; ++e) xk[p][e] = rnd<S>(rnd<S>(xk[p][e] * inv) * wk[p][e]);
        V::store(dst + base + i, xk[p]);
      }
    }
    return;
  }
  for (int i = threadIdx.x * N; i < hidden; i += blockDim.x * N) {
    float x[N];
    V::load(input + base + i, x);
    if constexpr (FUSED) {
      float r[N];
      V::load(residual + base + i, r);
#pragma unroll
      for (int e = 0; e < N; ++e) x[e] = rnd<S>(x[e] + r[e]);
      V::store(residual + base + i, x);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) var += x[e] * x[e];
  }
  var = block_sum_f32(var, red);
  const float inv = rsqrtf(var / (float)hidden + eps);
  const typename S::T* src = FUSED ? residual : input;
  typename S::T* dst = FUSED ? input : out;
  for (int i = threadIdx.x * N; i < hidden; i += blockDim.x * N) {
    float x[N], w[N];
    V::load(src + base + i, x);
    V::load(weight + i, w);
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = rnd<S>(rnd<S>(x[e] * inv) * w[e]);
    V::store(dst + base + i, x);
  }
}

// one workgroup per token; a lane rotates one (x, y) pair of one head per iteration
template <class S, bool NEOX>
__global__ __launch_bounds__(256) void rotary_kernel(const int64_t* __restrict__ positions,
                                                     typename S::T* __restrict__ query,
                                                     typename S::T* __restrict__ key,
                                                     const typename S::T* __restrict__ cos_sin_cache,
                                                     int rot_dim, int64_t q_stride, int64_t k_stride,
                                                     int num_heads, int num_kv_heads, int head_size) {
  using T = typename S::T;
  const int64_t tok = blockIdx.x;
  const T* cache = cos_sin_cache + positions[tok] * rot_dim;
  const int embed = rot_dim / 2;
  const int nq = num_heads * embed, nk = num_kv_heads * embed;
  for (int i = threadIdx.x; i < nq + nk; i += blockDim.x) {
#pragma clang fp contract(off)
    const bool is_q = i < nq;
    const int j = is_q ? i : i - nq;
    const int head = j / embed, off = j - head * embed;
    T* arr = (is_q ? query + tok * q_stride : key + tok * k_stride) + (int64_t)head * head_size;
    const int xi = NEOX ? off : 2 * off, yi = NEOX ? embed + off : 2 * off + 1;
    const float c = S::to_f32(cache[off]), s = S::to_f32(cache[embed + off]);
    const float x = S::to_f32(arr[xi]), y = S::to_f32(arr[yi]);
    // each product and each sum is rounded to the model dtype, as scalar_t arithmetic does
    // (for float32 that means no multiply-add contraction)
    arr[xi] = S::from_f32(rnd<S>(x * c) - rnd<S>(y * s));
    arr[yi] = S::from_f32(rnd<S>(y * c) + rnd<S>(x * s));
  }
}

template <class S>
__global__ __launch_bounds__(256) void silu_and_mul_kernel(typename S::T* __restrict__ out,
                                                           const typename S::T* __restrict__ in,
                                                           int d) {
  using V = Vec16<S>;
  constexpr int N = V::N;
  const int64_t tok = blockIdx.x;
  for (int i = (blockIdx.y * blockDim.x + threadIdx.x) * N; i < d; i += gridDim.y * blockDim.x * N) {
    float x[N], y[N];
    V::load(in + tok * 2 * d + i, x);
    V::load(in + tok * 2 * d + d + i, y);
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = rnd<S>(rnd<S>(x[e] / (1.0f + expf(-x[e]))) * y[e]);
    V::store(out + tok * d + i, x);
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

static bool vec_ok(const void* p, int64_t elems, int dtype) {
  const int es = dtype == VLA_F32 ? 4 : 2;
  return ((uintptr_t)p % 16 == 0) && ((elems * es) % 16 == 0);
}

extern "C" int vla_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                            int num_tokens, int hidden_size, int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!out || !input || !weight || num_tokens < 0 || hidden_size <= 0) return VLA_ERR_ARG;
  if (!vec_ok(out, hidden_size, dtype) || !vec_ok(input, hidden_size, dtype) ||
      !vec_ok(weight, hidden_size, dtype))
    return VLA_ERR_ARG;
  // one 16-byte vector per thread and pass, up to 1024 threads: a row of 8192 16-bit elements
  // (Llama-3-70B) is one pass of one workgroup instead of four passes of 256 threads
  const int vecs = hidden_size / (dtype == VLA_F32 ? 4 : 8);
  const int threads = vecs >= 1024 ? 1024 : ((vecs + 63) / 64) * 64;
  const bool keep = vecs <= 2 * threads;
#define VLA_RMS(KEEP_)                                                                          \
  hipLaunchKernelGGL((rms_norm_kernel<S, false, KEEP_>), dim3(num_tokens), dim3(threads), 0,     \
                     (hipStream_t)stream, (typename S::T*)out, (typename S::T*)input,            \
                     (typename S::T*)nullptr, (const typename S::T*)weight, epsilon, hidden_size)
  if (keep) {
    VLA_BY_DTYPE(dtype, VLA_RMS(2));
  } else {
    VLA_BY_DTYPE(dtype, VLA_RMS(0));
  }
#undef VLA_RMS
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_fused_add_rms_norm(void* input, void* residual, const void* weight,
                                      float epsilon, int num_tokens, int hidden_size, int dtype,
                                      void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!input || !residual || !weight || num_tokens < 0 || hidden_size <= 0) return VLA_ERR_ARG;
  if (!vec_ok(input, hidden_size, dtype) || !vec_ok(residual, hidden_size, dtype) ||
      !vec_ok(weight, hidden_size, dtype))
    return VLA_ERR_ARG;
  const int vecs = hidden_size / (dtype == VLA_F32 ? 4 : 8);
  const int threads = vecs >= 1024 ? 1024 : ((vecs + 63) / 64) * 64;
  const bool keep = vecs <= 2 * threads;
#define VLA_RMS(KEEP_)                                                                          \
  hipLaunchKernelGGL((rms_norm_kernel<S, true, KEEP_>), dim3(num_tokens), dim3(threads), 0,      \
                     (hipStream_t)stream, (typename S::T*)nullptr, (typename S::T*)input,        \
                     (typename S::T*)residual, (const typename S::T*)weight, epsilon, hidden_size)
  if (keep) {
    VLA_BY_DTYPE(dtype, VLA_RMS(2));
  } else {
    VLA_BY_DTYPE(dtype, VLA_RMS(0));
  }
#undef VLA_RMS
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_rotary_embedding(const int64_t* positions, void* query, void* key,
                                    const void* cos_sin_cache, int num_tokens, int num_heads,
                                    int num_kv_heads, int head_size, int rot_dim,
                                    int64_t query_stride, int64_t key_stride, int is_neox,
                                    int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!positions || !query || !key || !cos_sin_cache || num_tokens < 0 || rot_dim <= 0 ||
      rot_dim % 2 || rot_dim > head_size)
    return VLA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (is_neox) {
    VLA_BY_DTYPE(dtype, hipLaunchKernelGGL((rotary_kernel<S, true>), dim3(num_tokens), dim3(256), 0,
                                           s, positions, (typename S::T*)query,
                                           (typename S::T*)key, (const typename S::T*)cos_sin_cache,
                                           rot_dim, query_stride, key_stride, num_heads,
                                           num_kv_heads, head_size));
  } else {
    VLA_BY_DTYPE(dtype, hipLaunchKernelGGL((rotary_kernel<S, false>), dim3(num_tokens), dim3(256),
                                           0, s, positions, (typename S::T*)query,
                                           (typename S::T*)key, (const typename S::T*)cos_sin_cache,
                                           rot_dim, query_stride, key_stride, num_heads,
                                           num_kv_heads, head_size));
  }
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dtype,
                                void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!out || !input || num_tokens < 0 || d <= 0) return VLA_ERR_ARG;
  if (!vec_ok(out, d, dtype) || !vec_ok(input, d, dtype)) return VLA_ERR_ARG;
  const int n = dtype == VLA_F32 ? 4 : 8;
  // column parts: one pass per workgroup while the launch stays under about two workgroups per
  // CU, at most eight parts otherwise
  int gy = (d / n + 255) / 256;
  const int cap = num_tokens >= 64 ? 8 : (512 + num_tokens - 1) / num_tokens;
  if (gy > cap) gy = cap;
  VLA_BY_DTYPE(dtype, hipLaunchKernelGGL((silu_and_mul_kernel<S>), dim3(num_tokens, gy), dim3(256),
                                         0, (hipStream_t)stream, (typename S::T*)out,
                                         (const typename S::T*)input, d));
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}
