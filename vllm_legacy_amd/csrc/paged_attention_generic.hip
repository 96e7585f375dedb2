// Generic single-query paged attention for gfx950: any supported head size / block size /
// model dtype (fp16, bf16, fp32), "auto" or fp8-e4m3 KV.  One 256-thread workgroup per
// (query head, sequence[, 512-token partition]); the sequence is walked in 512-token chunks
// with an online softmax so LDS use does not grow with the context length.  This is the
// coverage kernel (fp32 models, head sizes 80/112, block size 8 ...); the MFMA kernel in
// paged_attention_mfma.hip takes the 16-bit shapes that matter for throughput.
//
// Math follows the reference kernel body (csrc/attention/attention_kernels.cu:90-496):
// fp32 logits = scale * q.k (+ alibi), probabilities rounded to the model dtype before the
// value product, fp32 accumulation, normaliser 1 / (sum + 1e-6).  Block-sparse attention
// (:209-254, :385-393) is a per-KV-block mask: a skipped block's tokens get logit -FLT_MAX and
// its values are never loaded.
#include "vla_common.h"

#include <float.h>

namespace vla {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;
constexpr int kChunk = VLA_PARTITION_SIZE;  // tokens per online-softmax chunk (= V2 partition)

template <class S, bool FP8>
struct Cache {
  using T = typename S::T;
  static constexpr int kVec = 16 / sizeof(T);  // elements per 16-byte load
  static __device__ __forceinline__ void load(const void* base, int64_t idx, float kv_scale,
                                              float (&out)[16 / sizeof(T)]) {
    const T* p = reinterpret_cast<const T*>(base) + idx;
    if constexpr (sizeof(T) == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
      for (int i = 0; i < 4; ++i) out[i] = v[i];
    } else {
      u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t w = v[i];
        out[2 * i] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(w & 0xffff)));
        out[2 * i + 1] = S::to_f32(__builtin_bit_cast(T, (uint16_t)(w >> 16)));
      }
    }
  }
};

// fp8 cache: 8 elements (8 bytes) per load; value = round_to_model_dtype(fp8 * kv_scale)
// (scaled_convert, csrc/quantization/fp8/amd/quant_utils.cuh:310-318, :377-385, :427-433)
template <class S>
struct Cache<S, true> {
  static constexpr int kVec = 8;
  static __device__ __forceinline__ void load(const void* base, int64_t idx, float kv_scale,
                                              float (&out)[8]) {
    u32x2 v = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(base) + idx);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x2 a = fp8x2_to_f32<false>(v[i]);
      f32x2 b = fp8x2_to_f32<true>(v[i]);
      out[4 * i + 0] = S::to_f32(S::from_f32(a[0] * kv_scale));
      out[4 * i + 1] = S::to_f32(S::from_f32(a[1] * kv_scale));
      out[4 * i + 2] = S::to_f32(S::from_f32(b[0] * kv_scale));
      out[4 * i + 3] = S::to_f32(S::from_f32(b[1] * kv_scale));
    }
  }
};

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < kWaves; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < kWaves; ++i) r += red[i];
  return r;
}

template <class S, bool FP8, bool PARTITIONED, bool SPARSE>
__global__ __launch_bounds__(kThreads) void pa_generic_kernel(AttnParams p) {
  using T = typename S::T;
  using C = Cache<S, FP8>;
  constexpr int VEC = C::kVec;
  constexpr int X = FP8 ? 16 : 16 / (int)sizeof(T);  // elements per 16-byte K chunk

  const int head = blockIdx.x, seq = p.seq_order ? p.seq_order[blockIdx.y] : blockIdx.y;
  const int part = blockIdx.z;
  const int tid = threadIdx.x;
  const int seq_len = p.seq_lens[seq];
  const int tok_begin = PARTITIONED ? part * kChunk : 0;
  if (tok_begin >= seq_len) {
    // partition past the end: nothing written (ref :116-119).  V1 on an empty sequence stores
    // zeros, as the reference kernel does (exp_sum 0, zero accumulators, :333-345, :483-495).
    if constexpr (!PARTITIONED) {
      T* out = reinterpret_cast<T*>(p.out) + ((int64_t)seq * p.num_heads + head) * p.head_size;
      for (int i = tid; i < p.head_size; i += kThreads) out[i] = S::from_f32(0.f);
    }
    return;
  }
  const int tok_end = PARTITIONED ? min(tok_begin + kChunk, seq_len) : seq_len;

  const int D = p.head_size, BS = p.block_size;
  const int kv_head = head / (p.num_heads / p.num_kv_heads);
  const float alibi = p.alibi_slopes ? p.alibi_slopes[head] : 0.f;
  const int* bt = p.block_tables + (int64_t)seq * p.max_num_blocks_per_seq;
  const int64_t head_off = (int64_t)kv_head * p.kv_head_stride;

  // block-sparse mask of KV block b: attended when its sparse block is "remote" or "local"
  int bs_offset = 0, q_bs = 0;
  if constexpr (SPARSE) {
    q_bs = (seq_len - 1) / p.bs_block_size;
    bs_offset = p.bs_head_sliding_step >= 0
                    ? (p.tp_rank * p.num_heads + head) * p.bs_head_sliding_step + 1
                    : (p.tp_rank * p.num_kv_heads + kv_head) * (-p.bs_head_sliding_step) + 1;
  }
  auto attended = [&](int block_idx) -> bool {
    if constexpr (!SPARSE) return true;
    const int k_bs = block_idx * BS / p.bs_block_size;
    return ((k_bs + bs_offset) % p.bs_vert_stride == 0) || (k_bs > q_bs - p.bs_local_blocks);
  };

  __shared__ float q_s[256];
  __shared__ __attribute__((aligned(16))) float logits[kChunk];
  __shared__ float red[kWaves];

  {
    const T* q = reinterpret_cast<const T*>(p.q) + (int64_t)seq * p.q_stride + (int64_t)head * D;
    for (int i = tid; i < D; i += kThreads) q_s[i] = S::to_f32(q[i]);
  }
  __syncthreads();

  // value phase mapping: thread -> (token-block group, head dim)
  const int ngroups = max(1, kThreads / D);
  const int grp = tid / D, d = tid - grp * D;
  const bool pv_active = grp < ngroups;

  float m = -FLT_MAX, l = 0.f, acc = 0.f;

  for (int c0 = tok_begin; c0 < tok_end; c0 += kChunk) {
    const int n = min(kChunk, tok_end - c0);
    // ---- logits
    float lmax = -FLT_MAX;
    for (int i = tid; i < n; i += kThreads) {
      const int t = c0 + i;
      if (!attended(t / BS)) {
        logits[i] = -FLT_MAX;
        continue;
      }
      const int64_t blk = bt[t / BS];
      const int off = t % BS;
      const int64_t base = blk * p.kv_block_stride + head_off + (int64_t)off * X;
      float dot = 0.f;
      for (int ch = 0; ch < D / X; ++ch) {
#pragma unroll
        for (int v = 0; v < X / VEC; ++v) {
          float kk[VEC];
          C::load(p.k_cache, base + (int64_t)ch * BS * X + v * VEC, p.kv_scale, kk);
#pragma unroll
          for (int e = 0; e < VEC; ++e) dot = fmaf(q_s[ch * X + v * VEC + e], kk[e], dot);
        }
      }
      float qk = p.scale * dot;
      if (alibi != 0.f) qk += alibi * (float)(t - seq_len + 1);
      logits[i] = qk;
      lmax = fmaxf(lmax, qk);
    }
    const float cmax = block_max(lmax, red);
    const float m_new = fmaxf(m, cmax);
    const float alpha = __expf(m - m_new);
    float lsum = 0.f;
    for (int i = tid; i < n; i += kThreads) {
      float e = __expf(logits[i] - m_new);
      lsum += e;
      logits[i] = S::to_f32(S::from_f32(e));  // probability as the model dtype sees it
    }
    l = l * alpha + block_sum(lsum, red);  // block_sum's barriers also publish logits[]
    m = m_new;
    acc *= alpha;
    // ---- values
    if (pv_active) {
      const int b0 = c0 / BS, nb = (n + BS - 1) / BS;
      for (int b = grp; b < nb; b += ngroups) {
        if (!attended(b0 + b)) continue;
        const int64_t blk = bt[b0 + b];
        const int64_t row = blk * p.kv_block_stride + head_off + (int64_t)d * BS;
        const int valid = min(BS, n - b * BS);
        for (int e0 = 0; e0 < BS; e0 += VEC) {
          if (e0 >= valid) break;
          float vv[VEC];
          C::load(p.v_cache, row + e0, p.kv_scale, vv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            // tokens past the end may hold NaNs (ref :420-430): never touch them
            const float pe = (e0 + e < valid) ? logits[b * BS + e0 + e] : 0.f;
            const float ve = (e0 + e < valid) ? vv[e] : 0.f;
            acc = fmaf(pe, ve, acc);
          }
        }
      }
    }
    __syncthreads();  // logits[] is rewritten by the next chunk
  }

  // ---- combine the token-block groups
  if (ngroups > 1) {
    if (pv_active && grp > 0) logits[(grp - 1) * D + d] = acc;
    __syncthreads();
    if (grp == 0)
      for (int g = 1; g < ngroups; ++g) acc += logits[(g - 1) * D + d];
  }
  if (grp == 0 && d < D) {
    const float inv = 1.f / (l + 1e-6f);
    T* out = reinterpret_cast<T*>(p.out);
    if (PARTITIONED) {
      const int64_t o = (((int64_t)seq * p.num_heads + head) * p.max_num_partitions + part) * D;
      out[o + d] = S::from_f32(acc * inv);
    } else {
      out[((int64_t)seq * p.num_heads + head) * D + d] = S::from_f32(acc * inv);
    }
  }
  if (PARTITIONED && tid == 0) {
    const int64_t o = ((int64_t)seq * p.num_heads + head) * p.max_num_partitions + part;
    p.exp_sums[o] = l;
    p.max_logits[o] = m;
  }
}

// paged_attention_v2_reduce_kernel (ref :567-669): merge the partitions of one (seq, head).
template <class S>
__global__ __launch_bounds__(128) void pa_v2_reduce_kernel(
    typename S::T* __restrict__ out, const float* __restrict__ exp_sums,
    const float* __restrict__ max_logits, const typename S::T* __restrict__ tmp_out,
    const int* __restrict__ seq_lens, int head_size, int max_num_partitions) {
  using T = typename S::T;
  const int num_heads = gridDim.x, head = blockIdx.x, seq = blockIdx.y;
  const int np = (seq_lens[seq] + kChunk - 1) / kChunk;
  const int64_t sh = (int64_t)seq * num_heads + head;
  const T* tmp = tmp_out + sh * max_num_partitions * head_size;
  T* o = out + sh * head_size;
  if (np == 1) {  // single partition: plain copy (ref :582-594)
    for (int i = threadIdx.x; i < head_size; i += blockDim.x) o[i] = tmp[i];
    return;
  }
  extern __shared__ float smem[];  // [np] rescaled exp sums
  __shared__ float red[2];
  const float* ml = max_logits + sh * max_num_partitions;
  const float* es = exp_sums + sh * max_num_partitions;
  float mx = -FLT_MAX;
  for (int i = threadIdx.x; i < np; i += blockDim.x) mx = fmaxf(mx, ml[i]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(red[0], red[1]);
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    float r = es[i] * expf(ml[i] - mx);
    smem[i] = r;
    sum += r;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + 1e-6f);
  for (int i = threadIdx.x; i < head_size; i += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < np; ++j) acc += S::to_f32(tmp[j * head_size + i]) * smem[j] * inv;
    o[i] = S::from_f32(acc);
  }
}

template <class S, bool FP8>
void launch_generic_t(const AttnParams& p, hipStream_t s) {
  dim3 grid(p.num_heads, p.num_seqs, p.max_num_partitions > 0 ? p.max_num_partitions : 1);
  const bool sparse = p.bs_vert_stride > 1;
  if (p.max_num_partitions > 0) {
    if (sparse)
      hipLaunchKernelGGL((pa_generic_kernel<S, FP8, true, true>), grid, dim3(kThreads), 0, s, p);
    else
      hipLaunchKernelGGL((pa_generic_kernel<S, FP8, true, false>), grid, dim3(kThreads), 0, s, p);
  } else {
    if (sparse)
      hipLaunchKernelGGL((pa_generic_kernel<S, FP8, false, true>), grid, dim3(kThreads), 0, s, p);
    else
      hipLaunchKernelGGL((pa_generic_kernel<S, FP8, false, false>), grid, dim3(kThreads), 0, s, p);
  }
}

}  // namespace

int launch_paged_attention_generic(const AttnParams& p, int dtype, int kv_dtype, hipStream_t s) {
  const bool fp8 = kv_dtype == VLA_KV_FP8_E4M3;
  switch (dtype) {
    case VLA_F16: fp8 ? launch_generic_t<F16, true>(p, s) : launch_generic_t<F16, false>(p, s); break;
    case VLA_BF16: fp8 ? launch_generic_t<BF16, true>(p, s) : launch_generic_t<BF16, false>(p, s); break;
    case VLA_F32: fp8 ? launch_generic_t<F32, true>(p, s) : launch_generic_t<F32, false>(p, s); break;
    default: return VLA_ERR_DTYPE;
  }
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

int launch_paged_attention_v2_reduce(void* out, const float* exp_sums, const float* max_logits,
                                     const void* tmp_out, const int* seq_lens, int num_seqs,
                                     int num_heads, int head_size, int max_num_partitions,
                                     int dtype, hipStream_t s) {
  dim3 grid(num_heads, num_seqs);
  const size_t smem = sizeof(float) * (size_t)max_num_partitions;
  switch (dtype) {
    case VLA_F16:
      hipLaunchKernelGGL(pa_v2_reduce_kernel<F16>, grid, dim3(128), smem, s, (_Float16*)out,
                         exp_sums, max_logits, (const _Float16*)tmp_out, seq_lens, head_size,
                         max_num_partitions);
      break;
    case VLA_BF16:
      hipLaunchKernelGGL(pa_v2_reduce_kernel<BF16>, grid, dim3(128), smem, s, (__bf16*)out,
                         exp_sums, max_logits, (const __bf16*)tmp_out, seq_lens, head_size,
                         max_num_partitions);
      break;
    case VLA_F32:
      hipLaunchKernelGGL(pa_v2_reduce_kernel<F32>, grid, dim3(128), smem, s, (float*)out,
                         exp_sums, max_logits, (const float*)tmp_out, seq_lens, head_size,
                         max_num_partitions);
      break;
    default: return VLA_ERR_DTYPE;
  }
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

}  // namespace vla
