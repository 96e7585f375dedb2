// Routing glue of the quantized Mixtral block for gfx950: what lies between the router's top-k and
// the grouped int4 GEMMs, and between the second GEMM and the block's output
// (vllm/model_executor/models/mixtral_quant.py:129-156).
//
//   moe_route_topk   softmax of the router logits in fp32, top-k, renormalisation (:131-138)
//   moe_sort_pairs   the (token, choice) pairs, bucketed by local expert, in pair order inside a
//                    bucket (a stable counting sort): per-bucket row offsets for the grouped launch,
//                    the token each sorted row reads, and where every pair went
//   moe_combine      final[t] = sum over t's local experts, ascending expert index, of
//                    model_dtype(y[row of (t, e)] * weight(t, e)), every partial sum rounded to the
//                    model dtype -- the arithmetic of `expert_out.mul_(w)` / `final.add_(...)` in
//                    the reference's loop (:141-151), whose skipped experts add exact zeros
//
// Both are index / byte work plus two roundings per element: one launch each instead of the
// argsort, one-hot count, cumulative sum, gathers, scaled copy and scatter-add of a tensor-library
// formulation (about 150 us of small launches per block at decode sizes).
#include "vla_common.h"

namespace vla {
namespace {

constexpr int kSortThreads = 1024;
constexpr int kSortLdsInts = 15 * 1024;   // segment counters: 60 KiB

// One workgroup.  A segment is 64 consecutive pairs (one wave pass); cnt[seg][b] first holds how
// many pairs of segment seg fall into bucket b, then how many pairs of bucket b precede segment
// seg, so a pair's row is offset[b] + cnt[seg][b] + (pairs of b before it inside the segment).
__global__ __launch_bounds__(kSortThreads) void moe_sort_pairs_kernel(
    const int64_t* __restrict__ selected, const int* __restrict__ bucket_of, int num_experts,
    int num_pairs, int top_k, int num_buckets, int* __restrict__ row_offsets,
    int* __restrict__ sorted_token, int* __restrict__ pair_row, int* __restrict__ row_bucket) {
  __shared__ int cnt[kSortLdsInts];
  __shared__ int offs[130];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int nseg = (num_pairs + 63) >> 6;
  auto bucket = [&](int p) -> int {
    if (p >= num_pairs) return -1;
    const int64_t e = selected[p];
    // an index outside the expert range belongs to nobody: the trailing bucket
    return (e >= 0 && e < num_experts) ? bucket_of[e] : num_buckets - 1;
  };
  for (int seg = wave; seg < nseg; seg += nwaves) {
    const int b = bucket(seg * 64 + lane);
    for (int bb = 0; bb < num_buckets; ++bb) {
      const uint64_t mask = __ballot(b == bb);
      if (lane == 0) cnt[seg * num_buckets + bb] = __popcll(mask);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < num_buckets) {
    int run = 0;
    for (int seg = 0; seg < nseg; ++seg) {
      const int c = cnt[seg * num_buckets + threadIdx.x];
      cnt[seg * num_buckets + threadIdx.x] = run;
      run += c;
    }
    offs[threadIdx.x + 1] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    offs[0] = 0;
    for (int b = 0; b < num_buckets; ++b) offs[b + 1] += offs[b];
    for (int b = 0; b <= num_buckets; ++b) row_offsets[b] = offs[b];
  }
  __syncthreads();
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int seg = wave; seg < nseg; seg += nwaves) {
    const int p = seg * 64 + lane;
    const int b = bucket(p);
    int before = 0;
    for (int bb = 0; bb < num_buckets; ++bb) {
      const uint64_t mask = __ballot(b == bb);
      if (b == bb) before = __popcll(mask & below);
    }
    if (b >= 0) {
      const int row = offs[b] + cnt[seg * num_buckets + b] + before;
      sorted_token[row] = p / top_k;
      pair_row[p] = row;
      if (row_bucket) row_bucket[row] = b;
    }
  }
}

// One wave per token, expert e on lane e.  Softmax as a tensor library's warp softmax evaluates
// it (fp32: exp(x - max) / sum, the sum as an xor tree over the lanes), then k rounds of "largest
// remaining value, lowest index among equals" -- the choice a radix-select top-k makes when values
// tie at the boundary -- and the division by the sum of the chosen weights.
template <class S>
__global__ __launch_bounds__(256) void moe_route_topk_kernel(
    const typename S::T* __restrict__ logits, int num_tokens, int num_experts, int top_k,
    float* __restrict__ weights, int64_t* __restrict__ experts) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= num_tokens) return;
  const bool has = lane < num_experts;
  const float x = has ? S::to_f32(logits[(int64_t)tok * num_experts + lane]) : -INFINITY;
  const float mx = wave_max(x);
  const float e = has ? expf(x - mx) : 0.f;
  const float p = e / wave_sum(e);
  float v = has ? p : -INFINITY;
  float chosen_w = 0.f, total = 0.f;    // lane j keeps the j-th choice
  int chosen_e = 0;
  for (int j = 0; j < top_k; ++j) {
    const float best = wave_max(v);
    const uint64_t at = __ballot(v == best);
    const int idx = __ffsll((unsigned long long)at) - 1;
    if (lane == j) {
      chosen_w = best;
      chosen_e = idx;
    }
    total += best;                      // summed in choice order, like a sum over the last dim
    if (lane == idx) v = -INFINITY;
  }
  if (lane < top_k) {
    weights[(int64_t)tok * top_k + lane] = chosen_w / total;
    experts[(int64_t)tok * top_k + lane] = chosen_e;
  }
}

// The router's projection and its top-k in one launch: one workgroup per token computes the NE
// logits x . gate[:, e] (fp32 sums, rounded to the model dtype as the gate's linear layer rounds
// its output, mixtral_quant.py:130-131), then wave 0 runs the softmax / top-k / renormalisation of
// moe_route_topk_kernel on them.  At decode sizes the library GEMM for a [tokens, hidden] x
// [hidden, 8] product and the separate top-k launch cost 13 + 5 us; this is one small launch.
// gate is [hidden][NE] (x @ gate); 8 hidden elements per thread and pass.
template <class S, int NE>
__global__ __launch_bounds__(256) void moe_gate_topk_kernel(
    const typename S::T* __restrict__ x, const typename S::T* __restrict__ gate, int hidden,
    int top_k, float* __restrict__ weights, int64_t* __restrict__ experts,
    typename S::T* __restrict__ logits_out) {
  using T = typename S::T;
  __shared__ float part[4][NE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tok = blockIdx.x;
  float acc[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) acc[e] = 0.f;
  for (int h = threadIdx.x * 8; h < hidden; h += 256 * 8) {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(x + tok * hidden + h);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = S::to_f32(__builtin_bit_cast(T, (uint16_t)(xv[j >> 1] >> (16 * (j & 1)))));
      const T* g = gate + (int64_t)(h + j) * NE;
#pragma unroll
      for (int e8 = 0; e8 < NE; e8 += 8) {
        const u32x4 gv = *reinterpret_cast<const u32x4*>(g + e8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          acc[e8 + e] += xf * S::to_f32(__builtin_bit_cast(T, (uint16_t)(gv[e >> 1] >> (16 * (e & 1)))));
      }
    }
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const float v = wave_sum(acc[e]);
    if (lane == 0) part[wave][e] = v;
  }
  __syncthreads();
  if (wave != 0) return;
  const bool has = lane < NE;
  float xl = -INFINITY;
  if (has) {
    const T r = S::from_f32(part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
    if (logits_out) logits_out[tok * NE + lane] = r;
    xl = S::to_f32(r);
  }
  const float mx = wave_max(xl);
  const float ex = has ? expf(xl - mx) : 0.f;
  const float p = ex / wave_sum(ex);
  float v = has ? p : -INFINITY;
  float chosen_w = 0.f, total = 0.f;    // lane j keeps the j-th choice
  int chosen_e = 0;
  for (int j = 0; j < top_k; ++j) {
    const float best = wave_max(v);
    const uint64_t at = __ballot(v == best);
    const int idx = __ffsll((unsigned long long)at) - 1;
    if (lane == j) {
      chosen_w = best;
      chosen_e = idx;
    }
    total += best;
    if (lane == idx) v = -INFINITY;
  }
  if (lane < top_k) {
    weights[tok * top_k + lane] = chosen_w / total;
    experts[tok * top_k + lane] = chosen_e;
  }
}

// One workgroup per token; 8 elements (16 bytes) per lane and pass.
template <class S, int MAXK>
__global__ __launch_bounds__(256) void moe_combine_kernel(
    const typename S::T* __restrict__ y, const float* __restrict__ weights,
    const int64_t* __restrict__ selected, const int* __restrict__ pair_row,
    const int* __restrict__ bucket_of, int num_experts, int num_local, int top_k, int hidden,
    typename S::T* __restrict__ out) {
  using T = typename S::T;
  const int tok = blockIdx.x;
  // this token's local pairs in ascending expert order (thread 0 sorts at most MAXK entries)
  __shared__ int64_t ex[MAXK];
  __shared__ int row[MAXK];
  __shared__ float w[MAXK];
  __shared__ int cnt_s;
  if (threadIdx.x == 0) {
    int c = 0;
    for (int j = 0; j < top_k && j < MAXK; ++j) {
      const int p = tok * top_k + j;
      const int64_t e = selected[p];
      if (e < 0 || e >= num_experts || bucket_of[e] >= num_local) continue;
      int at = c++;
      while (at > 0 && ex[at - 1] > e) {
        ex[at] = ex[at - 1];
        row[at] = row[at - 1];
        w[at] = w[at - 1];
        --at;
      }
      ex[at] = e;
      row[at] = pair_row[p];
      w[at] = weights[p];
    }
    cnt_s = c;
  }
  __syncthreads();
  const int cntl = cnt_s;
  for (int i = threadIdx.x * 8; i < hidden; i += blockDim.x * 8) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < MAXK; ++s) {
      if (s >= cntl) break;
      const u32x4 v = *reinterpret_cast<const u32x4*>(y + (int64_t)row[s] * hidden + i);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma clang fp contract(off)
        const T yv = __builtin_bit_cast(T, (uint16_t)(v[e >> 1] >> (16 * (e & 1))));
        // fp32 product, THEN the rounding to the model dtype (what a tensor library's mixed
        // half x float multiply does); left to itself the compiler fuses the two into
        // v_fma_mixlo_f16, which rounds the exact product once and differs in the last place
        // when the fp32 rounding lands on a 16-bit midpoint
        float prod = S::to_f32(yv) * w[s];
        asm volatile("" : "+v"(prod));
        const float scaled = S::to_f32(S::from_f32(prod));
        // the first contribution is taken as it is (the reference's `final = current`)
        acc[e] = s == 0 ? scaled : S::to_f32(S::from_f32(acc[e] + scaled));
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = (uint32_t)__builtin_bit_cast(uint16_t, S::from_f32(acc[2 * e])) |
             ((uint32_t)__builtin_bit_cast(uint16_t, S::from_f32(acc[2 * e + 1])) << 16);
    *reinterpret_cast<u32x4*>(out + (int64_t)tok * hidden + i) = o;
  }
}

}  // namespace
}  // namespace vla

using namespace vla;

extern "C" int vla_moe_route_topk(const void* router_logits, int num_tokens, int num_experts,
                                  int top_k, float* routing_weights, int64_t* selected_experts,
                                  int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!router_logits || !routing_weights || !selected_experts || num_tokens < 0 ||
      num_experts <= 0 || num_experts > 64 || top_k <= 0 || top_k > num_experts || top_k > 64)
    return VLA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((num_tokens + 3) / 4);
  if (dtype == VLA_F16)
    hipLaunchKernelGGL((moe_route_topk_kernel<F16>), grid, dim3(256), 0, s,
                       (const _Float16*)router_logits, num_tokens, num_experts, top_k,
                       routing_weights, selected_experts);
  else if (dtype == VLA_BF16)
    hipLaunchKernelGGL((moe_route_topk_kernel<BF16>), grid, dim3(256), 0, s,
                       (const __bf16*)router_logits, num_tokens, num_experts, top_k,
                       routing_weights, selected_experts);
  else if (dtype == VLA_F32)
    hipLaunchKernelGGL((moe_route_topk_kernel<F32>), grid, dim3(256), 0, s,
                       (const float*)router_logits, num_tokens, num_experts, top_k,
                       routing_weights, selected_experts);
  else
    return VLA_ERR_DTYPE;
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_moe_gate_topk(const void* x, const void* gate, int num_tokens, int hidden,
                                 int num_experts, int top_k, float* routing_weights,
                                 int64_t* selected_experts, void* logits_out, int dtype,
                                 void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!x || !gate || !routing_weights || !selected_experts || num_tokens < 0 || hidden <= 0 ||
      hidden % 8 || (num_experts != 8 && num_experts != 16) || top_k <= 0 || top_k > num_experts ||
      ((uintptr_t)x % 16) || ((uintptr_t)gate % 16))
    return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return VLA_ERR_DTYPE;
  hipStream_t s = (hipStream_t)stream;
#define VLA_GATE(S_, NE_)                                                                        \
  hipLaunchKernelGGL((moe_gate_topk_kernel<S_, NE_>), dim3(num_tokens), dim3(256), 0, s,          \
                     (const typename S_::T*)x, (const typename S_::T*)gate, hidden, top_k,       \
                     routing_weights, selected_experts, (typename S_::T*)logits_out)
  if (dtype == VLA_F16) {
    if (num_experts == 8) VLA_GATE(F16, 8); else VLA_GATE(F16, 16);
  } else {
    if (num_experts == 8) VLA_GATE(BF16, 8); else VLA_GATE(BF16, 16);
  }
#undef VLA_GATE
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_moe_sort_pairs(const int64_t* selected_experts, const int* bucket_of,
                                  int num_experts, int num_tokens, int top_k, int num_buckets,
                                  int* row_offsets, int* sorted_token, int* pair_row,
                                  int* row_bucket, void* stream) {
  (void)hipGetLastError();
  if (!selected_experts || !bucket_of || !row_offsets || !sorted_token || !pair_row ||
      num_experts <= 0 || num_tokens < 0 || top_k <= 0 || num_buckets <= 0 || num_buckets > 128)
    return VLA_ERR_ARG;
  const int64_t pairs = (int64_t)num_tokens * top_k;
  const int64_t nseg = (pairs + 63) / 64;
  if (nseg * num_buckets > kSortLdsInts) return VLA_ERR_ARG;   // caller keeps its previous path
  hipLaunchKernelGGL(moe_sort_pairs_kernel, dim3(1), dim3(kSortThreads), 0, (hipStream_t)stream,
                     selected_experts, bucket_of, num_experts, (int)pairs, top_k, num_buckets,
                     row_offsets, sorted_token, pair_row, row_bucket);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

namespace vla {
namespace {
// out[r, c] = x[src_row[r] (or r), perm[row_set[r], c]], 16-bit elements.  One workgroup per row;
// the permutation row is read coalesced, the gathered elements are 2-byte reads of one x row
// (L2-resident: a row is a few KiB and is read cols times).
__global__ __launch_bounds__(256) void moe_gather_perm_kernel(const uint16_t* __restrict__ x,
                                                              const int* __restrict__ src_row,
                                                              const int* __restrict__ row_set,
                                                              const int* __restrict__ perm,
                                                              uint16_t* __restrict__ out, int cols,
                                                              int64_t x_stride) {
  const int r = blockIdx.x;
  const uint16_t* xr = x + (int64_t)(src_row ? src_row[r] : r) * x_stride;
  const int* pr = perm + (int64_t)row_set[r] * cols;
  uint16_t* o = out + (int64_t)r * cols;
  for (int c = 2 * threadIdx.x; c < cols; c += 512) {
    const int2 p = *reinterpret_cast<const int2*>(pr + c);
    const uint32_t lo = xr[p.x], hi = xr[p.y];
    *reinterpret_cast<uint32_t*>(o + c) = lo | (hi << 16);
  }
}
}  // namespace
}  // namespace vla

extern "C" int vla_moe_gather_perm(const void* x, const int* src_row, const int* row_set,
                                   const int* perm, void* out, int rows, int cols, int64_t x_stride,
                                   void* stream) {
  (void)hipGetLastError();
  if (rows == 0) return VLA_OK;
  if (!x || !row_set || !perm || !out || rows < 0 || cols <= 0 || cols % 2 || x_stride < cols ||
      ((uintptr_t)out % 4) || ((uintptr_t)perm % 8))
    return VLA_ERR_ARG;
  hipLaunchKernelGGL(vla::moe_gather_perm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x, src_row, row_set, perm, (uint16_t*)out, cols, x_stride);
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}

extern "C" int vla_moe_combine(const void* y, const float* routing_weights,
                               const int64_t* selected_experts, const int* pair_row,
                               const int* bucket_of, int num_experts, int num_local, int num_tokens,
                               int top_k, int hidden, void* out, int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_tokens == 0) return VLA_OK;
  if (!y || !routing_weights || !selected_experts || !pair_row || !bucket_of || !out ||
      num_tokens < 0 || top_k <= 0 || top_k > 8 || hidden <= 0 || hidden % 8 ||
      ((uintptr_t)y % 16) || ((uintptr_t)out % 16))
    return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return VLA_ERR_DTYPE;
  hipStream_t s = (hipStream_t)stream;
#define VLA_COMBINE(S_, K_)                                                                        \
  hipLaunchKernelGGL((moe_combine_kernel<S_, K_>), dim3(num_tokens), dim3(256), 0, s,              \
                     (const typename S_::T*)y, routing_weights, selected_experts, pair_row,        \
                     bucket_of, num_experts, num_local, top_k, hidden, (typename S_::T*)out)
  if (dtype == VLA_F16) {
    if (top_k <= 2) VLA_COMBINE(F16, 2); else VLA_COMBINE(F16, 8);
  } else {
    if (top_k <= 2) VLA_COMBINE(BF16, 2); else VLA_COMBINE(BF16, 8);
  }
#undef VLA_COMBINE
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}
