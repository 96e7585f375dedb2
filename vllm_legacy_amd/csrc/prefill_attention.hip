// Causal self-attention over packed prompt tokens (prefill) for gfx950, 16-bit models.
//
// Replaces what the reference computes for prompts without a cached prefix
// (vllm/attention/backends/rocm_flash_attn.py:305-383: Triton / CK flash attention or the naive
// masked path; semantics as ref_masked_attention, tests/kernels/test_attention.py:43-55):
// per sequence s with tokens [cu[s], cu[s+1]) and per head, out_i = softmax_j<=i(scale q_i.k_j
// + slope (j - i)) v_j, with grouped KV heads.
//
// One 256-thread workgroup per (128-query tile, head, sequence); wave w owns queries
// 32w .. 32w+31 as two 16-row MFMA tiles.  Keys/values are walked in tiles of 64 staged through
// LDS (shared by the four waves), next tile's global loads in flight during the current tile's
// math.  Both products run on 16x16x32 MFMAs:
//   S^T tile:  A = K   lane(g, r): key r of a 16-key block, dims 8g..8g+7 (+32 ks)   (LDS, b128)
//              B = q^T lane(g, c): query c, the same dims                            (registers)
//              C       lane(g, c): query c, keys 4g..4g+3
//   O tile:    A = P   lane(g, r): query r, keys 8g..8g+7 (+32 kk)                   (LDS, b128)
//              B = V   lane(g, c): dim 16n + c, the same keys    (LDS, two transposed b64 reads)
//              C       lane(g, c): dim 16n + c, queries 4g..4g+3
// so the softmax of a query is a reduction over registers and the four lanes that share c, the
// probability tile changes lanes through a per-wave LDS tile, and V stays row-major in LDS
// (ds_read_b64_tr_b16 delivers the column-major operand).
// Numerics: fp32 logits and running max/sum, exp values rounded to the model dtype before the
// value product, fp32 accumulation, one division at the end.
#include "vla_common.h"

#include <math.h>

namespace vla {
namespace {

constexpr int PF_BM = 128, PF_BN = 64, PF_THREADS = 256;

template <class S>
__device__ __forceinline__ f32x4 pf_mfma(u32x4 a, u32x4 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 pf_mfma<F16>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 pf_mfma<BF16>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <class S>
__device__ __forceinline__ uint32_t pf_pack2(float a, float b) {
  typename S::T x = S::from_f32(a), y = S::from_f32(b);
  return (uint32_t)__builtin_bit_cast(uint16_t, x) |
         ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

struct PrefillParams {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const int* cu_seqlens;
  const float* alibi_slopes;
  int num_heads, num_kv_heads;
  int64_t q_stride, k_stride, v_stride, out_stride;
  float scale;
};

// 16-byte chunk swizzles of the LDS images (row = key, RB bytes per row):
//   K image, read by ds_read_b128 with rows on lanes: 256-byte rows chunk ^ (row & 15);
//            128-byte rows chunk ^ (row & 7) (the row parity picks the half of the bank row)
//   V image, read by ds_read_b64_tr_b16 (8 keys x 32 bytes per half wave): 256-byte rows
//            chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)); 128-byte rows
//            chunk ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2))
template <int D>
__device__ __forceinline__ int k_swz(int row, int ch) {
  return D == 128 ? (ch ^ (row & 15)) : (ch ^ (row & 7));
}
template <int D>
__device__ __forceinline__ int v_swz(int row, int ch) {
  return D == 128 ? (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)))
                  : (ch ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)));
}

template <class S, int D>
__global__ __launch_bounds__(PF_THREADS, 2) void prefill_attention_kernel(PrefillParams p) {
  using T = typename S::T;
  constexpr int RB = D * 2;                 // bytes per K/V row in LDS
  constexpr int CH = D / 8;                 // 16-byte chunks per row
  constexpr int KS = D / 32;                // MFMA k-steps of the S^T product
  constexpr int ND = D / 16;                // output dim tiles
  constexpr int LD = (PF_BN * CH) / PF_THREADS;  // 16-byte loads per thread and tile (K or V)
  constexpr int P_PITCH = PF_BN * 2 + 16;   // bytes per query row of the probability tile
  __shared__ __attribute__((aligned(16))) unsigned char k_lds[PF_BN * RB];
  __shared__ __attribute__((aligned(16))) unsigned char v_lds[PF_BN * RB];
  __shared__ __attribute__((aligned(16))) unsigned char p_all[4 * 32 * P_PITCH];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int head = blockIdx.y, seq = blockIdx.z;
  const int kv_head = head / (p.num_heads / p.num_kv_heads);
  const int seq_start = p.cu_seqlens[seq];
  const int seq_len = p.cu_seqlens[seq + 1] - seq_start;
  const int q0 = blockIdx.x * PF_BM;
  if (q0 >= seq_len) return;
  unsigned char* p_lds = p_all + wave * (32 * P_PITCH);

  const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)seq_start * p.q_stride + (int64_t)head * D;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)seq_start * p.k_stride + (int64_t)kv_head * D;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)seq_start * p.v_stride + (int64_t)kv_head * D;

  // ---- query fragments; rows past the sequence are clamped (their results are never stored)
  u32x4 qf[2][KS];
  int qrow[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    qrow[sub] = q0 + 32 * wave + 16 * sub + c;
    const T* qp = qb + (int64_t)min(qrow[sub], seq_len - 1) * p.q_stride;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[sub][ks] = *reinterpret_cast<const u32x4*>(qp + 8 * g + 32 * ks);
  }
  constexpr float kLog2e = 1.4426950408889634f;
  const float scale2 = p.scale * kLog2e;
  const float slope2 = p.alibi_slopes ? p.alibi_slopes[head] * kLog2e : 0.f;

  f32x4 o[2][ND];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int n = 0; n < ND; ++n) o[sub][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  // causal: keys up to the last query of this tile
  const int k_end = min(seq_len, q0 + PF_BM);
  const int ntiles = (k_end + PF_BN - 1) / PF_BN;

  u32x4 kreg[LD], vreg[LD];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int id = tid + PF_THREADS * i, row = id / CH, ch = id % CH;
      const int64_t tok = min(kt * PF_BN + row, seq_len - 1);
      kreg[i] = *reinterpret_cast<const u32x4*>(kb + tok * p.k_stride + 8 * ch);
      vreg[i] = *reinterpret_cast<const u32x4*>(vb + tok * p.v_stride + 8 * ch);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int id = tid + PF_THREADS * i, row = id / CH, ch = id % CH;
      *reinterpret_cast<u32x4*>(k_lds + row * RB + 16 * k_swz<D>(row, ch)) = kreg[i];
      *reinterpret_cast<u32x4*>(v_lds + row * RB + 16 * v_swz<D>(row, ch)) = vreg[i];
    }
  };

  for (int kt = 0; kt < ntiles; ++kt) {
    // two workgroups share a CU: while this one waits for its tile, the other one computes
    load_tile(kt);
    __syncthreads();                 // every wave is done reading the previous tile
    store_tile();
    __syncthreads();
    const int k0 = kt * PF_BN;
    // this wave's queries all precede the tile: nothing to add (wave-uniform)
    if (k0 > q0 + 32 * wave + 31) continue;

    // ---- online softmax of query c (per query tile) over this lane's keys, in base 2: logits
    // are scaled by scale * log2(e) so that every exponential is a bare v_exp_f32.  A tile that
    // lies entirely below the diagonal of this wave's queries and inside the sequence needs no
    // mask (wave-uniform test).
    const bool interior = (k0 + PF_BN - 1 <= q0 + 32 * wave) && (k0 + PF_BN <= seq_len);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // S^T = K q^T for this query tile: 4 key blocks of 16 (the two query tiles go one after the
      // other so that only one logit tile is live)
      f32x4 st[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int row = 16 * t + c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4 kf = *reinterpret_cast<const u32x4*>(k_lds + row * RB +
                                                           16 * k_swz<D>(row, g + 4 * ks));
          acc = pf_mfma<S>(kf, qf[sub][ks], acc);
        }
        st[t] = acc;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = k0 + 16 * t + 4 * g + i;
          float x = st[t][i] * scale2;
          if (slope2 != 0.f) x += slope2 * (float)(key - qrow[sub]);
          if (!interior) x = (key <= qrow[sub] && key < seq_len) ? x : -INFINITY;
          st[t][i] = x;
          mx = fmaxf(mx, x);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[sub], mx);
      // a query with no live key so far keeps m = -inf: use 0 as the reference point then
      const float m_ref = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run[sub] - m_ref);
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = __builtin_amdgcn_exp2f(st[t][i] - m_ref);
          psum += e[i];
        }
        u32x2 w;
        w[0] = pf_pack2<S>(e[0], e[1]);
        w[1] = pf_pack2<S>(e[2], e[3]);
        *reinterpret_cast<u32x2*>(p_lds + (16 * sub + c) * P_PITCH + (16 * t + 4 * g) * 2) = w;
      }
      l_run[sub] = l_run[sub] * alpha + psum;
      if (__any(m_new > m_run[sub])) {
        float a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __shfl(alpha, 4 * g + i, 64);
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) o[sub][n][i] *= a[i];
      }
      m_run[sub] = m_new;
    }

    // ---- O += P V
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 pf[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
        pf[sub] = *reinterpret_cast<const u32x4*>(p_lds + (16 * sub + c) * P_PITCH +
                                                  (8 * g + 32 * kk) * 2);
      const int r0 = 32 * kk + 8 * g + (c >> 2);   // key row this lane addresses (first 4 keys)
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        const int ch = 2 * n + ((c & 3) >> 1), half = (c & 1) * 8;
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_ptr)(v_lds + r0 * RB + 16 * v_swz<D>(r0, ch) + half));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_ptr)(v_lds + (r0 + 4) * RB + 16 * v_swz<D>(r0 + 4, ch) + half));
        u32x4 vf;
        const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
        vf[0] = lo2[0]; vf[1] = lo2[1]; vf[2] = hi2[0]; vf[3] = hi2[1];
        o[0][n] = pf_mfma<S>(pf[0], vf, o[0][n]);
        o[1][n] = pf_mfma<S>(pf[1], vf, o[1][n]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- normalise and store: lane (g, c) holds dim 16n + c of queries 4g..4g+3 of each tile
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float l = l_run[sub];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float li = __shfl(l, 4 * g + i, 64);
      const int row = q0 + 32 * wave + 16 * sub + 4 * g + i;
      if (row >= seq_len) continue;
      T* op = reinterpret_cast<T*>(p.out) + (int64_t)(seq_start + row) * p.out_stride +
              (int64_t)head * D;
      const float inv = 1.f / li;
#pragma unroll
      for (int n = 0; n < ND; ++n) op[16 * n + c] = S::from_f32(o[sub][n][i] * inv);
    }
  }
}

}  // namespace
}  // namespace vla

using namespace vla;

extern "C" int vla_prefill_attention(void* out, const void* query, const void* key,
                                     const void* value, const int* cu_seqlens, int num_seqs,
                                     int max_seqlen, int num_heads, int num_kv_heads, int head_size,
                                     int64_t query_stride, int64_t key_stride, int64_t value_stride,
                                     int64_t out_stride, float scale, const float* alibi_slopes,
                                     int dtype, void* stream) {
  (void)hipGetLastError();
  if (num_seqs == 0 || max_seqlen == 0) return VLA_OK;
  if (!out || !query || !key || !value || !cu_seqlens || num_seqs < 0 || max_seqlen < 0 ||
      num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads)
    return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16) return VLA_ERR_DTYPE;
  if (head_size != 64 && head_size != 128) return VLA_ERR_HEAD_SIZE;
  if ((query_stride | key_stride | value_stride | out_stride) % 8 || ((uintptr_t)query % 16) ||
      ((uintptr_t)key % 16) || ((uintptr_t)value % 16))
    return VLA_ERR_ARG;
  PrefillParams p{out, query, key, value, cu_seqlens, alibi_slopes, num_heads, num_kv_heads,
                  query_stride, key_stride, value_stride, out_stride, scale};
  dim3 grid((max_seqlen + PF_BM - 1) / PF_BM, num_heads, num_seqs);
  hipStream_t s = (hipStream_t)stream;
#define VLA_PF(ST, DD) \
  hipLaunchKernelGGL((prefill_attention_kernel<ST, DD>), grid, dim3(PF_THREADS), 0, s, p)
  if (dtype == VLA_F16) {
    if (head_size == 128) VLA_PF(F16, 128); else VLA_PF(F16, 64);
  } else {
    if (head_size == 128) VLA_PF(BF16, 128); else VLA_PF(BF16, 64);
  }
#undef VLA_PF
  VLA_HIP_CHECK(hipGetLastError());
  return VLA_OK;
}
