// Shared device helpers for the gfx950 paged-attention backend: dtype traits, the OCP e4m3fn
// KV codec on the hardware converters, wave64 reductions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vla_paged_attention.h"

namespace vla {

constexpr int kWave = 64;  // gfx950 wavefront

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- scalar traits
struct F16 {
  using T = _Float16;
  using V8 = f16x8;
  static constexpr int kCode = VLA_F16;
  static __device__ __forceinline__ float to_f32(T v) { return (float)v; }
  static __device__ __forceinline__ T from_f32(float v) { return (T)v; }
};
struct BF16 {
  using T = __bf16;
  using V8 = bf16x8;
  static constexpr int kCode = VLA_BF16;
  static __device__ __forceinline__ float to_f32(T v) { return (float)v; }
  static __device__ __forceinline__ T from_f32(float v) { return (T)v; }  // v_cvt_pk_bf16_f32: RNE, NaN kept
};
struct F32 {
  using T = float;
  static constexpr int kCode = VLA_F32;
  static __device__ __forceinline__ float to_f32(T v) { return v; }
  static __device__ __forceinline__ T from_f32(float v) { return v; }
};

// ---------------------------------------------------------------- fp8 e4m3fn (OCP) codec
// gfx950 converts OCP fp8 in hardware.  Encode = saturating-finite RNE (what the reference's
// CUDA path asks of __nv_cvt_float_to_fp8 with __NV_SATFINITE,
// csrc/quantization/fp8/nvidia/quant_utils.cuh:482-487): finite |x| > 448 and +-inf clamp to
// +-448, NaN stays NaN (0x7f | sign).
__device__ __forceinline__ uint8_t fp8_e4m3_encode(float x) {
  float c = __builtin_amdgcn_fmed3f(x, 448.0f, -448.0f);
  c = (x != x) ? x : c;  // fmed3 would turn a NaN into a bound
  int packed = __builtin_amdgcn_cvt_pk_fp8_f32(c, c, 0, false);
  return (uint8_t)(packed & 0xff);
}

// two fp8 bytes (low or high half of w) -> two fp32, exact
template <bool HI>
__device__ __forceinline__ f32x2 fp8x2_to_f32(uint32_t w) {
  return __builtin_amdgcn_cvt_pk_f32_fp8(w, HI);
}
__device__ __forceinline__ float fp8_e4m3_decode(uint8_t b) {
  return fp8x2_to_f32<false>((uint32_t)b)[0];
}

// four fp8 bytes -> four 16-bit floats, exact (every e4m3fn value is representable in both
// fp16 and bf16), packed as two dwords
template <class S>
__device__ __forceinline__ u32x2 fp8x4_to_16(uint32_t w);
template <>
__device__ __forceinline__ u32x2 fp8x4_to_16<F16>(uint32_t w) {
  f16x2 lo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
  f16x2 hi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true);
  u32x2 r;
  r[0] = __builtin_bit_cast(uint32_t, lo);
  r[1] = __builtin_bit_cast(uint32_t, hi);
  return r;
}
template <>
__device__ __forceinline__ u32x2 fp8x4_to_16<BF16>(uint32_t w) {
  bf16x2 lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false);
  bf16x2 hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true);
  u32x2 r;
  r[0] = __builtin_bit_cast(uint32_t, lo);
  r[1] = __builtin_bit_cast(uint32_t, hi);
  return r;
}

// ---------------------------------------------------------------- wave64 / block reductions
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// host-side launch helpers
#define VLA_HIP_CHECK(expr)                 \
  do {                                      \
    hipError_t e_ = (expr);                 \
    if (e_ != hipSuccess) {                 \
      vla::set_last_hip_error((int)e_);     \
      return VLA_ERR_HIP;                   \
    }                                       \
  } while (0)

void set_last_hip_error(int e);

inline bool head_size_supported(int d) {
  // vllm/attention/ops/paged_attn.py:33-34, attention_kernels.cu:738-766
  return d == 64 || d == 80 || d == 96 || d == 112 || d == 128 || d == 192 || d == 256;
}
inline bool block_size_supported(int b) { return b == 8 || b == 16 || b == 32; }  // :789-803

struct AttnParams {
  void* out;            // V1: [S, Hq, D]; V2 partition pass: tmp_out [S, Hq, P, D]
  float* exp_sums;      // V2 only
  float* max_logits;    // V2 only
  const void* q;
  const void* k_cache;
  const void* v_cache;
  const int* block_tables;
  const int* seq_lens;
  const int* seq_order;   // optional: grid row y works on sequence seq_order[y] (longest first)
  const float* alibi_slopes;
  int num_seqs, num_heads, num_kv_heads, head_size, block_size;
  int max_num_blocks_per_seq;
  int max_num_partitions;  // 0 = V1 (whole sequence in one workgroup)
  float scale, kv_scale;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  // block-sparse attention (ref attention_kernels.cu:209-254); bs_vert_stride <= 1 = dense
  int bs_vert_stride = 0, bs_local_blocks = 0, bs_block_size = 64, bs_head_sliding_step = 0;
  int tp_rank = 0;
  // K / V loads carry the nontemporal hint (every byte is read once, by one workgroup): set by
  // the entry points from vla_set_kv_load_policy and the size of what the launch will read
  int kv_nontemporal = 0;
};

// implemented in paged_attention_generic.hip / paged_attention_mfma.hip
int launch_paged_attention_generic(const AttnParams& p, int dtype, int kv_dtype, hipStream_t s);
// returns -1 when the shape is not covered by the MFMA path (caller falls back to generic)
// Rotary embedding of this step's query and key and (write_kv) the cache write of the rotated key
// and of the value, done inside the paged-attention launch (paged_attention_mfma.hip, FUSE instantiations):
// token t of the step is sequence t.  AttnParams::q then holds the UNROTATED query (ignored when
// `slabs` is set: q, k, v are columns q_col / k_col / v_col of the fp32 split-K slabs
// [num_slabs][tokens][slab_row] of the qkv projection, summed in slab order and rounded to the
// model dtype first).
struct RopeFuse {
  const int64_t* positions;   // [tokens]
  const int64_t* slots;       // [tokens], < 0: the token is not cached
  const void* cos_sin;        // [max_position][head_size] model dtype: cos | sin (NeoX, whole head)
  const void* k;              // [tokens, k_stride] model dtype (no slabs)
  const void* v;
  int64_t k_stride, v_stride;
  const float* slabs;
  int num_slabs;
  int64_t slab_stride, slab_row;
  int q_col, k_col, v_col;
  int write_kv;               // 0: the key / value of the step are in the cache already; only the query is rotated here
};
bool paged_attention_rope_fusable(int num_heads, int num_kv_heads, int head_size, int block_size,
                                  int dtype);
int launch_paged_attention_mfma(const AttnParams& p, int dtype, int kv_dtype, hipStream_t s,
                                const RopeFuse* fuse = nullptr);
int launch_paged_attention_v2_reduce(void* out, const float* exp_sums, const float* max_logits,
                                     const void* tmp_out, const int* seq_lens, int num_seqs,
                                     int num_heads, int head_size, int max_num_partitions,
                                     int dtype, hipStream_t s);

}  // namespace vla
