/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's decode hot path.
 *
 * This file is the checker for the HIP kernels, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Each
 * function cites the reference lines (paths relative to the reference root) whose
 * arithmetic it restates.  It is written as plain scalar C (optionally OpenMP over
 * independent (sequence, head) pairs) so that it can be read against the reference.
 *
 * Pinning: the reference holds no golden vectors for this path (its kernel tests
 * are randomized-vs-PyTorch, tests/kernels/test_attention.py:114-274), so this
 * restatement is pinned against outputs of the reference's own CPU kernels
 * compiled from the reference tree (oracle/build_ref.py -> oracle/_ref/) and the
 * fixtures generated from them under tests/golden/: paged attention (f32, bf16),
 * reshape_and_cache, copy_blocks, RMSNorm / rotary / SiLU-and-mul, the FNUZ codec.
 * The GPTQ functions (dequantise, exllama shuffle, act-order) are pinned against the
 * reference's own Python packer (quantize_weights / gptq_pack / sort_weights of
 * vllm/model_executor/layers/quantization/utils/quant_utils.py:39-146) through the
 * fixtures tests/golden/gptq_*.npz made by tests/golden/make_golden_quant.py.
 * Parity unpinned (no reference fixture, no reference CPU implementation): fp16 and
 * fp8-KV attention, the block-sparse mask, swap_blocks, the AWQ functions (the
 * reference's AWQ kernels are CUDA/PTX only and it holds no packer for the format),
 * and prompt attention with or without a cached prefix (checked against a float64
 * evaluation of the definition the reference's tests use instead).
 *
 * dtype codes: 0 = float16, 1 = bfloat16, 2 = float32.
 * kv codes:    0 = "auto" (cache holds the model dtype), 1 = fp8 e4m3 (OCP e4m3fn,
 *              the format of the reference's CUDA path,
 *              csrc/quantization/fp8/nvidia/quant_utils.cuh:456-487, and of gfx950
 *              hardware), 2 = fp8 e4m3 FNUZ (the format of the reference's ROCm path,
 *              csrc/quantization/fp8/amd/hip_float8.h:41-44; kept for cross-checks).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define DT_F16 0
#define DT_BF16 1
#define DT_F32 2
#define KV_AUTO 0
#define KV_FP8_E4M3 1
#define KV_FP8_E4M3_FNUZ 2

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---------------------------------------------------------------- f16 / bf16 */

float oracle_f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    float v = (float)m * (1.0f / 16777216.0f); /* m * 2^-24 */
    return sign ? -v : v;
  }
  if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
  return u2f(sign | ((e + 112) << 23) | (m << 13));
}

/* round-to-nearest-even, overflow to inf, NaN kept quiet */
uint16_t oracle_f32_to_f16(float f) {
  uint32_t u = f2u(f);
  uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((a >> 13) & 0x3ff));
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* >= 65520 -> inf */
  if (a < 0x38800000u) { /* < 2^-14: subnormal half, step 2^-24 */
    float scaled = u2f(a) * 16777216.0f;
    uint32_t m = (uint32_t)rintf(scaled); /* RNE in the default rounding mode */
    return (uint16_t)(sign | m);         /* m == 1024 encodes 2^-14 correctly */
  }
  uint32_t r = a + 0xfffu + ((a >> 13) & 1u);
  return (uint16_t)(sign | (((r >> 23) - 112) << 10) | ((r >> 13) & 0x3ff));
}

float oracle_bf16_to_f32(uint16_t b) { return u2f((uint32_t)b << 16); }

uint16_t oracle_f32_to_bf16(float f) {
  uint32_t u = f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static inline float load_scalar(const void* p, int64_t i, int dt) {
  if (dt == DT_F16) return oracle_f16_to_f32(((const uint16_t*)p)[i]);
  if (dt == DT_BF16) return oracle_bf16_to_f32(((const uint16_t*)p)[i]);
  return ((const float*)p)[i];
}
static inline void store_scalar(void* p, int64_t i, int dt, float v) {
  if (dt == DT_F16) ((uint16_t*)p)[i] = oracle_f32_to_f16(v);
  else if (dt == DT_BF16) ((uint16_t*)p)[i] = oracle_f32_to_bf16(v);
  else ((float*)p)[i] = v;
}
/* value as it would be after a round trip through the model dtype */
static inline float round_scalar(float v, int dt) {
  if (dt == DT_F16) return oracle_f16_to_f32(oracle_f32_to_f16(v));
  if (dt == DT_BF16) return oracle_bf16_to_f32(oracle_f32_to_bf16(v));
  return v;
}
static inline int scalar_bytes(int dt) { return dt == DT_F32 ? 4 : 2; }

/* ------------------------------------------------------------------ fp8 codecs */

/* OCP e4m3fn: bias 7, max 448, 0x7f/0xff NaN, no inf.  Saturating-finite RNE encode
 * = __nv_cvt_float_to_fp8(x, __NV_SATFINITE, __NV_E4M3) of the reference's CUDA path
 * (csrc/quantization/fp8/nvidia/quant_utils.cuh:482-487). */
uint8_t oracle_e4m3fn_encode(float f) {
  uint32_t u = f2u(f);
  uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint8_t)(sign | 0x7f);
  if (a >= 0x43e00000u) return (uint8_t)(sign | 0x7e); /* |x| >= 448 (and inf) */
  if (a < 0x3c800000u) { /* |x| < 2^-6: subnormal, step 2^-9 */
    uint32_t m = (uint32_t)rintf(u2f(a) * 512.0f);
    return (uint8_t)(sign | m);
  }
  uint32_t r = a + 0x7ffffu + ((a >> 20) & 1u);
  return (uint8_t)(sign | (((r >> 23) - 120) << 3) | ((r >> 20) & 7u));
}

float oracle_e4m3fn_decode(uint8_t b) {
  uint32_t e = (b >> 3) & 0xf, m = b & 7;
  float v;
  if ((b & 0x7f) == 0x7f) return u2f(0x7fc00000u | ((uint32_t)(b & 0x80) << 24));
  if (e == 0) v = (float)m * (1.0f / 512.0f);
  else v = u2f(((e + 120) << 23) | (m << 20));
  return (b & 0x80) ? -v : v;
}

/* e4m3 FNUZ: bias 8, max 240, 0x80 = NaN, no -0, no inf.  Restates
 * to_float8<4,3,float,true,true> / from_float8<4,3,float,true>
 * (csrc/quantization/fp8/amd/hip_float8_impl.h:51-222, :224-314): RNE, clip to
 * +-240, inf/NaN -> 0x80, values that round to zero -> 0x00. */
uint8_t oracle_e4m3fnuz_encode(float f) {
  uint32_t u = f2u(f);
  uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
  uint32_t a = u & 0x7fffffffu;
  if (a >= 0x7f800000u) return 0x80;
  if (a >= 0x43700000u) return (uint8_t)(sign | 0x7f); /* |x| >= 240 -> clip */
  if (a < 0x3c000000u) { /* |x| < 2^-7: subnormal, step 2^-10 */
    uint32_t m = (uint32_t)rintf(u2f(a) * 1024.0f);
    return m == 0 ? 0 : (uint8_t)(sign | m);
  }
  uint32_t r = a + 0x7ffffu + ((a >> 20) & 1u);
  uint32_t code = (((r >> 23) - 119) << 3) | ((r >> 20) & 7u);
  if (code > 0x7f) code = 0x7f; /* 248 and up rounds past the top code: clip */
  return (uint8_t)(sign | code);
}

float oracle_e4m3fnuz_decode(uint8_t b) {
  uint32_t e = (b >> 3) & 0xf, m = b & 7;
  float v;
  if (b == 0x80) return u2f(0x7f800001u);
  if (e == 0) v = (float)m * (1.0f / 1024.0f);
  else v = u2f(((e + 119) << 23) | (m << 20));
  return (b & 0x80) ? -v : v;
}

static inline uint8_t fp8_encode(float v, int kv) {
  return kv == KV_FP8_E4M3_FNUZ ? oracle_e4m3fnuz_encode(v) : oracle_e4m3fn_encode(v);
}
static inline float fp8_decode(uint8_t b, int kv) {
  return kv == KV_FP8_E4M3_FNUZ ? oracle_e4m3fnuz_decode(b) : oracle_e4m3fn_decode(b);
}

/* Cache element -> the model-dtype value the kernel works with.
 * fp8: scaled_convert<scalar_t, uint8_t>(x, kv_scale) = round_to_scalar(float(fp8) * kv_scale)
 * (csrc/quantization/fp8/amd/quant_utils.cuh:310-318, :377-385, :427-433). */
static inline float load_cache(const void* cache, int64_t i, int dt, int kv, float kv_scale) {
  if (kv == KV_AUTO) return load_scalar(cache, i, dt);
  return round_scalar(fp8_decode(((const uint8_t*)cache)[i], kv) * kv_scale, dt);
}

/* convert_fp8 (csrc/cache_kernels.cu:350-360, :370-420): elementwise, direction chosen by
 * which side is the byte tensor.  to_fp8 != 0: hp -> fp8(x / scale); else fp8 -> hp(x * scale). */
void oracle_convert_fp8(void* dst, const void* src, int64_t n, int dt, int kv, float scale,
                        int to_fp8) {
  for (int64_t i = 0; i < n; ++i) {
    if (to_fp8) ((uint8_t*)dst)[i] = fp8_encode(load_scalar(src, i, dt) / scale, kv);
    else store_scalar(dst, i, dt, fp8_decode(((const uint8_t*)src)[i], kv) * scale);
  }
}

/* --------------------------------------------------------------- cache updates */

/* reshape_and_cache (csrc/cache_kernels.cu:151-202; CPU twin csrc/cpu/cache.cpp:36-82).
 * key/value: [T, H, D] with token strides; key_cache [NB, H, D/x, BS, x];
 * value_cache [NB, H, D, BS]; slot < 0 is padding and skipped (:164-167). */
void oracle_reshape_and_cache(const void* key, const void* value, void* key_cache,
                              void* value_cache, const int64_t* slot_mapping, int num_tokens,
                              int num_heads, int head_size, int block_size, int x,
                              int64_t key_stride, int64_t value_stride, int dt, int kv,
                              float kv_scale) {
  for (int64_t t = 0; t < num_tokens; ++t) {
    int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    int64_t block_idx = slot / block_size, block_off = slot % block_size;
    for (int i = 0; i < num_heads * head_size; ++i) {
      int h = i / head_size, d = i % head_size;
      int64_t tk = block_idx * num_heads * (head_size / x) * block_size * x +
                   (int64_t)h * (head_size / x) * block_size * x +
                   (int64_t)(d / x) * block_size * x + block_off * x + d % x;
      int64_t tv = block_idx * num_heads * head_size * block_size +
                   (int64_t)h * head_size * block_size + (int64_t)d * block_size + block_off;
      if (kv == KV_AUTO) {
        int eb = scalar_bytes(dt);
        memcpy((char*)key_cache + tk * eb, (const char*)key + (t * key_stride + i) * eb, eb);
        memcpy((char*)value_cache + tv * eb, (const char*)value + (t * value_stride + i) * eb, eb);
      } else {
        ((uint8_t*)key_cache)[tk] = fp8_encode(load_scalar(key, t * key_stride + i, dt) / kv_scale, kv);
        ((uint8_t*)value_cache)[tv] = fp8_encode(load_scalar(value, t * value_stride + i, dt) / kv_scale, kv);
      }
    }
  }
}

/* copy_blocks for one layer (csrc/cache_kernels.cu:70-95): pairs applied in order. */
void oracle_copy_blocks(void* key_cache, void* value_cache, const int64_t* block_mapping,
                        int num_pairs, int64_t block_bytes) {
  for (int p = 0; p < num_pairs; ++p) {
    int64_t s = block_mapping[2 * p], d = block_mapping[2 * p + 1];
    memmove((char*)key_cache + d * block_bytes, (char*)key_cache + s * block_bytes, block_bytes);
    memmove((char*)value_cache + d * block_bytes, (char*)value_cache + s * block_bytes, block_bytes);
  }
}

/* swap_blocks (csrc/cache_kernels.cu:25-64): dst[d] = src[s] per pair. */
void oracle_swap_blocks(const void* src, void* dst, const int64_t* block_mapping, int num_pairs,
                        int64_t block_bytes) {
  for (int p = 0; p < num_pairs; ++p) {
    int64_t s = block_mapping[2 * p], d = block_mapping[2 * p + 1];
    memcpy((char*)dst + d * block_bytes, (const char*)src + s * block_bytes, block_bytes);
  }
}

/* -------------------------------------------------------------- paged attention */

/* One (sequence, head, token range) of the reference kernel body
 * (csrc/attention/attention_kernels.cu:90-496): fp32 logits = scale * q.k (+ alibi, :294-297),
 * max (:313-330), __expf and sum (:333-339), inv = 1 / (sum + 1e-6) (:342), probabilities
 * rounded to the model dtype before the value product (:398-400), fp32 accumulate, output
 * rounded to the model dtype (:492).  Returns max and exp_sum through the pointers. */
/* Block-sparse attention (csrc/attention/attention_kernels.cu:209-254, :385-393): a KV block is
 * attended when its sparse block is "remote" ((k_bs + offset) % vert_stride == 0) or "local"
 * (k_bs > q_bs - local_blocks); skipped blocks get logit -FLT_MAX and no value product.
 * offset slides with the query head (head_sliding_step >= 0) or the KV head (< 0). */
typedef struct { int vert_stride, local_blocks, bs_block_size, offset; } BsCtx;

static inline int bs_attended(const BsCtx* bs, int block_idx, int block_size, int seq_len) {
  if (!bs) return 1;
  int k_bs = block_idx * block_size / bs->bs_block_size;
  int q_bs = (seq_len - 1) / bs->bs_block_size;
  return ((k_bs + bs->offset) % bs->vert_stride == 0) || (k_bs > q_bs - bs->local_blocks);
}

static void attend_range(const BsCtx* bs,
                         float* out_f32 /*[D], written already rounded to dt*/, const float* qf,
                         const void* k_cache, const void* v_cache, int kv_head, int head_size,
                         int block_size, int x, float scale, const int* block_table, int seq_len,
                         int tok_begin, int tok_end, float alibi_slope, int64_t kv_block_stride,
                         int64_t kv_head_stride, int dt, int kv, float kv_scale, float* logits,
                         float* max_out, float* sum_out) {
  int n = tok_end - tok_begin;
  float qk_max = -3.402823466e+38f;
  for (int t = tok_begin; t < tok_end; ++t) {
    if (!bs_attended(bs, t / block_size, block_size, seq_len)) {
      logits[t - tok_begin] = -3.402823466e+38f;
      continue;
    }
    int64_t blk = block_table[t / block_size];
    int off = t % block_size;
    int64_t base = blk * kv_block_stride + (int64_t)kv_head * kv_head_stride;
    float dot = 0.f;
    for (int d = 0; d < head_size; ++d) {
      int64_t ki = base + (int64_t)(d / x) * block_size * x + (int64_t)off * x + d % x;
      dot += qf[d] * load_cache(k_cache, ki, dt, kv, kv_scale);
    }
    float qk = scale * dot;
    if (alibi_slope != 0.f) qk += alibi_slope * (float)(t - seq_len + 1);
    logits[t - tok_begin] = qk;
    qk_max = fmaxf(qk_max, qk);
  }
  float exp_sum = 0.f;
  for (int i = 0; i < n; ++i) {
    float v = expf(logits[i] - qk_max);
    logits[i] = v;
    exp_sum += v;
  }
  float inv = 1.f / (exp_sum + 1e-6f);
  for (int i = 0; i < n; ++i) logits[i] = round_scalar(logits[i] * inv, dt);
  for (int d = 0; d < head_size; ++d) {
    float acc = 0.f;
    for (int t = tok_begin; t < tok_end; ++t) {
      if (!bs_attended(bs, t / block_size, block_size, seq_len)) continue;
      int64_t blk = block_table[t / block_size];
      int off = t % block_size;
      int64_t vi = blk * kv_block_stride + (int64_t)kv_head * kv_head_stride +
                   (int64_t)d * block_size + off;
      acc += logits[t - tok_begin] * load_cache(v_cache, vi, dt, kv, kv_scale);
    }
    out_f32[d] = round_scalar(acc, dt);
  }
  *max_out = qk_max;
  *sum_out = exp_sum;
}

/* paged_attention_v1 (csrc/attention/attention_kernels.cu:805-825, launcher :693-767).
 * out,q: [num_seqs, num_heads, head_size] (q row stride q_stride elements);
 * returns 0, or 1 for an unsupported head/block size (the reference raises, :763-765, :800-802). */
/* bsp: NULL (dense) or {tp_rank, local_blocks, vert_stride, bs_block_size, head_sliding_step} */
static BsCtx bs_ctx(const int* bsp, int num_heads, int num_kv_heads, int head, int kv_head) {
  BsCtx c = {bsp[2], bsp[1], bsp[3], 0};
  c.offset = bsp[4] >= 0 ? (bsp[0] * num_heads + head) * bsp[4] + 1
                         : (bsp[0] * num_kv_heads + kv_head) * (-bsp[4]) + 1;
  return c;
}

static int pa_v1_impl(void* out, const void* q, const void* k_cache, const void* v_cache,
                      int num_seqs, int num_heads, int num_kv_heads, int head_size,
                      int block_size, int x, float scale, const int* block_tables,
                      const int* seq_lens, int max_blocks_per_seq, const float* alibi_slopes,
                      int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                      int dt, int kv, float kv_scale, const int* bsp) {
  if (num_heads % num_kv_heads) return 1;
  int qpk = num_heads / num_kv_heads;
  int max_len = 0;
  for (int s = 0; s < num_seqs; ++s) if (seq_lens[s] > max_len) max_len = seq_lens[s];
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (size_t)(max_len + 1));
    float* qf = (float*)malloc(sizeof(float) * (size_t)head_size);
    float* of = (float*)malloc(sizeof(float) * (size_t)head_size);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int s = 0; s < num_seqs; ++s) {
      for (int h = 0; h < num_heads; ++h) {
        int seq_len = seq_lens[s];
        float mx, sm;
        for (int d = 0; d < head_size; ++d)
          qf[d] = load_scalar(q, (int64_t)s * q_stride + (int64_t)h * head_size + d, dt);
        BsCtx bc;
        if (bsp) bc = bs_ctx(bsp, num_heads, num_kv_heads, h, h / qpk);
        attend_range(bsp ? &bc : NULL, of, qf, k_cache, v_cache, h / qpk, head_size, block_size,
                     x, scale, block_tables + (int64_t)s * max_blocks_per_seq, seq_len, 0, seq_len,
                     alibi_slopes ? alibi_slopes[h] : 0.f, kv_block_stride, kv_head_stride, dt,
                     kv, kv_scale, logits, &mx, &sm);
        for (int d = 0; d < head_size; ++d)
          store_scalar(out, ((int64_t)s * num_heads + h) * head_size + d, dt, of[d]);
      }
    }
    free(logits); free(qf); free(of);
  }
  return 0;
}

int oracle_paged_attention_v1(void* out, const void* q, const void* k_cache, const void* v_cache,
                              int num_seqs, int num_heads, int num_kv_heads, int head_size,
                              int block_size, int x, float scale, const int* block_tables,
                              const int* seq_lens, int max_blocks_per_seq, const float* alibi_slopes,
                              int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                              int dt, int kv, float kv_scale) {
  return pa_v1_impl(out, q, k_cache, v_cache, num_seqs, num_heads, num_kv_heads, head_size,
                    block_size, x, scale, block_tables, seq_lens, max_blocks_per_seq, alibi_slopes,
                    q_stride, kv_block_stride, kv_head_stride, dt, kv, kv_scale, NULL);
}

/* the block-sparse form (is_block_sparse = vert_stride > 1, attention_kernels.cu:821) */
int oracle_paged_attention_v1_bs(void* out, const void* q, const void* k_cache, const void* v_cache,
                                 int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                 int block_size, int x, float scale, const int* block_tables,
                                 const int* seq_lens, int max_blocks_per_seq,
                                 const float* alibi_slopes, int64_t q_stride,
                                 int64_t kv_block_stride, int64_t kv_head_stride, int dt, int kv,
                                 float kv_scale, const int* bsp) {
  return pa_v1_impl(out, q, k_cache, v_cache, num_seqs, num_heads, num_kv_heads, head_size,
                    block_size, x, scale, block_tables, seq_lens, max_blocks_per_seq, alibi_slopes,
                    q_stride, kv_block_stride, kv_head_stride, dt, kv, kv_scale,
                    bsp && bsp[2] > 1 ? bsp : NULL);
}

/* paged_attention_v2 (csrc/attention/attention_kernels.cu:984-1014; partition kernel :534-562
 * over 512-token partitions, :846; reduce kernel :567-669).  exp_sums/max_logits:
 * [num_seqs, num_heads, P] f32, tmp_out: [num_seqs, num_heads, P, head_size] model dtype, P =
 * max_num_partitions.  Partitions past a sequence's end are left untouched (:116-119). */
static int pa_v2_impl(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                      const void* q, const void* k_cache, const void* v_cache,
                      int num_seqs, int num_heads, int num_kv_heads, int head_size,
                      int block_size, int x, float scale, const int* block_tables,
                      const int* seq_lens, int max_blocks_per_seq, int max_num_partitions,
                      const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                      int64_t kv_head_stride, int dt, int kv, float kv_scale, const int* bsp) {
  const int PART = 512;
  if (num_heads % num_kv_heads) return 1;
  int qpk = num_heads / num_kv_heads;
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (PART + 1));
    float* qf = (float*)malloc(sizeof(float) * (size_t)head_size);
    float* of = (float*)malloc(sizeof(float) * (size_t)head_size);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int s = 0; s < num_seqs; ++s) {
      for (int h = 0; h < num_heads; ++h) {
        int seq_len = seq_lens[s];
        int np = (seq_len + PART - 1) / PART;
        int64_t sh = (int64_t)s * num_heads + h;
        for (int d = 0; d < head_size; ++d)
          qf[d] = load_scalar(q, (int64_t)s * q_stride + (int64_t)h * head_size + d, dt);
        for (int p = 0; p < np && p < max_num_partitions; ++p) {
          int b = p * PART, e = b + PART < seq_len ? b + PART : seq_len;
          float mx, sm;
          BsCtx bc;
          if (bsp) bc = bs_ctx(bsp, num_heads, num_kv_heads, h, h / qpk);
          attend_range(bsp ? &bc : NULL, of, qf, k_cache, v_cache, h / qpk, head_size, block_size,
                       x, scale, block_tables + (int64_t)s * max_blocks_per_seq, seq_len, b, e,
                       alibi_slopes ? alibi_slopes[h] : 0.f, kv_block_stride, kv_head_stride, dt,
                       kv, kv_scale, logits, &mx, &sm);
          max_logits[sh * max_num_partitions + p] = mx;
          exp_sums[sh * max_num_partitions + p] = sm;
          for (int d = 0; d < head_size; ++d)
            store_scalar(tmp_out, (sh * max_num_partitions + p) * head_size + d, dt, of[d]);
        }
        /* reduce (:567-669) */
        if (np == 1) {
          for (int d = 0; d < head_size; ++d)
            store_scalar(out, sh * head_size + d, dt,
                         load_scalar(tmp_out, sh * max_num_partitions * head_size + d, dt));
          continue;
        }
        float gmax = -3.402823466e+38f, gsum = 0.f;
        for (int p = 0; p < np; ++p) gmax = fmaxf(gmax, max_logits[sh * max_num_partitions + p]);
        for (int p = 0; p < np; ++p) {
          float r = exp_sums[sh * max_num_partitions + p] *
                    expf(max_logits[sh * max_num_partitions + p] - gmax);
          logits[p] = r;
          gsum += r;
        }
        float inv = 1.f / (gsum + 1e-6f);
        for (int d = 0; d < head_size; ++d) {
          float acc = 0.f;
          for (int p = 0; p < np; ++p)
            acc += load_scalar(tmp_out, (sh * max_num_partitions + p) * head_size + d, dt) *
                   logits[p] * inv;
          store_scalar(out, sh * head_size + d, dt, acc);
        }
      }
    }
    free(logits); free(qf); free(of);
  }
  return 0;
}

int oracle_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                              const void* q, const void* k_cache, const void* v_cache,
                              int num_seqs, int num_heads, int num_kv_heads, int head_size,
                              int block_size, int x, float scale, const int* block_tables,
                              const int* seq_lens, int max_blocks_per_seq, int max_num_partitions,
                              const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                              int64_t kv_head_stride, int dt, int kv, float kv_scale) {
  return pa_v2_impl(out, exp_sums, max_logits, tmp_out, q, k_cache, v_cache, num_seqs, num_heads,
                    num_kv_heads, head_size, block_size, x, scale, block_tables, seq_lens,
                    max_blocks_per_seq, max_num_partitions, alibi_slopes, q_stride,
                    kv_block_stride, kv_head_stride, dt, kv, kv_scale, NULL);
}

int oracle_paged_attention_v2_bs(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                                 const void* q, const void* k_cache, const void* v_cache,
                                 int num_seqs, int num_heads, int num_kv_heads, int head_size,
                                 int block_size, int x, float scale, const int* block_tables,
                                 const int* seq_lens, int max_blocks_per_seq,
                                 int max_num_partitions, const float* alibi_slopes,
                                 int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                 int dt, int kv, float kv_scale, const int* bsp) {
  return pa_v2_impl(out, exp_sums, max_logits, tmp_out, q, k_cache, v_cache, num_seqs, num_heads,
                    num_kv_heads, head_size, block_size, x, scale, block_tables, seq_lens,
                    max_blocks_per_seq, max_num_partitions, alibi_slopes, q_stride,
                    kv_block_stride, kv_head_stride, dt, kv, kv_scale,
                    bsp && bsp[2] > 1 ? bsp : NULL);
}

/* ------------------------------------------------------------- prompt (prefill) attention */

/* Causal self-attention over packed prompt tokens, the expectation of the reference's kernel
 * tests for the prompt path (ref_masked_attention, tests/kernels/test_attention.py:43-55, used
 * per sequence with a lower-triangular mask, :333-366): logits = scale * q.k in fp32 (+ ALiBi
 * bias slope * (j - i), vllm/attention/backends/rocm_flash_attn.py:153-173), softmax in fp32
 * rounded to the model dtype, value product accumulated in fp32, output rounded.
 * q/out: [T, Hq, D], k/v: [T, Hkv, D] with row strides in elements; cu_seqlens [num_seqs + 1]. */
void oracle_prefill_attention(void* out, const void* q, const void* k, const void* v,
                              const int* cu_seqlens, int num_seqs, int num_heads, int num_kv_heads,
                              int head_size, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                              int64_t out_stride, float scale, const float* alibi_slopes, int dt) {
  int qpk = num_heads / num_kv_heads;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int s = 0; s < num_seqs; ++s) {
    for (int h = 0; h < num_heads; ++h) {
      int start = cu_seqlens[s], n = cu_seqlens[s + 1] - start;
      float* logits = (float*)malloc(sizeof(float) * (size_t)(n + 1));
      float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
      int kvh = h / qpk;
      for (int i = 0; i < n; ++i) {
        float mx = -3.402823466e+38f;
        for (int j = 0; j <= i; ++j) {
          float dot = 0.f;
          for (int d = 0; d < head_size; ++d)
            dot += load_scalar(q, (int64_t)(start + i) * q_stride + (int64_t)h * head_size + d, dt) *
                   load_scalar(k, (int64_t)(start + j) * k_stride + (int64_t)kvh * head_size + d, dt);
          float x = scale * dot;
          if (alibi_slopes) x += alibi_slopes[h] * (float)(j - i);
          logits[j] = x;
          mx = fmaxf(mx, x);
        }
        float sum = 0.f;
        for (int j = 0; j <= i; ++j) { logits[j] = expf(logits[j] - mx); sum += logits[j]; }
        for (int d = 0; d < head_size; ++d) acc[d] = 0.f;
        for (int j = 0; j <= i; ++j) {
          float pj = round_scalar(logits[j] / sum, dt);
          for (int d = 0; d < head_size; ++d)
            acc[d] += pj * load_scalar(v, (int64_t)(start + j) * v_stride +
                                              (int64_t)kvh * head_size + d, dt);
        }
        for (int d = 0; d < head_size; ++d)
          store_scalar(out, (int64_t)(start + i) * out_stride + (int64_t)h * head_size + d, dt,
                       acc[d]);
      }
      free(logits); free(acc);
    }
  }
}

/* Prompt attention over a cached prefix: the new tokens of sequence s attend the ctx_lens[s]
 * tokens already in the paged cache and, causally, each other
 * (vllm/attention/ops/prefix_prefill.py:54-230 and :443-680; called from
 * vllm/attention/ops/paged_attn.py:185-216).  Key j of the sequence (absolute position) is read
 * from the cache when j < ctx and from k/v row query_start_loc[s] + j - ctx otherwise; query i
 * sits at position ctx + i.  ALiBi adds slope * (j - (ctx + i)) (:524-565, :608-638); a sliding
 * window W > 0 keeps keys with (ctx + i) - j < W (:130-147, :199-203; its -10000 substitute is a
 * zero weight whenever the diagonal key is live, which it always is).
 * The reference kernel bitcasts a byte cache to e5m2 without a scale (:119-120, :169-170) although
 * reshape_and_cache wrote e4m3 / kv_scale; this restatement decodes with the codec that wrote the
 * cache (load_cache above), which is what the decode kernels do.
 * Probabilities are normalised, then rounded to the value dtype before the value product (:166-172). */
void oracle_prefix_prefill_attention(void* out, const void* q, const void* k, const void* v,
                                     const void* k_cache, const void* v_cache,
                                     const int* block_tables, int max_blocks_per_seq,
                                     const int* query_start_loc, const int* seq_lens,
                                     const int* ctx_lens, int num_seqs, int num_heads,
                                     int num_kv_heads, int head_size, int block_size, int x,
                                     int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                     int64_t out_stride, int64_t kv_block_stride,
                                     int64_t kv_head_stride, float scale, const float* alibi_slopes,
                                     int sliding_window, int dt, int kv, float kv_scale) {
  int qpk = num_heads / num_kv_heads;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int s = 0; s < num_seqs; ++s) {
    for (int h = 0; h < num_heads; ++h) {
      int start = query_start_loc[s], ctx = ctx_lens[s], n = seq_lens[s] - ctx;
      int total = ctx + n;
      float* logits = (float*)malloc(sizeof(float) * (size_t)(total + 1));
      float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
      float* kd = (float*)malloc(sizeof(float) * (size_t)(total + 1) * head_size);
      float* vd = (float*)malloc(sizeof(float) * (size_t)(total + 1) * head_size);
      int kvh = h / qpk;
      const int* bt = block_tables + (int64_t)s * max_blocks_per_seq;
      for (int j = 0; j < total; ++j)
        for (int d = 0; d < head_size; ++d) {
          if (j < ctx) {
            int64_t base = (int64_t)bt[j / block_size] * kv_block_stride + (int64_t)kvh * kv_head_stride;
            int off = j % block_size;
            kd[(int64_t)j * head_size + d] = load_cache(
                k_cache, base + (int64_t)(d / x) * block_size * x + (int64_t)off * x + d % x, dt, kv, kv_scale);
            vd[(int64_t)j * head_size + d] = load_cache(
                v_cache, base + (int64_t)d * block_size + off, dt, kv, kv_scale);
          } else {
            kd[(int64_t)j * head_size + d] =
                load_scalar(k, (int64_t)(start + j - ctx) * k_stride + (int64_t)kvh * head_size + d, dt);
            vd[(int64_t)j * head_size + d] =
                load_scalar(v, (int64_t)(start + j - ctx) * v_stride + (int64_t)kvh * head_size + d, dt);
          }
        }
      for (int i = 0; i < n; ++i) {
        int qpos = ctx + i;
        int j0 = sliding_window > 0 && qpos - sliding_window + 1 > 0 ? qpos - sliding_window + 1 : 0;
        float mx = -3.402823466e+38f;
        for (int j = j0; j <= qpos; ++j) {
          float dot = 0.f;
          for (int d = 0; d < head_size; ++d)
            dot += load_scalar(q, (int64_t)(start + i) * q_stride + (int64_t)h * head_size + d, dt) *
                   kd[(int64_t)j * head_size + d];
          float xx = scale * dot;
          if (alibi_slopes) xx += alibi_slopes[h] * (float)(j - qpos);
          logits[j] = xx;
          mx = fmaxf(mx, xx);
        }
        float sum = 0.f;
        for (int j = j0; j <= qpos; ++j) { logits[j] = expf(logits[j] - mx); sum += logits[j]; }
        for (int d = 0; d < head_size; ++d) acc[d] = 0.f;
        for (int j = j0; j <= qpos; ++j) {
          float pj = round_scalar(logits[j] / sum, dt);
          for (int d = 0; d < head_size; ++d) acc[d] += pj * vd[(int64_t)j * head_size + d];
        }
        for (int d = 0; d < head_size; ++d)
          store_scalar(out, (int64_t)(start + i) * out_stride + (int64_t)h * head_size + d, dt,
                       acc[d]);
      }
      free(logits); free(acc); free(kd); free(vd);
    }
  }
}

/* ------------------------------------------------------- int4 weight quantisation */

/* AWQ: W[k, 8j + i] = (nib(qweight[k, j], P[i]) - nib(qzeros[k/G, j], P[i])) * scales[k/G, 8j+i]
 * with P = {0,4,1,5,2,6,3,7}: dequantize_s4_to_fp16x2 returns the nibbles of a word in the
 * order n0,n4,n1,n5,n2,n6,n3,n7 (csrc/quantization/awq/dequantize.cuh:41-61); (q - z) is
 * exact in fp16, then one fp16 multiply by the scale
 * (csrc/quantization/awq/gemm_kernels.cu:395-424).  out: [IC, OC] fp16 bits. */
static const int AWQ_ORDER[8] = {0, 4, 1, 5, 2, 6, 3, 7};

void oracle_awq_dequantize(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                           uint16_t* out, int in_c, int out_c, int group) {
  int qoc = out_c / 8;
  for (int k = 0; k < in_c; ++k) {
    int g = k / group;
    for (int j = 0; j < qoc; ++j) {
      uint32_t qw = (uint32_t)qweight[(int64_t)k * qoc + j];
      uint32_t qz = (uint32_t)qzeros[(int64_t)g * qoc + j];
      for (int i = 0; i < 8; ++i) {
        int sh = 4 * AWQ_ORDER[i];
        float w = (float)((int)((qw >> sh) & 15) - (int)((qz >> sh) & 15));
        float s = oracle_f16_to_f32(scales[(int64_t)g * out_c + 8 * j + i]);
        out[(int64_t)k * out_c + 8 * j + i] = oracle_f32_to_f16(w * s);
      }
    }
  }
}

/* GPTQ (4 bit): W[k, n] = (nib(qweight[k/8, n], k%8) - (nib(qzeros[g, n/8], n%8) + 1)) * scales[g, n],
 * g = g_idx[k] or k / G (csrc/quantization/gptq/q_gemm.cu:1386-1420; zero lookup
 * csrc/quantization/gptq/matrix_view.cuh:109-137).  shuffled != 0: qweight words are in the
 * exllama order written by gptq_shuffle (csrc/quantization/gptq/qdq_4.cuh:16-29; reconstruct
 * csrc/quantization/gptq/q_gemm.cu:856-959).  out: [K, N] fp16 bits. */
static inline int exllama_nibble_pos(int i) { return (i & 1) ? 4 + (i >> 1) : (i >> 1); }

void oracle_gptq_dequantize(const int32_t* qweight, const int32_t* qzeros, const uint16_t* scales,
                            const int32_t* g_idx, uint16_t* out, int size_k, int size_n, int groups,
                            int shuffled) {
  int gs = size_k / groups;
  for (int k = 0; k < size_k; ++k) {
    int g = g_idx ? g_idx[k] : k / gs;
    int pos = shuffled ? exllama_nibble_pos(k & 7) : (k & 7);
    for (int n = 0; n < size_n; ++n) {
      uint32_t qw = (uint32_t)qweight[(int64_t)(k >> 3) * size_n + n];
      uint32_t qz = (uint32_t)qzeros[(int64_t)g * (size_n / 8) + (n >> 3)];
      int w = (int)((qw >> (4 * pos)) & 15) - ((int)((qz >> (4 * (n & 7))) & 15) + 1);
      float s = oracle_f16_to_f32(scales[(int64_t)g * size_n + n]);
      out[(int64_t)k * size_n + n] = oracle_f32_to_f16((float)w * s);
    }
  }
}

/* gptq_shuffle for 4 bit (csrc/quantization/gptq/q_gemm.cu:1543-1554, :1595-1624,
 * :1771-1818): optional row gather by q_perm (make_sequential), then per word the
 * nibble permutation of shuffle_4bit_8 (qdq_4.cuh:16-29). In place. */
void oracle_gptq_shuffle(int32_t* qweight, const int32_t* q_perm, int size_k, int size_n) {
  int rows = size_k / 8;
  if (q_perm) {
    uint32_t* nw = (uint32_t*)calloc((size_t)rows * size_n, 4);
    for (int r = 0; r < rows; ++r)
      for (int n = 0; n < size_n; ++n) {
        uint32_t dst = 0;
        for (int i = 0; i < 8; ++i) {
          int src_row = q_perm[r * 8 + i];
          uint32_t w = (uint32_t)qweight[(int64_t)(src_row >> 3) * size_n + n];
          dst |= ((w >> (4 * (src_row & 7))) & 15u) << (4 * i);
        }
        nw[(int64_t)r * size_n + n] = dst;
      }
    memcpy(qweight, nw, (size_t)rows * size_n * 4);
    free(nw);
  }
  for (int64_t i = 0; i < (int64_t)rows * size_n; ++i) {
    uint32_t qa = (uint32_t)qweight[i], qb = 0;
    for (int j = 0; j < 4; ++j) {
      qb |= ((qa >> (8 * j)) & 15u) << (4 * j);
      qb |= ((qa >> (8 * j + 4)) & 15u) << (4 * j + 16);
    }
    qweight[i] = (int32_t)qb;
  }
}

/* C[M, N] = A[M, K] (fp16) x W[K, N] (fp16), fp32 accumulate, rounded once to fp16: the
 * contraction both quantised paths reduce to at M >= 256 (awq.py:164-172 torch.matmul;
 * q_gemm.cu:1496-1518 cublasHgemm).  a_perm (nullable) gathers A's columns first, as the
 * exllama act-order path does (q_gemm.cu:224-227). */
void oracle_gemm_f16(const uint16_t* a, const uint16_t* w, uint16_t* c, int m, int n, int k,
                     const int32_t* a_perm) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; ++i) {
    float* acc = (float*)calloc((size_t)n, 4);
    for (int kk = 0; kk < k; ++kk) {
      float av = oracle_f16_to_f32(a[(int64_t)i * k + (a_perm ? a_perm[kk] : kk)]);
      const uint16_t* wr = w + (int64_t)kk * n;
      for (int j = 0; j < n; ++j) acc[j] += av * oracle_f16_to_f32(wr[j]);
    }
    for (int j = 0; j < n; ++j) c[(int64_t)i * n + j] = oracle_f32_to_f16(acc[j]);
    free(acc);
  }
}

/* ------------------------------------------------------------ layer glue (next to attention) */

/* rms_norm / fused_add_rms_norm (csrc/layernorm_kernels.cu:22-46, :254-289). fused != 0:
 * residual = round(input + residual) first and the result goes to `input`; else out = norm(input). */
void oracle_rms_norm(void* out, void* input, void* residual, const void* weight, float eps,
                     int num_tokens, int hidden, int dt, int fused) {
  for (int64_t t = 0; t < num_tokens; ++t) {
    float var = 0.f;
    for (int i = 0; i < hidden; ++i) {
      float x = load_scalar(input, t * hidden + i, dt);
      if (fused) {
        x = round_scalar(x + load_scalar(residual, t * hidden + i, dt), dt);
        store_scalar(residual, t * hidden + i, dt, x);
      }
      var += x * x;
    }
    float inv = 1.0f / sqrtf(var / (float)hidden + eps);
    for (int i = 0; i < hidden; ++i) {
      float x = load_scalar(fused ? residual : input, t * hidden + i, dt);
      float y = round_scalar(round_scalar(x * inv, dt) * load_scalar(weight, i, dt), dt);
      store_scalar(fused ? input : out, t * hidden + i, dt, y);
    }
  }
}

/* rotary_embedding (csrc/pos_encoding_kernels.cu:10-93): scalar_t arithmetic, i.e. every
 * product and every sum rounded to the model dtype. */
void oracle_rotary_embedding(const int64_t* positions, void* query, void* key,
                             const void* cos_sin_cache, int num_tokens, int num_heads,
                             int num_kv_heads, int head_size, int rot_dim, int64_t q_stride,
                             int64_t k_stride, int is_neox, int dt) {
  int embed = rot_dim / 2;
  for (int64_t t = 0; t < num_tokens; ++t) {
    int64_t cb = positions[t] * rot_dim;
    for (int pass = 0; pass < 2; ++pass) {
      void* arr = pass ? key : query;
      int nh = pass ? num_kv_heads : num_heads;
      int64_t stride = pass ? k_stride : q_stride;
      for (int h = 0; h < nh; ++h)
        for (int o = 0; o < embed; ++o) {
          int64_t base = t * stride + (int64_t)h * head_size;
          int xi = is_neox ? o : 2 * o, yi = is_neox ? embed + o : 2 * o + 1;
          float c = load_scalar(cos_sin_cache, cb + o, dt);
          float s = load_scalar(cos_sin_cache, cb + embed + o, dt);
          float x = load_scalar(arr, base + xi, dt), y = load_scalar(arr, base + yi, dt);
          store_scalar(arr, base + xi, dt, round_scalar(x * c, dt) - round_scalar(y * s, dt));
          store_scalar(arr, base + yi, dt, round_scalar(y * c, dt) + round_scalar(x * s, dt));
        }
    }
  }
}

/* silu_and_mul (csrc/activation_kernels.cu:14-33): out = T(x / (1 + exp(-x))) * y. */
void oracle_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dt) {
  for (int64_t t = 0; t < num_tokens; ++t)
    for (int i = 0; i < d; ++i) {
      float x = load_scalar(input, t * 2 * d + i, dt), y = load_scalar(input, t * 2 * d + d + i, dt);
      float a = round_scalar(x / (1.0f + expf(-x)), dt);
      store_scalar(out, t * d + i, dt, a * y);
    }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
