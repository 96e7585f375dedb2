/*
 * C-ABI of the MI355X (gfx950) quantized paged-attention backend.
 *
 * One shared library (libvla_gfx950.so) exports these entry points.  They are what a
 * binding for the reference's `vllm._C` ops on this path would call: plain pointers, sizes,
 * strides and a HIP stream -- no torch types.  Every function enqueues its work on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream) and returns immediately; none
 * of them allocates, frees or synchronises, so all of them may be captured in a HIP graph
 * (vla_swap_blocks with a host side is the exception, as in the reference).
 *
 * Return value: VLA_OK (0) or a VLA_ERR_* code; vla_status_string() gives the text the
 * reference would have put in its TORCH_CHECK message.  A non-zero status means nothing was
 * launched.
 *
 * Citations are paths in the reference tree (DeepAuto-AI/vllm-legacy).
 */
#ifndef VLA_PAGED_ATTENTION_H_
#define VLA_PAGED_ATTENTION_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* model ("scalar") dtypes */
#define VLA_F16 0
#define VLA_BF16 1
#define VLA_F32 2

/* KV-cache dtypes: "auto" = the model dtype; "fp8" / "fp8_e4m3" = OCP e4m3fn bytes with one
 * fp32 scale per call (quantise: fp8(x / kv_scale), dequantise: fp8 * kv_scale). */
#define VLA_KV_AUTO 0
#define VLA_KV_FP8_E4M3 1

#define VLA_OK 0
#define VLA_ERR_HEAD_SIZE 1    /* "Unsupported head size"   attention_kernels.cu:763-765 */
#define VLA_ERR_BLOCK_SIZE 2   /* "Unsupported block size"  attention_kernels.cu:800-802 */
#define VLA_ERR_DTYPE 3        /* "Unsupported input type of kv cache" amd/quant_utils.cuh:553 */
#define VLA_ERR_KV_DTYPE 4     /* "Unsupported data type of kv cache"  amd/quant_utils.cuh:569 */
#define VLA_ERR_ARG 5          /* inconsistent sizes / null pointers */
#define VLA_ERR_BLOCKSPARSE 6  /* blocksparse_vert_stride > 1 with blocksparse_block_size <= 0 */
#define VLA_ERR_HIP 7          /* a HIP runtime call failed (see vla_last_hip_error) */
#define VLA_ERR_QUANT_SHAPE 8  /* AWQ/GPTQ shape constraint, gemm_kernels.cu:515-522 */
#define VLA_ERR_SCRATCH 9      /* vla_paged_attention_v2_noalloc: problem exceeds its scratch */
#define VLA_ERR_NOT_PREPARED 10 /* vla_paged_attention_v2_noalloc: first call on a device came
                                   under stream capture without ..._noalloc_prepare() before it */

#define VLA_PARTITION_SIZE 512 /* attention_kernels.cu:846, vllm/attention/ops/paged_attn.py:10 */

const char* vla_status_string(int status);
int vla_last_hip_error(void);            /* hipError_t of the last VLA_ERR_HIP on this thread */
int vla_abi_version(void);
/* Test hook, no counterpart in the reference: 0 = choose the attention kernel per shape
 * (default), 1 = always run the generic kernel.  Process-wide (all threads, all devices), kept in
 * a relaxed atomic: a launch racing with the setter sees the old or the new value. */
int vla_set_attention_path(int mode);
/* No counterpart in the reference.  Cache policy of the K / V loads of the MFMA attention
 * kernel: 0 = by size (default: nontemporal loads when the launch reads more than the 256 MiB the
 * Infinity Cache holds, num_seqs x max_seq_len tokens of K and V), 1 = always nontemporal, 2 =
 * never.  Results do not depend on it.  Process-wide (all threads, all devices), kept in a
 * relaxed atomic; the environment variable VLA_KV_LOAD_POLICY sets the initial value. */
int vla_set_kv_load_policy(int mode);
/* No counterpart in the reference.  Where the int4 GEMM kernel for row blocks of 17..80 rows
 * (16-row MFMA blocks, x shared through LDS; csrc/q4_gemm_mid.hip) runs: 0 = grouped launches
 * (vla_q4_gemm_grouped*: the experts of an MoE block) only (default), 1 = dense products of 17..80
 * rows too (tests, kernel studies: the tile kernels are faster there), 2 = nowhere.  Results agree
 * to the GEMM tolerance whichever kernel runs.  Process-wide, relaxed atomic; the environment
 * variable VLA_Q4_MID sets the initial value. */
int vla_set_q4_row_block_kernel(int mode);
/* No counterpart in the reference; test hook and kernel studies.  Which form of the prompt
 * attention kernel vla_prefill_attention / vla_prefix_prefill_attention launch: 0 = by the longest
 * prompt of the call (default), 1 = 256 threads, one 128-query tile of one head per workgroup,
 * 2 = 512 threads, two heads of a KV group (or two query tiles) per workgroup sharing each staged
 * K/V tile.  Results agree to the attention tolerance whichever runs.  Process-wide, relaxed
 * atomic; the environment variable VLA_PF_FORM sets the initial value. */
int vla_set_prefill_form(int mode);
/* No counterpart in the reference; for measurement tooling.  Writes into buf the name of the
 * kernel instantiation a paged-attention call of this shape launches, spelled as a profiler's
 * kernel trace prints it (e.g. "pa_mfma_kernel<vla::F16, true, 128, 16, false, true, false>": model
 * dtype, fp8 cache, head size, block size, partitioned, nontemporal K/V loads, rotary + cache
 * write fused in (true only from vla_rope_paged_attention); or
 * "pa_generic_kernel"), so that a stored profile can be matched to the kernel a run launches.
 * Assumes 16-byte aligned cache pointers and strides.  partitioned: 0 = V1, 1 = V2. */
int vla_paged_attention_kernel_name(char* buf, int buf_len, int num_seqs, int num_kv_heads,
                                    int head_size, int block_size, int max_seq_len, int dtype,
                                    int kv_dtype, int partitioned);
/* "auto" -> VLA_KV_AUTO, "fp8" / "fp8_e4m3" -> VLA_KV_FP8_E4M3, anything else -> -1
 * (DISPATCH_BY_KV_CACHE_DTYPE, csrc/quantization/fp8/amd/quant_utils.cuh:545-571). */
int vla_kv_dtype_from_string(const char* name);

/* Replaces paged_attention_v1 (csrc/ops.h:5-12; csrc/attention/attention_kernels.cu:805-825).
 *   out, query   [num_seqs, num_heads, head_size] model dtype; query row stride q_stride
 *                (elements), out contiguous
 *   key_cache    [num_blocks, num_kv_heads, head_size/x, block_size, x], x = 16 / sizeof(cache
 *                element); value_cache [num_blocks, num_kv_heads, head_size, block_size]
 *   kv_block_stride, kv_head_stride: elements, shared by both caches (:706-707)
 *   block_tables [num_seqs, max_num_blocks_per_seq] int32; seq_lens [num_seqs] int32
 *   alibi_slopes [num_heads] fp32 or NULL
 * head_size in {64,80,96,112,128,192,256}, block_size in {8,16,32}. */
int vla_paged_attention_v1(void* out, const void* query, const void* key_cache,
                           const void* value_cache, int num_seqs, int num_heads,
                           int num_kv_heads, int head_size, int block_size, float scale,
                           const int* block_tables, const int* seq_lens,
                           int max_num_blocks_per_seq, int max_seq_len,
                           const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                           int64_t kv_head_stride, int dtype, int kv_dtype, float kv_scale,
                           int tp_rank, int blocksparse_local_blocks,
                           int blocksparse_vert_stride, int blocksparse_block_size,
                           int blocksparse_head_sliding_step, void* stream);

/* Replaces paged_attention_v2 (csrc/ops.h:14-30; attention_kernels.cu:984-1014) including its
 * reduce kernel (:567-669).  Caller-owned scratch, P = ceil(max_seq_len / 512):
 *   exp_sums, max_logits [num_seqs, num_heads, P] fp32; tmp_out [num_seqs, num_heads, P,
 *   head_size] model dtype.  The fork passes the caches as decimal pointer strings plus
 *   stride(0)/stride(1) (csrc/ops.h:21-24); here they are the pointers and strides themselves. */
int vla_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                           const void* query, const void* key_cache, const void* value_cache,
                           int num_seqs, int num_heads, int num_kv_heads, int head_size,
                           int block_size, float scale, const int* block_tables,
                           const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len,
                           const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                           int64_t kv_head_stride, int dtype, int kv_dtype, float kv_scale,
                           int tp_rank, int blocksparse_local_blocks,
                           int blocksparse_vert_stride, int blocksparse_block_size,
                           int blocksparse_head_sliding_step, void* stream);

/* The two calls above with one more argument; no counterpart in the reference.  seq_order (int32
 * [num_seqs] on the device, a permutation of 0..num_seqs-1, or NULL) says which sequence each row
 * of the launch grid works on.  Workgroups are handed out in grid order, so with the sequences
 * sorted by length, longest first, a ragged batch finishes without a tail of long sequences that
 * started last (5.3 -> 5.7 TB/s on lengths U(1, 4096)).  Results do not depend on it. */
int vla_paged_attention_v1_ordered(void* out, const void* query, const void* key_cache,
                                   const void* value_cache, int num_seqs, int num_heads,
                                   int num_kv_heads, int head_size, int block_size, float scale,
                                   const int* block_tables, const int* seq_lens,
                                   int max_num_blocks_per_seq, int max_seq_len,
                                   const float* alibi_slopes, int64_t q_stride,
                                   int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
                                   int kv_dtype, float kv_scale, int tp_rank,
                                   int blocksparse_local_blocks, int blocksparse_vert_stride,
                                   int blocksparse_block_size, int blocksparse_head_sliding_step,
                                   const int* seq_order, void* stream);
int vla_paged_attention_v2_ordered(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                                   const void* query, const void* key_cache,
                                   const void* value_cache, int num_seqs, int num_heads,
                                   int num_kv_heads, int head_size, int block_size, float scale,
                                   const int* block_tables, const int* seq_lens,
                                   int max_num_blocks_per_seq, int max_seq_len,
                                   const float* alibi_slopes, int64_t q_stride,
                                   int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
                                   int kv_dtype, float kv_scale, int tp_rank,
                                   int blocksparse_local_blocks, int blocksparse_vert_stride,
                                   int blocksparse_block_size, int blocksparse_head_sliding_step,
                                   const int* seq_order, void* stream);

/* No counterpart in the reference.  vla_paged_attention_v2 for callers that cannot allocate the
 * scratch tensors it needs (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc): exp_sums, max_logits and tmp_out live in a static device array of
 * VLA_PA_V2_SCRATCH_BYTES inside the library (held by every GPU the library runs on until the
 * process ends) instead of being passed in; all other arguments and the results are those of
 * vla_paged_attention_v2, bit for bit.  The call allocates and frees nothing.
 *   Limit: round256(num_seqs * num_heads * P * head_size * sizeof(model dtype)) +
 *   2 * round256(num_seqs * num_heads * P * 4) <= VLA_PA_V2_SCRATCH_BYTES, P = ceil(max_seq_len /
 *   512); larger problems return VLA_ERR_SCRATCH and launch nothing.
 *   All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two
 *   streams, two threads) race; calls on one stream, or ordered by events, are safe.  The first
 *   call on a device looks up the array's address on that device (no allocation; a HIP API call).
 *   That lookup must not happen inside a stream capture: call
 *   vla_paged_attention_v2_noalloc_prepare() once per device (with that device current) before
 *   capturing, or make one eager call first; a first call that arrives under capture returns
 *   VLA_ERR_NOT_PREPARED and launches nothing. */
#define VLA_PA_V2_SCRATCH_BYTES (32 << 20)
int vla_paged_attention_v2_noalloc(void* out, const void* query, const void* key_cache,
                                   const void* value_cache, int num_seqs, int num_heads,
                                   int num_kv_heads, int head_size, int block_size, float scale,
                                   const int* block_tables, const int* seq_lens,
                                   int max_num_blocks_per_seq, int max_seq_len,
                                   const float* alibi_slopes, int64_t q_stride,
                                   int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
                                   int kv_dtype, float kv_scale, int tp_rank,
                                   int blocksparse_local_blocks, int blocksparse_vert_stride,
                                   int blocksparse_block_size, int blocksparse_head_sliding_step,
                                   void* stream);
/* Looks up the scratch array's address on the current device (idempotent, no allocation). */
int vla_paged_attention_v2_noalloc_prepare(void);

/* Replaces reshape_and_cache (csrc/cache.h:15-25; csrc/cache_kernels.cu:251-287).
 *   key, value [num_tokens, num_heads, head_size] model dtype, token strides in elements;
 *   slot_mapping [num_tokens] int64, negative = padding (skipped). */
int vla_reshape_and_cache(const void* key, const void* value, void* key_cache,
                          void* value_cache, const int64_t* slot_mapping, int num_tokens,
                          int num_heads, int head_size, int block_size, int x,
                          int64_t key_stride, int64_t value_stride, int dtype, int kv_dtype,
                          float kv_scale, void* stream);

/* Replaces copy_blocks (csrc/cache.h:11-13; csrc/cache_kernels.cu:99-146).
 *   key_caches, value_caches: HOST arrays of num_layers device pointers;
 *   block_mapping: DEVICE int64 [num_pairs, 2] (src, dst); block_bytes = bytes of one block of
 *   one cache.  The pointer tables travel as kernel arguments: no allocation, no host sync
 *   (the reference does a synchronous H2D copy, :124-131). */
int vla_copy_blocks(void* const* key_caches, void* const* value_caches, int num_layers,
                    const int64_t* block_mapping, int num_pairs, int64_t block_bytes,
                    void* stream);

/* Replaces swap_blocks (csrc/cache.h:8-9; csrc/cache_kernels.cu:25-64).
 *   block_mapping: HOST int64 [num_pairs, 2]; direction: 0 = device->device (same GPU),
 *   1 = device->host, 2 = host->device.  Runs of consecutive (src, dst) pairs are merged into
 *   one hipMemcpyAsync each. */
#define VLA_SWAP_D2D 0
#define VLA_SWAP_D2H 1
#define VLA_SWAP_H2D 2
int vla_swap_blocks(const void* src, void* dst, const int64_t* block_mapping, int num_pairs,
                    int64_t block_bytes, int direction, void* stream);

/* Replaces convert_fp8 (csrc/cache.h:41-42; csrc/cache_kernels.cu:370-420): n elements,
 * to_fp8 != 0: model dtype -> fp8(x / scale); else fp8 -> model dtype (fp8 * scale). */
int vla_convert_fp8(void* dst, const void* src, int64_t n, int dtype, int kv_dtype,
                    float scale, int to_fp8, void* stream);

/* Replaces awq_dequantize (csrc/ops.h:73-76; csrc/quantization/awq/gemm_kernels.cu:436-484).
 *   qweight [in_c, out_c/8] int32, scales [in_c/group, out_c] fp16, qzeros [in_c/group,
 *   out_c/8] int32 -> out [in_c, out_c] fp16. */
int vla_awq_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros,
                       void* out, int in_c, int out_c, int group, void* stream);

/* Replaces awq_gemm (csrc/ops.h:69-71; gemm_kernels.cu:492-549): out[M, out_c] = x[M, in_c] .
 * dequant(qweight), fp16 in/out, fp32 accumulate, one rounding (no split-K partial sums). */
int vla_awq_gemm(const void* x, const int32_t* qweight, const void* scales,
                 const int32_t* qzeros, void* out, int m, int in_c, int out_c, int group,
                 void* stream);

/* Replaces the dequantisation inside gptq_gemm (csrc/quantization/gptq/q_gemm.cu:856-959,
 * :1386-1420), 4 bit: qweight [K/8, N] int32, qzeros [groups, N/8] int32, scales [groups, N]
 * fp16, g_idx [K] int32 or NULL (group = k / (K/groups)); shuffled != 0: qweight is in the
 * order written by vla_gptq_shuffle.  out [K, N] fp16. */
int vla_gptq_dequantize(const int32_t* qweight, const int32_t* qzeros, const void* scales,
                        const int32_t* g_idx, void* out, int size_k, int size_n, int groups,
                        int shuffled, void* stream);

/* Replaces gptq_gemm (csrc/ops.h:111-114; q_gemm.cu:1823-1846), 4 bit only.
 *   use_exllama != 0: qweight was shuffled and b_g_idx (nullable) is the column permutation
 *   applied to `a` (q_gemm.cu:224-227); use_exllama == 0: plain GPTQ layout, b_g_idx maps
 *   k -> group (q_gemm.cu:1410-1412). */
int vla_gptq_gemm(const void* a, const int32_t* qweight, const int32_t* qzeros,
                  const void* scales, const int32_t* b_g_idx, void* out, int m, int size_n,
                  int size_k, int groups, int use_exllama, int bit, void* stream);

/* No counterpart in the reference (its AWQ kernels are CUDA/PTX only,
 * csrc/quantization/awq/gemm_kernels.cu:204-346): one-time re-layout of an AWQ weight into the
 * k-packed pair order that gptq_shuffle produces, so that one MFMA kernel serves both formats.
 *   qweight [in_c, out_c/8], qzeros [in_c/group, out_c/8] (AWQ nibble order) ->
 *   out_qweight [in_c/8, out_c], out_qzeros [in_c/group, out_c/8] (natural nibble order). */
int vla_awq_repack(const int32_t* qweight, const int32_t* qzeros, int32_t* out_qweight,
                   int32_t* out_qzeros, int in_c, int out_c, int group, void* stream);

/* The dequant-GEMM behind awq_gemm / gptq_gemm for k-packed weights (gptq_shuffle'd GPTQ with
 * zero_offset 1, vla_awq_repack'ed AWQ with zero_offset 0): out[m, n] fp16 = x[m, k] fp16 . W.
 * workspace (nullable): device scratch of vla_q4_gemm_workspace_bytes() bytes for split-K
 * partial sums; without it the contraction is not split. k % 64 == 0, group % 32 == 0. */
int64_t vla_q4_gemm_workspace_bytes(int m, int n, int k, int group);
int vla_q4_gemm_packed(const void* x, const int32_t* qweight, const int32_t* qzeros,
                       const void* scales, void* out, int m, int n, int k, int group,
                       int zero_offset, void* workspace, int64_t workspace_bytes, void* stream);

/* Replaces gptq_shuffle (csrc/ops.h:116; q_gemm.cu:1848-1856), 4 bit, in place.
 *   q_perm [K] int32 or NULL; scratch: device buffer of K/8 * N int32, needed only when q_perm
 *   is given (the reference cudaMalloc's it, q_gemm.cu:1774-1775). */
int vla_gptq_shuffle(int32_t* qweight, const int32_t* q_perm, int32_t* scratch, int size_k,
                     int size_n, int bit, void* stream);

/* ---- the elementwise steps next to attention in a decoder layer ----------------------------- */

/* Replaces rms_norm (csrc/ops.h:32-33; csrc/layernorm_kernels.cu:22-46, :291-310):
 * out[t, :] = model_dtype(x * rsqrt(mean(x^2) + eps)) * weight; rows of hidden_size elements,
 * 16-byte aligned. */
int vla_rms_norm(void* out, const void* input, const void* weight, float epsilon, int num_tokens,
                 int hidden_size, int dtype, void* stream);

/* Replaces fused_add_rms_norm (csrc/ops.h:35-36; csrc/layernorm_kernels.cu:254-289), in place:
 * residual = input + residual; input = rms_norm(residual) * weight. */
int vla_fused_add_rms_norm(void* input, void* residual, const void* weight, float epsilon,
                           int num_tokens, int hidden_size, int dtype, void* stream);

/* Replaces rotary_embedding (csrc/ops.h:38-40; csrc/pos_encoding_kernels.cu:10-93, :130-165),
 * in place on query [num_tokens, num_heads, head_size] and key [num_tokens, num_kv_heads,
 * head_size] (token strides in elements); positions int64 [num_tokens]; cos_sin_cache
 * [max_position, rot_dim] model dtype (cos half then sin half). */
int vla_rotary_embedding(const int64_t* positions, void* query, void* key,
                         const void* cos_sin_cache, int num_tokens, int num_heads,
                         int num_kv_heads, int head_size, int rot_dim, int64_t query_stride,
                         int64_t key_stride, int is_neox, int dtype, void* stream);

/* Replaces silu_and_mul (csrc/ops.h:48; csrc/activation_kernels.cu:14-33, :62-77):
 * out[t, :d] = silu(input[t, :d]) * input[t, d:2d]. */
int vla_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dtype,
                     void* stream);

/* ---- prompt attention ----------------------------------------------------------------------- */

/* Causal self-attention over packed prompt tokens without a cached prefix: what the reference's
 * ROCm backend computes for prefill with Triton / CK flash attention or its naive masked path
 * (vllm/attention/backends/rocm_flash_attn.py:305-383).  query/out [T, num_heads, head_size],
 * key/value [T, num_kv_heads, head_size], row strides in elements; cu_seqlens [num_seqs + 1]
 * int32 on the device (the reference's seq_start_loc); alibi_slopes [num_heads] f32 or NULL
 * (bias slope * (j - i)).  fp16 / bf16, head_size 24/64/80/96/112/128; other shapes:
 * VLA_ERR_HEAD_SIZE / VLA_ERR_DTYPE (the caller keeps its previous path for those). */
int vla_prefill_attention(void* out, const void* query, const void* key, const void* value,
                          const int* cu_seqlens, int num_seqs, int max_seqlen, int num_heads,
                          int num_kv_heads, int head_size, int64_t query_stride,
                          int64_t key_stride, int64_t value_stride, int64_t out_stride,
                          float scale, const float* alibi_slopes, int dtype, void* stream);

/* Replaces context_attention_fwd (vllm/attention/ops/prefix_prefill.py:684-830, the Triton
 * kernels :10-230 and :443-680) as called by PagedAttention.forward_prefix
 * (vllm/attention/ops/paged_attn.py:185-216): the new tokens of sequence s (rows
 * query_start_loc[s] .. of query/key/value/out) attend the context_lens[s] tokens already in the
 * paged cache (block_tables[s], layouts as for paged_attention) and, causally, each other;
 * seq_lens[s] = context_lens[s] + number of new tokens.  All index arrays int32 on the device.
 * alibi_slopes as above with absolute positions; sliding_window > 0 keeps keys less than that
 * many positions behind the query (0 / negative: off).  kv_cache_dtype/kv_scale as for
 * paged_attention: a byte cache is decoded with the codec that wrote it.  fp16 / bf16,
 * head_size 64/80/96/112/128 (and 24 with a 16-bit cache, the smallest size of
 * tests/kernels/test_prefix_prefill.py), block_size 16/32/64 (and 8 for a 16-bit cache). */
int vla_prefix_prefill_attention(void* out, const void* query, const void* key, const void* value,
                                 const void* key_cache, const void* value_cache,
                                 const int* block_tables, int max_blocks_per_seq,
                                 const int* query_start_loc, const int* seq_lens,
                                 const int* context_lens, int num_seqs, int max_query_len,
                                 int num_heads, int num_kv_heads, int head_size, int block_size,
                                 int64_t query_stride, int64_t key_stride, int64_t value_stride,
                                 int64_t out_stride, int64_t kv_block_stride,
                                 int64_t kv_head_stride, float scale, const float* alibi_slopes,
                                 int sliding_window, int dtype, int kv_cache_dtype, float kv_scale,
                                 void* stream);

/* Several int4 products in one launch, for the experts of the reference's quantized Mixtral block
 * (vllm/model_executor/models/mixtral_quant.py:129-156, which runs one awq/gptq GEMM per expert
 * over all tokens): out[r, :] = x[r, :] . W[e] for the rows r in row_offsets[e] .. row_offsets[e+1]
 * (int32 [num_sets + 1] on the device, so the launch needs no host knowledge of the routing and
 * can be captured).  qweight / qzeros / scales hold the k-packed tensors of all sets back to back
 * ([num_sets, k/8, n], [num_sets, k/group, n/8], [num_sets, k/group, n]).  Rows outside every set
 * come back zero (the caller hands in a zeroed `out`; the K-split form clears its slabs).
 * max_rows_per_set bounds the grid;
 * group a multiple of 64.  workspace: vla_q4_gemm_grouped_workspace_bytes or NULL. */
int64_t vla_q4_gemm_grouped_workspace_bytes(int total_rows, int n, int k, int group, int num_sets);
int vla_q4_gemm_grouped(const void* x, const int32_t* qweight, const int32_t* qzeros,
                        const void* scales, void* out, const int* row_offsets, int num_sets,
                        int total_rows, int max_rows_per_set, int n, int k, int group,
                        int zero_offset, void* workspace, int64_t workspace_bytes, void* stream);
/* vla_q4_gemm_grouped with the gather of the routed tokens folded in: product row r reads
 * x[x_rows[r], :] (x_rows int32 [total_rows] on the device, e.g. sorted_token of
 * vla_moe_sort_pairs), so the expert loop's `hidden_states[tok]` copy
 * (vllm/model_executor/models/mixtral_quant.py:141-151 runs each expert over all tokens; the
 * routed form gathers first) is never materialised; x_rows NULL reads rows in place.  Rows
 * outside every set are left unwritten (no slab is cleared). */
int vla_q4_gemm_grouped_gather(const void* x, const int* x_rows, const int32_t* qweight,
                               const int32_t* qzeros, const void* scales, void* out,
                               const int* row_offsets, int num_sets, int total_rows,
                               int max_rows_per_set, int n, int k, int group, int zero_offset,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* Routing glue of the quantized Mixtral block (vllm/model_executor/models/mixtral_quant.py:129-156)
 * around the two grouped products, on the device and without host synchronisation.
 * vla_moe_route_topk: router_logits [num_tokens, num_experts] (the gate's output, :130-131) ->
 * routing_weights fp32 [num_tokens, top_k] = softmax in fp32, the top_k largest (on equal values
 * the lower expert index), divided by their sum (:132-138), and selected_experts int64 of the
 * same shape, choices in descending weight.  num_experts <= 64.
 * vla_moe_sort_pairs: selected_experts int64 [num_tokens * top_k] (torch.topk's indices, :136-137);
 * bucket_of int32 [num_experts] maps an expert to its bucket: 0 .. num_buckets - 2 for the experts
 * this rank holds (:109-123), num_buckets - 1 for all others.  Writes row_offsets
 * [num_buckets + 1] (rows of bucket b are row_offsets[b] .. row_offsets[b + 1], pairs of a bucket
 * in pair order), sorted_token [pairs] (the token a sorted row belongs to), pair_row [pairs] (the
 * sorted row of pair p = token * top_k + choice) and, unless NULL, row_bucket [pairs].
 * VLA_ERR_ARG when pairs / 64 * num_buckets exceeds 15360 (the caller keeps a slower path).
 * vla_moe_combine: out[t, :] = sum over the local experts e chosen by token t, ascending e, of
 * model_dtype(y[pair_row[(t, e)], :] * routing_weights[(t, e)]), every partial sum rounded to the
 * model dtype: `expert_layer(hidden_states).mul_(expert_weights)` and `final.add_(...)` of
 * :141-151, where the experts a token did not choose add exact zeros.  y [pairs, hidden] fp16 /
 * bf16, routing_weights fp32 [num_tokens * top_k], top_k <= 8, hidden a multiple of 8. */
/* vla_moe_gate_topk: the gate's projection and vla_moe_route_topk in one launch (the reference
 * runs `self.gate(hidden_states)` as a library GEMM, mixtral_quant.py:130-131; at decode sizes
 * that GEMM and the separate top-k launch cost more than everything else in the routing glue).
 * x [num_tokens, hidden], gate [hidden, num_experts] (x @ gate), fp16 / bf16, num_experts 8 or 16,
 * hidden a multiple of 8; logits = model_dtype(fp32 sums); logits_out (nullable) receives them. */
int vla_moe_gate_topk(const void* x, const void* gate, int num_tokens, int hidden, int num_experts,
                      int top_k, float* routing_weights, int64_t* selected_experts,
                      void* logits_out, int dtype, void* stream);
int vla_moe_route_topk(const void* router_logits, int num_tokens, int num_experts, int top_k,
                       float* routing_weights, int64_t* selected_experts, int dtype, void* stream);
int vla_moe_sort_pairs(const int64_t* selected_experts, const int* bucket_of, int num_experts,
                       int num_tokens, int top_k, int num_buckets, int* row_offsets,
                       int* sorted_token, int* pair_row, int* row_bucket, void* stream);
int vla_moe_combine(const void* y, const float* routing_weights, const int64_t* selected_experts,
                    const int* pair_row, const int* bucket_of, int num_experts, int num_local,
                    int num_tokens, int top_k, int hidden, void* out, int dtype, void* stream);
/* Act-ordered (desc_act) GPTQ experts, whose rows were sorted by group when they were shuffled
 * (vllm/model_executor/layers/quantization/gptq.py:209-217): every sorted row's columns in its own
 * expert's order, out[r, c] = x[src_row[r], perm[row_set[r], c]] (src_row NULL: row r itself),
 * 16-bit elements, x rows x_stride elements apart, perm int32 [num_sets, cols], cols even.  One
 * launch instead of an index tensor of [rows, cols] and two gathers. */
int vla_moe_gather_perm(const void* x, const int* src_row, const int* row_set, const int* perm,
                        void* out, int rows, int cols, int64_t x_stride, void* stream);

/* ---- all-reduce over xGMI peer memory ---------------------------------------------------------- */

/* Together these take the place of the reference's custom all-reduce (csrc/custom_all_reduce.cuh,
 * bound in csrc/pybind.cpp:95-110: init_custom_ar, all_reduce_reg/unreg, dispose, meta_size,
 * register_buffer, get_graph_buffer_ipc_meta, register_graph_buffers), which is CUDA-only.
 * One region per rank (uncached device memory: barrier slots + two staging halves of max_bytes)
 * is exported with vla_ar_export (a 64-byte HIP IPC handle the caller ships to the other
 * processes) and mapped there with vla_ar_import (peer_device: the exporter's device ordinal as
 * this process numbers devices, for an explicit peer-access check, or -1 if unknown).  vla_ar_create takes the `world` region
 * pointers as this process sees them (regions[rank] is its own).  The all-reduce stages the input
 * itself, so no tensor has to be registered, and it is safe to capture in a graph.  All ranks
 * must issue the same calls in the same order, one stream per rank; the other ranks' regions must
 * have been zeroed (vla_ar_alloc_region returns after that) before the first call. */
int vla_ar_region_bytes(int64_t max_bytes, int64_t* out);
int vla_ar_alloc_region(int64_t max_bytes, void** region);
int vla_ar_free_region(void* region);
int vla_ar_export(void* region, unsigned char* handle64);
int vla_ar_import(const unsigned char* handle64, int peer_device, void** peer_region);
int vla_ar_unimport(void* peer_region);
int vla_ar_create(void** ctx, int rank, int world, void* const* regions, int64_t max_bytes,
                  int num_blocks);
int vla_ar_destroy(void* ctx);
/* out = sum over ranks of in (out == in allowed); numel * element size a multiple of 16 and at
 * most max_bytes; fp32 accumulation in rank order, so every rank receives the same bits
 * (custom_all_reduce.cuh:297-356 is the two-shot scheme this follows). */
int vla_ar_all_reduce(void* ctx, void* out, const void* in, int64_t numel, int dtype,
                      void* stream);
/* vla_ar_all_reduce whose input still is the unreduced output of an int4 GEMM that split K:
 * num_slabs fp32 slabs of numel elements, slab_stride elements apart.  Each rank's slabs are
 * summed in slab order and rounded to `dtype` (VLA_F16 / VLA_BF16) while they are published --
 * the bits vla_q4_gemm_packed's own reduction produces -- so out equals vla_ar_all_reduce of the
 * finished products, one launch earlier.  No counterpart in the reference (its GEMMs return
 * finished matrices). */
int vla_ar_all_reduce_slabs(void* ctx, void* out, const float* slabs, int num_slabs,
                            int64_t slab_stride, int64_t numel, int dtype, void* stream);
/* The all-reduce of a row-parallel linear's output followed by what the layer does next with it,
 * residual = sum + residual; out = RMSNorm(residual) * weight (fused_add_rms_norm, csrc/ops.h:55-59,
 * as called from vllm/model_executor/models/llama.py:213-224), in one launch.
 *   in: [rows, hidden] of `dtype`, or with num_slabs > 0 that many fp32 slabs, slab_stride
 *   elements apart (see vla_ar_all_reduce_slabs); residual [rows, hidden] in/out; weight [hidden];
 *   out [rows, hidden].  hidden / 8 must be a multiple of the world size and at most 1024.
 * residual receives the bits of the separate calls; out may differ from them in the last place
 * (the sum of squares is accumulated in another order).  No counterpart in the reference. */
int vla_ar_all_reduce_add_rms_norm(void* ctx, void* out, const void* in, int num_slabs,
                                   int64_t slab_stride, void* residual, const void* weight,
                                   float epsilon, int rows, int hidden, int dtype, void* stream);
/* A barrier wait that saw no peer within its time bound (10 s of wall clock; VLA_XGMI_WAIT_SECONDS
 * at vla_ar_create overrides it) gives up so that the kernel terminates.  The event is counted,
 * the count is sticky, and from the first one on every call on that rank stores NaN in its output
 * (and, for the fused form, in residual): a lost barrier cannot pass for a sum.  The reference's
 * custom all-reduce waits without a bound instead (csrc/custom_all_reduce.cuh:180-205).
 * vla_ar_error_count: synchronous read of the count.  vla_ar_error_count_async: enqueues a 4-byte
 * copy of it to pinned host memory on `stream`, valid after the caller's next synchronisation --
 * one per step at a point where the caller synchronises anyway costs nothing. */
int vla_ar_error_count(void* ctx, int* out);
int vla_ar_error_count_async(void* ctx, int* host_out, void* stream);

/* ---- fused forms (same results as the separate calls, fewer launches) ----------------------- */

/* No counterpart in the reference (its MLP runs the projection and SiluAndMul as two ops,
 * vllm/model_executor/models/llama.py:76-79, csrc/activation_kernels.cu:14-33).  For decode batches
 * of at most 16 rows and a [gate | up] weight wide enough to run unsliced: out [m, n / 2] =
 * silu(gate) * up computed in the product's epilogue, one launch instead of two, the bits of
 * vla_q4_gemm_packed followed by vla_silu_and_mul.  vla_q4_gemm_silu_and_mul_supported says
 * whether a shape is taken (1 / 0); other shapes return VLA_ERR_QUANT_SHAPE and launch nothing. */
int vla_q4_gemm_silu_and_mul_supported(int m, int n, int k, int group);
int vla_q4_gemm_packed_silu_and_mul(const void* x, const int32_t* qweight, const int32_t* qzeros,
                                    const void* scales, void* out, int m, int n, int k, int group,
                                    int zero_offset, void* stream);

/* vla_q4_gemm_packed without its reduction launch: when the product was split along K,
 * *num_slabs (> 1) fp32 slabs [num_slabs][m][n] are left in `workspace` for a consumer below to
 * sum; *num_slabs == 0 means the result went straight to `out`.  The reference has no such
 * split: its GEMMs (csrc/quantization/awq/gemm_kernels.cu:492-549, gptq/q_gemm.cu:1490-1541)
 * always return the finished matrix. */
int vla_q4_gemm_packed_partial(const void* x, const int32_t* qweight, const int32_t* qzeros,
                               const void* scales, void* out, int m, int n, int k, int group,
                               int zero_offset, void* workspace, int64_t workspace_bytes,
                               int* num_slabs, void* stream);

/* rotary_embedding (csrc/ops.h:38-40) on query/key in place, then reshape_and_cache
 * (csrc/cache.h:15-25) of the rotated key and of value, in one launch -- the sequence every
 * decoder layer runs between its qkv projection and attention
 * (vllm/model_executor/models/llama.py:160-166, vllm/attention/ops/paged_attn.py:62-79).
 * With `slabs` != NULL the input is read from the fp32 split-K slabs of the qkv projection
 * ([num_slabs][num_tokens][slab_row], q/k/v starting at columns q_col/k_col/v_col), summed in
 * slab order and rounded to the model dtype; query/key/value then receive the (rotated) result. */
int vla_rotary_reshape_and_cache(const int64_t* positions, void* query, void* key, void* value,
                                 const void* cos_sin_cache, void* key_cache, void* value_cache,
                                 const int64_t* slot_mapping, int num_tokens, int num_heads,
                                 int num_kv_heads, int head_size, int rot_dim, int64_t query_stride,
                                 int64_t key_stride, int64_t value_stride, int block_size, int x,
                                 int is_neox, int dtype, int kv_dtype, float kv_scale,
                                 const float* slabs, int num_slabs, int64_t slab_stride,
                                 int64_t slab_row, int q_col, int k_col, int v_col, void* stream);

/* The same step carried into the attention launch of a decode batch: rotary_embedding
 * (csrc/ops.h:38-40, NeoX pairing over the whole head), reshape_and_cache (csrc/cache.h:15-25) of
 * the rotated key and of value, then paged_attention_v1 (csrc/ops.h:5-12; exp_sums, max_logits and
 * tmp_out NULL) or paged_attention_v2 (csrc/ops.h:14-30) over the cache including the new token,
 * in one launch (two for V2: its reduce).  Token t of the step is sequence t.  The cache bytes and
 * `out` equal those of vla_rotary_reshape_and_cache followed by vla_paged_attention_v1 / _v2 bit
 * for bit; the rotated query and key are NOT written back to query / key.  query / key / value:
 * the unrotated rows (model dtype, row strides in elements), or NULL with `slabs` as above.
 * vla_rope_paged_attention_supported says whether a shape is taken (head size 128, rot_dim ==
 * head size, NeoX, at most 16 query heads per KV head, block size 8 / 16 / 32, fp16 / bf16);
 * anything else returns VLA_ERR_ARG and the caller runs the separate entry points.
 * write_kv == 0: key and value of the step are in the cache already (vla_rotary_reshape_and_cache
 * with num_heads 0 rotates and caches them alone) and only the query is rotated here -- the form
 * for launches that saturate HBM, where stores scattered into the read stream cost more than the
 * launch they save (batch 256 x 4096 tokens: 375 us with the write inside, 345 + 6 without). */
int vla_rope_paged_attention_supported(int num_heads, int num_kv_heads, int head_size, int rot_dim,
                                       int is_neox, int block_size, int dtype);
int vla_rope_paged_attention(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    const void* key, const void* value, int64_t q_stride, int64_t k_stride, int64_t v_stride,
    const int64_t* positions, const void* cos_sin_cache, int rot_dim, int is_neox, void* key_cache,
    void* value_cache, const int64_t* slot_mapping, const float* slabs, int num_slabs,
    int64_t slab_stride, int64_t slab_row, int q_col, int k_col, int v_col, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len,
    const float* alibi_slopes, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_dtype, float kv_scale, const int* seq_order, int write_kv, void* stream);

/* fused_add_rms_norm (csrc/ops.h:35-36) whose `input` arrives as split-K slabs
 * [num_slabs][num_tokens][hidden_size] fp32: x = model_dtype(sum of slabs);
 * residual = x + residual; out = rms_norm(residual) * weight. */
int vla_fused_add_rms_norm_splitk(const float* slabs, int num_slabs, int64_t slab_stride, void* out,
                                  void* residual, const void* weight, float epsilon,
                                  int num_tokens, int hidden_size, int dtype, void* stream);

/* silu_and_mul (csrc/ops.h:62-64) whose input arrives as split-K slabs
 * [num_slabs][num_tokens][2 d] fp32: x = model_dtype(sum of slabs); out[t, j] = silu(x[t, j]) *
 * x[t, d + j].  Same bits as reducing first. */
int vla_silu_and_mul_splitk(const float* slabs, int num_slabs, int64_t slab_stride, void* out,
                            int num_tokens, int d, int dtype, void* stream);

/* Replaces get_max_shared_memory_per_block_device_attribute
 * (csrc/cuda_utils_kernels.cu:17-29). Returns bytes, or -1. */
int vla_get_max_shared_memory_per_block(int device);

#ifdef __cplusplus
}
#endif
#endif /* VLA_PAGED_ATTENTION_H_ */
