// extern "C" entry points of libvla_gfx950.so that are not defined next to their kernels:
// status strings, argument validation and dispatch for the two attention ops.
#include "vla_common.h"

#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>

namespace vla {
static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }
// Process-wide knobs, written by their setters and read by every launch on any thread: relaxed
// atomics (a launch sees either the old or the new value, never a torn one).
static std::atomic<int> g_attention_path{0};  // 0 = pick per shape, 1 = always the generic kernel
}  // namespace vla

using namespace vla;

extern "C" const char* vla_status_string(int status) {
  switch (status) {
    case VLA_OK: return "ok";
    case VLA_ERR_HEAD_SIZE: return "Unsupported head size";
    case VLA_ERR_BLOCK_SIZE: return "Unsupported block size";
    case VLA_ERR_DTYPE: return "Unsupported input type of kv cache";
    case VLA_ERR_KV_DTYPE: return "Unsupported data type of kv cache";
    case VLA_ERR_ARG: return "Invalid argument";
    case VLA_ERR_BLOCKSPARSE: return "Invalid blocksparse parameters (block size must be positive)";
    case VLA_ERR_HIP: return "HIP runtime error";
    case VLA_ERR_QUANT_SHAPE: return "Unsupported quantized weight shape";
    case VLA_ERR_SCRATCH:
      return "Problem too large for the static scratch of vla_paged_attention_v2_noalloc "
             "(num_seqs * num_heads * partitions * (head_size * element size + 8) bytes must fit "
             "VLA_PA_V2_SCRATCH_BYTES)";
    case VLA_ERR_NOT_PREPARED:
      return "vla_paged_attention_v2_noalloc: first call on this device is inside a stream capture; "
             "call vla_paged_attention_v2_noalloc_prepare() on the device before capturing";
    default: return "unknown status";
  }
}

extern "C" int vla_last_hip_error(void) { return g_last_hip_error; }
extern "C" int vla_set_attention_path(int mode) {
  if (mode != 0 && mode != 1) return VLA_ERR_ARG;
  g_attention_path.store(mode, std::memory_order_relaxed);
  return VLA_OK;
}
// K / V load policy of the MFMA attention kernel: 0 by size (nontemporal when the launch reads
// more than the 256 MiB of Infinity Cache can hold, i.e. when nothing it reads will be read again
// from cache), 1 always nontemporal, 2 never.
static std::atomic<int> g_kv_load_policy{[] {
  const char* e = getenv("VLA_KV_LOAD_POLICY");
  const int v = e ? atoi(e) : 0;
  return v >= 0 && v <= 2 ? v : 0;
}()};
extern "C" int vla_set_kv_load_policy(int mode) {
  if (mode < 0 || mode > 2) return VLA_ERR_ARG;
  g_kv_load_policy.store(mode, std::memory_order_relaxed);
  return VLA_OK;
}
static int kv_nontemporal(int num_seqs, int max_tokens, int num_kv_heads, int head_size, int kv_dtype) {
  const int policy = g_kv_load_policy.load(std::memory_order_relaxed);
  if (policy != 0) return policy == 1;
  const int64_t bytes = (int64_t)num_seqs * max_tokens * num_kv_heads * head_size * 2 *
                        (kv_dtype == VLA_KV_FP8_E4M3 ? 1 : 2);
  return bytes > (256ll << 20);
}
extern "C" int vla_abi_version(void) { return 1; }

// Which kernel a paged-attention call of this shape runs on, spelled the way a profiler prints the
// instantiation ("pa_mfma_kernel<vla::F16, true, 128, 16, false, true, false>": model dtype, fp8
// cache, head size, block size, partitioned (V2), nontemporal K/V loads, rotary + cache write fused
// in (vla_rope_paged_attention only)), or "pa_generic_kernel" for
// shapes the MFMA kernel does not take.  Assumes 16-byte aligned caches and strides.
extern "C" int vla_paged_attention_kernel_name(char* buf, int buf_len, int num_seqs, int num_kv_heads,
                                               int head_size, int block_size, int max_seq_len,
                                               int dtype, int kv_dtype, int partitioned) {
  if (!buf || buf_len <= 0) return VLA_ERR_ARG;
  const bool mfma_shape =
      g_attention_path.load(std::memory_order_relaxed) == 0 && (dtype == VLA_F16 || dtype == VLA_BF16) &&
      (block_size == 8 || block_size == 16 || block_size == 32) &&
      (head_size == 64 || head_size == 80 || head_size == 96 || head_size == 112 ||
       head_size == 128 || head_size == 192 || head_size == 256);
  if (!mfma_shape) {
    snprintf(buf, buf_len, "pa_generic_kernel");
    return VLA_OK;
  }
  const int nt = kv_nontemporal(num_seqs, max_seq_len, num_kv_heads, head_size, kv_dtype);
  snprintf(buf, buf_len, "pa_mfma_kernel<vla::%s, %s, %d, %d, %s, %s, false>", dtype == VLA_F16 ? "F16" : "BF16",
           kv_dtype == VLA_KV_FP8_E4M3 ? "true" : "false", head_size, block_size,
           partitioned ? "true" : "false", nt ? "true" : "false");
  return VLA_OK;
}

extern "C" int vla_kv_dtype_from_string(const char* name) {
  if (!name) return -1;
  if (!strcmp(name, "auto")) return VLA_KV_AUTO;
  if (!strcmp(name, "fp8") || !strcmp(name, "fp8_e4m3")) return VLA_KV_FP8_E4M3;
  return -1;
}

extern "C" int vla_get_max_shared_memory_per_block(int device) {
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess)
    return -1;
  return v;
}

static int check_attn(const void* out, const void* query, const void* kc, const void* vc,
                      int num_seqs, int num_heads, int num_kv_heads, int head_size,
                      int block_size, const int* block_tables, const int* seq_lens,
                      int max_blocks, int dtype, int kv_dtype, int vert_stride) {
  if (!out || !query || !kc || !vc || !block_tables || !seq_lens) return VLA_ERR_ARG;
  if (num_seqs < 0 || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads ||
      max_blocks <= 0)
    return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16 && dtype != VLA_F32) return VLA_ERR_DTYPE;
  if (kv_dtype != VLA_KV_AUTO && kv_dtype != VLA_KV_FP8_E4M3) return VLA_ERR_KV_DTYPE;
  if (!block_size_supported(block_size)) return VLA_ERR_BLOCK_SIZE;
  if (!head_size_supported(head_size)) return VLA_ERR_HEAD_SIZE;
  (void)vert_stride;
  return VLA_OK;
}

extern "C" int vla_paged_attention_v1_ordered(
    void* out, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int dtype, int kv_dtype, float kv_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size, int blocksparse_head_sliding_step,
    const int* seq_order, void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (blocksparse_vert_stride > 1 && blocksparse_block_size <= 0) return VLA_ERR_BLOCKSPARSE;
  int rc = check_attn(out, query, key_cache, value_cache, num_seqs, num_heads, num_kv_heads,
                      head_size, block_size, block_tables, seq_lens, max_num_blocks_per_seq,
                      dtype, kv_dtype, blocksparse_vert_stride);
  if (rc != VLA_OK) return rc;
  if (num_seqs == 0) return VLA_OK;
  AttnParams p{};
  p.out = out; p.q = query; p.k_cache = key_cache; p.v_cache = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.alibi_slopes = alibi_slopes;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.head_size = head_size; p.block_size = block_size;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq; p.max_num_partitions = 0;
  p.scale = scale; p.kv_scale = kv_scale;
  p.q_stride = q_stride; p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.bs_vert_stride = blocksparse_vert_stride; p.bs_local_blocks = blocksparse_local_blocks;
  p.bs_block_size = blocksparse_block_size; p.bs_head_sliding_step = blocksparse_head_sliding_step;
  p.tp_rank = tp_rank;
  p.seq_order = seq_order;
  p.kv_nontemporal = kv_nontemporal(
      num_seqs, max_seq_len > 0 ? max_seq_len : max_num_blocks_per_seq * block_size, num_kv_heads,
      head_size, kv_dtype);
  hipStream_t s = (hipStream_t)stream;
  // the block-sparse variant (vert_stride > 1, ref attention_kernels.cu:821) runs on the generic
  // kernel only
  const bool generic_only = g_attention_path.load(std::memory_order_relaxed) == 1 || blocksparse_vert_stride > 1;
  rc = generic_only ? -1 : launch_paged_attention_mfma(p, dtype, kv_dtype, s);
  if (rc >= 0) return rc;
  return launch_paged_attention_generic(p, dtype, kv_dtype, s);
}

extern "C" int vla_paged_attention_v1(
    void* out, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int dtype, int kv_dtype, float kv_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size, int blocksparse_head_sliding_step,
    void* stream) {
  return vla_paged_attention_v1_ordered(
      out, query, key_cache, value_cache, num_seqs, num_heads, num_kv_heads, head_size, block_size,
      scale, block_tables, seq_lens, max_num_blocks_per_seq, max_seq_len, alibi_slopes, q_stride,
      kv_block_stride, kv_head_stride, dtype, kv_dtype, kv_scale, tp_rank, blocksparse_local_blocks,
      blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step, nullptr,
      stream);
}

extern "C" int vla_paged_attention_v2_ordered(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    const void* key_cache, const void* value_cache, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len, const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype,
    float kv_scale, int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride,
    int blocksparse_block_size, int blocksparse_head_sliding_step, const int* seq_order,
    void* stream) {
  (void)hipGetLastError();  // a stale error of some earlier HIP call must not fail this one
  if (blocksparse_vert_stride > 1 && blocksparse_block_size <= 0) return VLA_ERR_BLOCKSPARSE;
  int rc = check_attn(out, query, key_cache, value_cache, num_seqs, num_heads, num_kv_heads,
                      head_size, block_size, block_tables, seq_lens, max_num_blocks_per_seq,
                      dtype, kv_dtype, blocksparse_vert_stride);
  if (rc != VLA_OK) return rc;
  if (!exp_sums || !max_logits || !tmp_out || max_seq_len <= 0) return VLA_ERR_ARG;
  if (num_seqs == 0) return VLA_OK;
  const int P = (max_seq_len + VLA_PARTITION_SIZE - 1) / VLA_PARTITION_SIZE;
  AttnParams p{};
  p.out = tmp_out; p.exp_sums = exp_sums; p.max_logits = max_logits;
  p.q = query; p.k_cache = key_cache; p.v_cache = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.alibi_slopes = alibi_slopes;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.head_size = head_size; p.block_size = block_size;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq; p.max_num_partitions = P;
  p.scale = scale; p.kv_scale = kv_scale;
  p.q_stride = q_stride; p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.bs_vert_stride = blocksparse_vert_stride; p.bs_local_blocks = blocksparse_local_blocks;
  p.bs_block_size = blocksparse_block_size; p.bs_head_sliding_step = blocksparse_head_sliding_step;
  p.tp_rank = tp_rank;
  p.seq_order = seq_order;
  p.kv_nontemporal = kv_nontemporal(num_seqs, max_seq_len, num_kv_heads, head_size, kv_dtype);
  hipStream_t s = (hipStream_t)stream;
  const bool generic_only = g_attention_path.load(std::memory_order_relaxed) == 1 || blocksparse_vert_stride > 1;
  rc = generic_only ? -1 : launch_paged_attention_mfma(p, dtype, kv_dtype, s);
  if (rc < 0) rc = launch_paged_attention_generic(p, dtype, kv_dtype, s);
  if (rc != VLA_OK) return rc;
  return launch_paged_attention_v2_reduce(out, exp_sums, max_logits, tmp_out, seq_lens, num_seqs,
                                          num_heads, head_size, P, dtype, s);
}

extern "C" int vla_paged_attention_v2(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    const void* key_cache, const void* value_cache, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len, const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype,
    float kv_scale, int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride,
    int blocksparse_block_size, int blocksparse_head_sliding_step, void* stream) {
  return vla_paged_attention_v2_ordered(
      out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
      num_kv_heads, head_size, block_size, scale, block_tables, seq_lens, max_num_blocks_per_seq,
      max_seq_len, alibi_slopes, q_stride, kv_block_stride, kv_head_stride, dtype, kv_dtype,
      kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
      blocksparse_head_sliding_step, nullptr, stream);
}

// ---- rotary embedding + cache write + paged attention of a decode step in one launch -----------
// What vla_rotary_reshape_and_cache followed by vla_paged_attention_v1 / _v2 computes (same
// arithmetic, same bits in the cache and in `out`), for the shapes paged_attention_rope_fusable
// names; the rotated query and key are not written back to their tensors.  write_kv == 0: the
// step's key and value are in the cache already (vla_rotary_reshape_and_cache with num_heads 0
// puts them there) and only the query's rotation rides in the attention launch.  exp_sums / max_logits /
// tmp_out null: V1.  Returns VLA_ERR_ARG for a shape or layout the fused kernel does not take.
extern "C" int vla_rope_paged_attention_supported(int num_heads, int num_kv_heads, int head_size,
                                                  int rot_dim, int is_neox, int block_size,
                                                  int dtype) {
  return (g_attention_path.load(std::memory_order_relaxed) == 0 && is_neox && rot_dim == head_size &&
          paged_attention_rope_fusable(num_heads, num_kv_heads, head_size, block_size, dtype)) ? 1 : 0;
}

extern "C" int vla_rope_paged_attention(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    const void* key, const void* value, int64_t q_stride, int64_t k_stride, int64_t v_stride,
    const int64_t* positions, const void* cos_sin_cache, int rot_dim, int is_neox, void* key_cache,
    void* value_cache, const int64_t* slot_mapping, const float* slabs, int num_slabs,
    int64_t slab_stride, int64_t slab_row, int q_col, int k_col, int v_col, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len,
    const float* alibi_slopes, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_dtype, float kv_scale, const int* seq_order, int write_kv, void* stream) {
  (void)hipGetLastError();
  int rc = check_attn(out, slabs ? (const void*)slabs : query, key_cache, value_cache, num_seqs,
                      num_heads, num_kv_heads, head_size, block_size, block_tables, seq_lens,
                      max_num_blocks_per_seq, dtype, kv_dtype, 0);
  if (rc != VLA_OK) return rc;
  if (!positions || !cos_sin_cache || !slot_mapping || max_seq_len <= 0) return VLA_ERR_ARG;
  if (!slabs && (!query || (write_kv && (!key || !value)))) return VLA_ERR_ARG;
  if (!vla_rope_paged_attention_supported(num_heads, num_kv_heads, head_size, rot_dim, is_neox,
                                          block_size, dtype))
    return VLA_ERR_ARG;
  const bool v2 = exp_sums || max_logits || tmp_out;
  if (v2 && (!exp_sums || !max_logits || !tmp_out)) return VLA_ERR_ARG;
  if (num_seqs == 0) return VLA_OK;
  const int P = v2 ? (max_seq_len + VLA_PARTITION_SIZE - 1) / VLA_PARTITION_SIZE : 0;
  AttnParams p{};
  p.out = v2 ? tmp_out : out; p.exp_sums = exp_sums; p.max_logits = max_logits;
  p.q = query; p.k_cache = key_cache; p.v_cache = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.alibi_slopes = alibi_slopes;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.head_size = head_size; p.block_size = block_size;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq; p.max_num_partitions = P;
  p.scale = scale; p.kv_scale = kv_scale;
  p.q_stride = q_stride; p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.seq_order = seq_order;
  p.kv_nontemporal = kv_nontemporal(num_seqs, max_seq_len, num_kv_heads, head_size, kv_dtype);
  RopeFuse rf{};
  rf.positions = positions; rf.slots = slot_mapping; rf.cos_sin = cos_sin_cache;
  rf.k = key; rf.v = value; rf.k_stride = k_stride; rf.v_stride = v_stride;
  rf.slabs = slabs; rf.num_slabs = num_slabs; rf.slab_stride = slab_stride; rf.slab_row = slab_row;
  rf.q_col = q_col; rf.k_col = k_col; rf.v_col = v_col;
  rf.write_kv = write_kv ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  rc = launch_paged_attention_mfma(p, dtype, kv_dtype, s, &rf);
  if (rc < 0) return VLA_ERR_ARG;
  if (rc != VLA_OK || !v2) return rc;
  return launch_paged_attention_v2_reduce(out, exp_sums, max_logits, tmp_out, seq_lens, num_seqs,
                                          num_heads, head_size, P, dtype, s);
}

// ---- paged_attention_v2 for callers that cannot allocate ---------------------------------------
// The partition pass and the reduce of V2 exchange tmp_out / exp_sums / max_logits.  Here they
// live in one file-scope device array instead of caller-owned tensors.  Every GPU the library
// runs on holds its own copy of the array at its own address, so the address is looked up once
// per device (with that device current) and only that pointer is ever handed to kernels; the
// array's name means nothing to host code.
static __device__ __align__(256) unsigned char scratch[VLA_PA_V2_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

// Device pointer of `scratch` on the current device.  The first call on a device looks the
// address up (a HIP API call that may load the code object): that must not happen while `stream`
// is being captured into a graph, so an unprepared device under capture is an error, not a lookup.
int scratch_for_current_device(unsigned char** out, hipStream_t stream = nullptr, bool check_capture = false) {
  int dev = -1;
  VLA_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return VLA_ERR_ARG;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    if (check_capture) {
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      VLA_HIP_CHECK(hipStreamIsCapturing(stream, &st));
      if (st != hipStreamCaptureStatusNone) return VLA_ERR_NOT_PREPARED;
    }
    void* ptr = nullptr;
    VLA_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return VLA_ERR_HIP;
    g_scratch_ptr[dev] = ptr;
  }
  *out = static_cast<unsigned char*>(g_scratch_ptr[dev]);
  return VLA_OK;
}

inline int64_t round_up_256(int64_t v) { return (v + 255) / 256 * 256; }
}  // namespace

extern "C" int vla_paged_attention_v2_noalloc_prepare(void) {
  (void)hipGetLastError();
  unsigned char* base = nullptr;
  return scratch_for_current_device(&base);
}

extern "C" int vla_paged_attention_v2_noalloc(
    void* out, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int dtype, int kv_dtype, float kv_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size, int blocksparse_head_sliding_step,
    void* stream) {
  (void)hipGetLastError();
  if (num_seqs < 0 || num_heads <= 0 || head_size <= 0 || max_seq_len <= 0) return VLA_ERR_ARG;
  if (dtype != VLA_F16 && dtype != VLA_BF16 && dtype != VLA_F32) return VLA_ERR_DTYPE;
  const int64_t P = (max_seq_len + VLA_PARTITION_SIZE - 1) / VLA_PARTITION_SIZE;
  const int64_t rows = (int64_t)num_seqs * num_heads * P;
  const int64_t esize = dtype == VLA_F32 ? 4 : 2;
  const int64_t tmp_bytes = round_up_256(rows * head_size * esize);
  const int64_t stat_bytes = round_up_256(rows * 4);
  if (tmp_bytes + 2 * stat_bytes > (int64_t)VLA_PA_V2_SCRATCH_BYTES) return VLA_ERR_SCRATCH;
  unsigned char* base = nullptr;
  int rc = scratch_for_current_device(&base, (hipStream_t)stream, true);
  if (rc != VLA_OK) return rc;
  // Nothing in the scratch has to start at zero: the reduce reads, per sequence, exactly the
  // partitions below ceil(seq_len / 512), and the partition pass of the same call has written
  // every one of them.  What an earlier call left behind is never read.
  void* tmp_out = base;
  float* exp_sums = reinterpret_cast<float*>(base + tmp_bytes);
  float* max_logits = reinterpret_cast<float*>(base + tmp_bytes + stat_bytes);
  return vla_paged_attention_v2_ordered(
      out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
      num_kv_heads, head_size, block_size, scale, block_tables, seq_lens, max_num_blocks_per_seq,
      max_seq_len, alibi_slopes, q_stride, kv_block_stride, kv_head_stride, dtype, kv_dtype,
      kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
      blocksparse_head_sliding_step, nullptr, stream);
}
