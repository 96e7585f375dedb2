// TEST INFRASTRUCTURE ONLY -- binding shim for the *reference's own* CPU kernels.
//
// The reference's csrc/cpu/attention.cpp and csrc/cpu/cache.cpp (and, for the glue kernels,
// layernorm.cpp, pos_encoding.cpp, activation.cpp) define the
// tensor-signature entry points (csrc/cpu/attention.cpp:421-438, :738-756;
// csrc/cpu/cache.cpp:85-129), while its csrc/cpu/pybind.cpp includes headers
// whose signatures were changed to pointer strings (csrc/ops.h:14-30,
// csrc/cache.h:15-25) and therefore no longer binds them.  This file (ours)
// declares the tensor signatures those two translation units actually define
// and exposes them to Python.  The two reference sources are compiled where
// they lie under /root/reference by oracle/build_ref.py; nothing from the
// reference is copied into this repository.
#include <torch/extension.h>

#include <string>
#include <vector>

void paged_attention_v1(
    torch::Tensor& out, torch::Tensor& query, torch::Tensor& key_cache,
    torch::Tensor& value_cache, int num_kv_heads, float scale,
    torch::Tensor& block_tables, torch::Tensor& seq_lens, int block_size,
    int max_seq_len, const c10::optional<torch::Tensor>& alibi_slopes,
    const std::string& kv_cache_dtype, float kv_scale, const int tp_rank,
    const int blocksparse_local_blocks, const int blocksparse_vert_stride,
    const int blocksparse_block_size, const int blocksparse_head_sliding_step);

void paged_attention_v2(
    torch::Tensor& out, torch::Tensor& exp_sums, torch::Tensor& max_logits,
    torch::Tensor& tmp_out, torch::Tensor& query, torch::Tensor& key_cache,
    torch::Tensor& value_cache, int num_kv_heads, float scale,
    torch::Tensor& block_tables, torch::Tensor& seq_lens, int block_size,
    int max_seq_len, const c10::optional<torch::Tensor>& alibi_slopes,
    const std::string& kv_cache_dtype, float kv_scale, const int tp_rank,
    const int blocksparse_local_blocks, const int blocksparse_vert_stride,
    const int blocksparse_block_size, const int blocksparse_head_sliding_step);

void reshape_and_cache(torch::Tensor& key, torch::Tensor& value,
                       torch::Tensor& key_cache, torch::Tensor& value_cache,
                       torch::Tensor& slot_mapping,
                       const std::string& kv_cache_dtype, float kv_scale);

void copy_blocks(std::vector<torch::Tensor>& key_caches,
                 std::vector<torch::Tensor>& value_caches,
                 const torch::Tensor& block_mapping);

// csrc/cpu/layernorm.cpp:90-118, csrc/cpu/pos_encoding.cpp:169-199, csrc/cpu/activation.cpp:80-90:
// the CPU twins of the glue kernels around attention (signatures as in csrc/ops.h:32-48).
void rms_norm(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, float epsilon);
void fused_add_rms_norm(torch::Tensor& input, torch::Tensor& residual, torch::Tensor& weight,
                        float epsilon);
void rotary_embedding(torch::Tensor& positions, torch::Tensor& query, torch::Tensor& key,
                      int head_size, torch::Tensor& cos_sin_cache, bool is_neox);
void silu_and_mul(torch::Tensor& out, torch::Tensor& input);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rms_norm", &rms_norm);
  m.def("fused_add_rms_norm", &fused_add_rms_norm);
  m.def("rotary_embedding", &rotary_embedding);
  m.def("silu_and_mul", &silu_and_mul);
  m.def("paged_attention_v1", &paged_attention_v1);
  m.def("paged_attention_v2", &paged_attention_v2);
  m.def("reshape_and_cache", &reshape_and_cache);
  m.def("copy_blocks", &copy_blocks);
}
