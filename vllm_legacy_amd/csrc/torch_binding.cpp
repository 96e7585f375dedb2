// Torch-extension binding of libvla_gfx950.so: the module a maintainer of the reference would build
// in place of `vllm._C` for the decode hot path.  Same sub-modules and function names as
// csrc/pybind.cpp:6-111 (`ops`, `cache_ops`, `cuda_utils`), same positional signatures as
// csrc/ops.h and csrc/cache.h -- including the fork's pointer-string form of the KV cache for
// paged_attention_v2 and reshape_and_cache (csrc/ops.h:14-30, csrc/cache.h:15-25).
//
// Host code only: it unwraps tensors into the pointers, strides and sizes of the C-ABI
// (include/vla_paged_attention.h), takes torch's current HIP stream, and turns a non-zero status
// into the RuntimeError the reference raises through TORCH_CHECK.  No kernels live here.
#include <torch/extension.h>

// PyTorch-ROCm keeps its GPU tensors under the device type "cuda"; these are the guard and stream
// types that go with that (the plain c10::hip ones refuse such a device)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <string>
#include <vector>

#include "vla_paged_attention.h"

namespace {

int dtype_code(const torch::Tensor& t) {
  switch (t.scalar_type()) {
    case at::kHalf: return VLA_F16;
    case at::kBFloat16: return VLA_BF16;
    case at::kFloat: return VLA_F32;
    default: TORCH_CHECK(false, "Unsupported input type of kv cache: ", t.scalar_type());
  }
  return -1;
}

int kv_code(const std::string& name) {
  int c = vla_kv_dtype_from_string(name.c_str());
  TORCH_CHECK(c >= 0, "Unsupported data type of kv cache: ", name);
  return c;
}

void* stream_of(const torch::Tensor& t) {
  return t.is_cuda() ? c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream() : nullptr;
}

// Makes the tensor's GPU current for the scope (the reference's OptionalCUDAGuard,
// csrc/attention/attention_kernels.cu:736).  A tensor that is not on a GPU selects nothing: the
// attention and GEMM bindings then run the library's argument validation alone (zero rows, so
// nothing is launched) -- shape errors are reported as on a GPU -- and refuse afterwards.
struct DeviceScope {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard;
  bool gpu;
  explicit DeviceScope(const torch::Tensor& t) : gpu(t.is_cuda()) {
    if (gpu) guard.set_device(t.device());
  }
  void require(const char* what) const { TORCH_CHECK(gpu, what, ": expected a tensor on the GPU"); }
};

void check(int rc, const char* what) {
  TORCH_CHECK(rc == VLA_OK, what, ": ", vla_status_string(rc),
              rc == VLA_ERR_HIP ? " (hipError_t " + std::to_string(vla_last_hip_error()) + ")" : "");
}

// std::stoull as the fork does it (csrc/attention/attention_kernels.cu:897-898)
void* parse_ptr(const std::string& s) { return reinterpret_cast<void*>(std::stoull(s)); }

const float* alibi_ptr(const c10::optional<torch::Tensor>& a) {
  return a.has_value() ? a->data_ptr<float>() : nullptr;
}

// ---------------------------------------------------------------- ops
void paged_attention_v1(torch::Tensor& out, torch::Tensor& query, torch::Tensor& key_cache,
                        torch::Tensor& value_cache, int num_kv_heads, float scale,
                        torch::Tensor& block_tables, torch::Tensor& seq_lens, int block_size,
                        int max_seq_len, const c10::optional<torch::Tensor>& alibi_slopes,
                        const std::string& kv_cache_dtype, float kv_scale, const int tp_rank,
                        const int blocksparse_local_blocks, const int blocksparse_vert_stride,
                        const int blocksparse_block_size, const int blocksparse_head_sliding_step) {
  const DeviceScope scope(query);
  check(vla_paged_attention_v1(
            out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
            scope.gpu ? (int)query.size(0) : 0, (int)query.size(1), num_kv_heads, (int)query.size(2), block_size,
            scale, block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(),
            (int)block_tables.stride(0), max_seq_len, alibi_ptr(alibi_slopes), query.stride(0),
            key_cache.stride(0), key_cache.stride(1), dtype_code(query), kv_code(kv_cache_dtype),
            kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
            blocksparse_block_size, blocksparse_head_sliding_step, stream_of(query)),
        "paged_attention_v1");
  scope.require("paged_attention_v1");
}

void paged_attention_v2(torch::Tensor& out, torch::Tensor& exp_sums, torch::Tensor& max_logits,
                        torch::Tensor& tmp_out, torch::Tensor& query,
                        const std::string& key_cache_data_ptr_str, int key_cache_stride_0,
                        int key_cache_stride_1, const std::string& value_cache_data_ptr_str,
                        int num_kv_heads, float scale, torch::Tensor& block_tables,
                        torch::Tensor& seq_lens, int block_size, int max_seq_len,
                        const c10::optional<torch::Tensor>& alibi_slopes,
                        const std::string& kv_cache_dtype, float kv_scale, const int tp_rank,
                        const int blocksparse_local_blocks, const int blocksparse_vert_stride,
                        const int blocksparse_block_size, const int blocksparse_head_sliding_step) {
  const DeviceScope scope(query);
  check(vla_paged_attention_v2(
            out.data_ptr(), exp_sums.data_ptr<float>(), max_logits.data_ptr<float>(),
            tmp_out.data_ptr(), query.data_ptr(), parse_ptr(key_cache_data_ptr_str),
            parse_ptr(value_cache_data_ptr_str), scope.gpu ? (int)query.size(0) : 0,
            (int)query.size(1),
            num_kv_heads, (int)query.size(2), block_size, scale, block_tables.data_ptr<int>(),
            seq_lens.data_ptr<int>(), (int)block_tables.stride(0), max_seq_len,
            alibi_ptr(alibi_slopes), query.stride(0), key_cache_stride_0, key_cache_stride_1,
            dtype_code(query), kv_code(kv_cache_dtype), kv_scale, tp_rank,
            blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
            blocksparse_head_sliding_step, stream_of(query)),
        "paged_attention_v2");
  scope.require("paged_attention_v2");
}

void rms_norm(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, float epsilon) {
  const DeviceScope scope(input);
  scope.require("rms_norm");
  const int hidden = (int)input.size(-1);
  check(vla_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), epsilon,
                     (int)(input.numel() / hidden), hidden, dtype_code(input), stream_of(input)),
        "rms_norm");
}

void fused_add_rms_norm(torch::Tensor& input, torch::Tensor& residual, torch::Tensor& weight,
                        float epsilon) {
  const DeviceScope scope(input);
  scope.require("fused_add_rms_norm");
  const int hidden = (int)input.size(-1);
  check(vla_fused_add_rms_norm(input.data_ptr(), residual.data_ptr(), weight.data_ptr(), epsilon,
                               (int)(input.numel() / hidden), hidden, dtype_code(input),
                               stream_of(input)),
        "fused_add_rms_norm");
}

void rotary_embedding(torch::Tensor& positions, torch::Tensor& query, torch::Tensor& key,
                      int head_size, torch::Tensor& cos_sin_cache, bool is_neox) {
  const DeviceScope scope(query);
  scope.require("rotary_embedding");
  const int64_t num_tokens = query.numel() / query.size(-1);      // pos_encoding_kernels.cu:136
  check(vla_rotary_embedding(positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(),
                             cos_sin_cache.data_ptr(), (int)num_tokens,
                             (int)(query.size(-1) / head_size), (int)(key.size(-1) / head_size),
                             head_size, (int)cos_sin_cache.size(1), query.stride(-2),
                             key.stride(-2), is_neox ? 1 : 0, dtype_code(query), stream_of(query)),
        "rotary_embedding");
}

void silu_and_mul(torch::Tensor& out, torch::Tensor& input) {
  const DeviceScope scope(input);
  scope.require("silu_and_mul");
  const int d = (int)(input.size(-1) / 2);
  check(vla_silu_and_mul(out.data_ptr(), input.data_ptr(), (int)(input.numel() / (2 * d)), d,
                         dtype_code(input), stream_of(input)),
        "silu_and_mul");
}

torch::Tensor awq_gemm(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor scaling_factors,
                       torch::Tensor zeros, int split_k_iters) {
  (void)split_k_iters;  // fp32 accumulation in one pass: nothing to split
  const DeviceScope scope(in_feats);
  auto x = in_feats.contiguous();
  const int m = (int)x.size(0), in_c = (int)x.size(1), out_c = (int)kernel.size(1) * 8;
  auto out = torch::empty({m, out_c}, x.options());
  check(vla_awq_gemm(x.data_ptr(), kernel.data_ptr<int32_t>(), scaling_factors.data_ptr(),
                     zeros.data_ptr<int32_t>(), out.data_ptr(), scope.gpu ? m : 0, in_c, out_c,
                     in_c / (int)scaling_factors.size(0), stream_of(x)),
        "awq_gemm");
  scope.require("awq_gemm");
  return out;
}

torch::Tensor awq_dequantize(torch::Tensor kernel, torch::Tensor scaling_factors,
                             torch::Tensor zeros, int split_k_iters, int thx, int thy) {
  (void)split_k_iters; (void)thx; (void)thy;  // launch-shape hints of the CUDA kernel
  const DeviceScope scope(kernel);
  scope.require("awq_dequantize");
  const int in_c = (int)kernel.size(0), out_c = (int)kernel.size(1) * 8;
  auto out = torch::empty({in_c, out_c}, scaling_factors.options());
  check(vla_awq_dequantize(kernel.data_ptr<int32_t>(), scaling_factors.data_ptr(),
                           zeros.data_ptr<int32_t>(), out.data_ptr(), in_c, out_c,
                           in_c / (int)scaling_factors.size(0), stream_of(kernel)),
        "awq_dequantize");
  return out;
}

torch::Tensor gptq_gemm(torch::Tensor a, torch::Tensor b_q_weight, torch::Tensor b_gptq_qzeros,
                        torch::Tensor b_gptq_scales, torch::Tensor b_g_idx, bool use_exllama,
                        int bit) {
  const DeviceScope scope(a);
  auto x = a.contiguous();
  const int m = (int)x.size(0), k = (int)x.size(1), n = (int)b_q_weight.size(1);
  auto out = torch::empty({m, n}, x.options());
  const bool has_idx = b_g_idx.defined() && !b_g_idx.is_meta() && b_g_idx.numel() > 0;
  check(vla_gptq_gemm(x.data_ptr(), b_q_weight.data_ptr<int32_t>(),
                      b_gptq_qzeros.data_ptr<int32_t>(), b_gptq_scales.data_ptr(),
                      has_idx ? b_g_idx.data_ptr<int32_t>() : nullptr, out.data_ptr(),
                      scope.gpu ? m : 0, n, k,
                      (int)b_gptq_qzeros.size(0), use_exllama ? 1 : 0, bit, stream_of(x)),
        "gptq_gemm");
  scope.require("gptq_gemm");
  return out;
}

void gptq_shuffle(torch::Tensor q_weight, torch::Tensor q_perm, int bit) {
  const DeviceScope scope(q_weight);
  scope.require("gptq_shuffle");
  const bool has_perm = q_perm.defined() && !q_perm.is_meta() && q_perm.numel() > 0;
  torch::Tensor scratch;
  if (has_perm) scratch = torch::empty_like(q_weight);   // the reference cudaMalloc's this buffer
  check(vla_gptq_shuffle(q_weight.data_ptr<int32_t>(),
                         has_perm ? q_perm.data_ptr<int32_t>() : nullptr,
                         has_perm ? scratch.data_ptr<int32_t>() : nullptr,
                         (int)q_weight.size(0) * 32 / bit, (int)q_weight.size(1), bit,
                         stream_of(q_weight)),
        "gptq_shuffle");
}

// ---------------------------------------------------------------- cache_ops
void swap_blocks(torch::Tensor& src, torch::Tensor& dst, const torch::Tensor& block_mapping) {
  int direction;
  torch::Device dev = src.device();
  if (src.is_cuda() && dst.is_cuda()) {
    TORCH_CHECK(src.get_device() == dst.get_device(), "src and dst must be on the same GPU");
    direction = VLA_SWAP_D2D;
  } else if (src.is_cuda() && dst.is_cpu()) {
    direction = VLA_SWAP_D2H;
  } else if (src.is_cpu() && dst.is_cuda()) {
    direction = VLA_SWAP_H2D;
    dev = dst.device();
  } else {
    TORCH_CHECK(false, "Invalid device combination");
  }
  TORCH_CHECK(block_mapping.device().is_cpu(), "block_mapping must be on CPU");
  auto bm = block_mapping.to(torch::kInt64).contiguous();
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);
  const int64_t block_bytes = src.element_size() * src[0].numel();
  check(vla_swap_blocks(src.data_ptr(), dst.data_ptr(), bm.data_ptr<int64_t>(), (int)bm.size(0),
                        block_bytes, direction,
                        c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream()),
        "swap_blocks");
}

void copy_blocks(std::vector<torch::Tensor>& key_caches, std::vector<torch::Tensor>& value_caches,
                 const torch::Tensor& block_mapping) {
  const int num_layers = (int)key_caches.size();
  TORCH_CHECK(num_layers == (int)value_caches.size());
  if (num_layers == 0) return;
  TORCH_CHECK(key_caches[0].is_cuda() && block_mapping.is_cuda());
  const DeviceScope scope(key_caches[0]);
  std::vector<void*> kp(num_layers), vp(num_layers);
  for (int i = 0; i < num_layers; ++i) {
    kp[i] = key_caches[i].data_ptr();
    vp[i] = value_caches[i].data_ptr();
  }
  auto bm = block_mapping.contiguous();
  const int64_t block_bytes = key_caches[0].element_size() * key_caches[0][0].numel();
  check(vla_copy_blocks(kp.data(), vp.data(), num_layers, bm.data_ptr<int64_t>(), (int)bm.size(0),
                        block_bytes, stream_of(key_caches[0])),
        "copy_blocks");
}

void reshape_and_cache(torch::Tensor& key, torch::Tensor& value,
                       const std::string& key_cache_data_ptr_str, int key_cache_block_size,
                       int key_cache_x, const std::string& value_cache_data_ptr_str,
                       torch::Tensor& slot_mapping, const std::string& kv_cache_dtype,
                       const float kv_scale) {
  const DeviceScope scope(key);
  scope.require("reshape_and_cache");
  check(vla_reshape_and_cache(key.data_ptr(), value.data_ptr(), parse_ptr(key_cache_data_ptr_str),
                              parse_ptr(value_cache_data_ptr_str), slot_mapping.data_ptr<int64_t>(),
                              (int)key.size(0), (int)key.size(1), (int)key.size(2),
                              key_cache_block_size, key_cache_x, key.stride(0), value.stride(0),
                              dtype_code(key), kv_code(kv_cache_dtype), kv_scale, stream_of(key)),
        "reshape_and_cache");
}

void convert_fp8(torch::Tensor& dst_cache, torch::Tensor& src_cache, const float scale,
                 const std::string& kv_cache_dtype) {
  TORCH_CHECK(src_cache.is_cuda() && dst_cache.is_cuda(), "src and dst must be on a GPU");
  TORCH_CHECK(src_cache.get_device() == dst_cache.get_device(),
              "src and dst must be on the same GPU");
  const DeviceScope scope(src_cache);
  const bool to_fp8 = dst_cache.scalar_type() == at::kByte;
  TORCH_CHECK(to_fp8 != (src_cache.scalar_type() == at::kByte),
              "convert_fp8 needs exactly one uint8 side");
  check(vla_convert_fp8(dst_cache.data_ptr(), src_cache.data_ptr(), src_cache.numel(),
                        dtype_code(to_fp8 ? src_cache : dst_cache), kv_code(kv_cache_dtype), scale,
                        to_fp8 ? 1 : 0, stream_of(src_cache)),
        "convert_fp8");
}

int get_max_shared_memory_per_block_device_attribute(int device_id) {
  return vla_get_max_shared_memory_per_block(device_id);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  pybind11::module ops = m.def_submodule("ops", "gfx950 operators (the hot-path subset of vllm._C.ops)");
  ops.def("paged_attention_v1", &paged_attention_v1, "PagedAttention V1 (tensor caches).");
  ops.def("paged_attention_v2", &paged_attention_v2, "PagedAttention V2 (pointer-string caches).");
  ops.def("silu_and_mul", &silu_and_mul, "Activation function used in SwiGLU.");
  ops.def("rms_norm", &rms_norm, "Root mean square normalization.");
  ops.def("fused_add_rms_norm", &fused_add_rms_norm, "In-place fused add and RMS normalization.");
  ops.def("rotary_embedding", &rotary_embedding, "Rotary embedding on query and key, in place.");
  ops.def("awq_gemm", &awq_gemm, "Quantized GEMM for AWQ");
  ops.def("awq_dequantize", &awq_dequantize, "Dequantization for AWQ");
  ops.def("gptq_gemm", &gptq_gemm, "Quantized GEMM for GPTQ");
  ops.def("gptq_shuffle", &gptq_shuffle, "Post processing for GPTQ");

  pybind11::module cache_ops = m.def_submodule("cache_ops", "gfx950 cache ops");
  cache_ops.def("swap_blocks", &swap_blocks, "Swap blocks between device and host (or on device)");
  cache_ops.def("copy_blocks", &copy_blocks, "Copy cache blocks from src to dst");
  cache_ops.def("reshape_and_cache", &reshape_and_cache, "Write key / value into the paged cache");
  cache_ops.def("convert_fp8", &convert_fp8, "Convert a cache between the model dtype and fp8");

  pybind11::module cuda_utils = m.def_submodule("cuda_utils", "device utilities");
  cuda_utils.def("get_max_shared_memory_per_block_device_attribute",
                 &get_max_shared_memory_per_block_device_attribute,
                 "Maximum LDS bytes per workgroup of a device");
}
