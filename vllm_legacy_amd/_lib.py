"""ctypes binding of libvla_gfx950.so (the C-ABI declared in include/vla_paged_attention.h).

There is no CPU fallback: if the library is missing, importing an op raises, and a non-zero
status from the library becomes a RuntimeError carrying the message the reference would have
raised through TORCH_CHECK.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# VLA_LIB_PATH: another build of the same library (kernel studies: A/B in one process tree)
LIB_PATH = os.environ.get("VLA_LIB_PATH") or os.path.join(HERE, "libvla_gfx950.so")
CSRC = os.path.join(HERE, "csrc")

F16, BF16, F32 = 0, 1, 2
KV_AUTO, KV_FP8_E4M3 = 0, 1
PARTITION_SIZE = 512
PA_V2_SCRATCH_BYTES = 32 << 20
SWAP_D2D, SWAP_D2H, SWAP_H2D = 0, 1, 2

_lib = None

EXPORTS = [
    "vla_status_string", "vla_last_hip_error", "vla_abi_version", "vla_set_attention_path",
    "vla_set_kv_load_policy", "vla_set_q4_row_block_kernel", "vla_set_prefill_form",
    "vla_kv_dtype_from_string", "vla_paged_attention_v1", "vla_paged_attention_v2",
    "vla_paged_attention_v1_ordered", "vla_paged_attention_v2_ordered",
    "vla_paged_attention_v2_noalloc", "vla_paged_attention_v2_noalloc_prepare",
    "vla_paged_attention_kernel_name",
    "vla_reshape_and_cache", "vla_copy_blocks", "vla_swap_blocks", "vla_convert_fp8",
    "vla_awq_dequantize", "vla_awq_gemm", "vla_gptq_dequantize", "vla_gptq_gemm",
    "vla_gptq_shuffle", "vla_get_max_shared_memory_per_block", "vla_awq_repack",
    "vla_q4_gemm_workspace_bytes", "vla_q4_gemm_packed", "vla_rms_norm",
    "vla_fused_add_rms_norm", "vla_rotary_embedding", "vla_silu_and_mul",
    "vla_prefill_attention", "vla_prefix_prefill_attention", "vla_ar_region_bytes", "vla_ar_alloc_region",
    "vla_ar_free_region", "vla_ar_export", "vla_ar_import", "vla_ar_unimport", "vla_ar_create", "vla_ar_destroy",
    "vla_ar_all_reduce", "vla_ar_all_reduce_slabs", "vla_ar_all_reduce_add_rms_norm", "vla_ar_error_count", "vla_ar_error_count_async", "vla_q4_gemm_grouped", "vla_q4_gemm_grouped_workspace_bytes",
    "vla_q4_gemm_grouped_gather", "vla_moe_route_topk", "vla_moe_gate_topk", "vla_moe_sort_pairs", "vla_moe_combine", "vla_moe_gather_perm",
    "vla_q4_gemm_packed_partial", "vla_q4_gemm_silu_and_mul_supported", "vla_q4_gemm_packed_silu_and_mul", "vla_rotary_reshape_and_cache", "vla_rope_paged_attention_supported", "vla_rope_paged_attention", "vla_fused_add_rms_norm_splitk", "vla_silu_and_mul_splitk",
]


def build(jobs: int = 8) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", CSRC])
    return LIB_PATH


TORCH_BINDING_NAME = "vla_torch_C"
TORCH_BINDING_PATH = os.path.join(HERE, TORCH_BINDING_NAME + ".so")
_torch_binding = None


def build_torch_binding(verbose: bool = False) -> str:
    """Compile csrc/torch_binding.cpp -- the pybind11 / torch-extension face of the C-ABI, the
    module a maintainer would build in place of `vllm._C` -- with torch.utils.cpp_extension.
    Host compiler only (no device code); links libvla_gfx950.so next to it."""
    import shutil
    import torch  # noqa: F401
    from torch.utils import cpp_extension
    src = os.path.join(CSRC, "torch_binding.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "vla_paged_attention.h")]
    if os.path.exists(TORCH_BINDING_PATH) and all(
            os.path.getmtime(TORCH_BINDING_PATH) >= os.path.getmtime(d) for d in deps):
        return TORCH_BINDING_PATH
    build_dir = os.path.join(HERE, "build", "torch_binding")
    os.makedirs(build_dir, exist_ok=True)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cpp_extension.load(
        name=TORCH_BINDING_NAME, sources=[src], build_directory=build_dir, verbose=verbose,
        extra_include_paths=[os.path.join(HERE, "..", "include"), "/opt/rocm/include"],
        extra_cflags=["-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
        extra_ldflags=[f"-L{HERE}", "-lvla_gfx950", f"-L{torch_lib}", "-lc10_hip", "-ltorch_hip",
                       # $$ survives ninja, the quotes survive the shell: the library sits next to this
                       # module (and two levels up from the build directory)
                       "'-Wl,-rpath,$$ORIGIN'", "'-Wl,-rpath,$$ORIGIN/../..'"],
        with_cuda=False, is_python_module=False)
    shutil.copy2(os.path.join(build_dir, TORCH_BINDING_NAME + ".so"), TORCH_BINDING_PATH)
    return TORCH_BINDING_PATH


def torch_binding():
    """The compiled torch-extension module (sub-modules ops / cache_ops / cuda_utils, as
    csrc/pybind.cpp:6-111).  Built by build_torch_binding(); never built here."""
    global _torch_binding
    if _torch_binding is None:
        if not os.path.exists(TORCH_BINDING_PATH):
            raise ImportError(f"{TORCH_BINDING_PATH} is missing: build it with "
                              "vllm_legacy_amd._lib.build_torch_binding() (__graft_entry__.build())")
        import importlib.util
        lib()       # libvla_gfx950.so (and torch) first: the module links against both
        spec = importlib.util.spec_from_file_location(TORCH_BINDING_NAME, TORCH_BINDING_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _torch_binding = mod
    return _torch_binding


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or "
                "__graft_entry__.build()). This backend has no fallback path.")
        # torch first: it brings its own HIP runtime, and the library must bind to the runtime
        # that owns the device memory and streams it is handed (loading in the other order
        # leaves launches failing with hipErrorNoDevice)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.vla_status_string.restype = C.c_char_p
        L.vla_status_string.argtypes = [C.c_int]
        L.vla_kv_dtype_from_string.argtypes = [C.c_char_p]
        i, i64, f, p = C.c_int, C.c_int64, C.c_float, C.c_void_p
        L.vla_paged_attention_v1.argtypes = [p, p, p, p, i, i, i, i, i, f, p, p, i, i, p, i64,
                                             i64, i64, i, i, f, i, i, i, i, i, p]
        L.vla_paged_attention_v2.argtypes = [p, p, p, p, p, p, p, i, i, i, i, i, f, p, p, i, i,
                                             p, i64, i64, i64, i, i, f, i, i, i, i, i, p]
        L.vla_paged_attention_v1_ordered.argtypes = [p, p, p, p, i, i, i, i, i, f, p, p, i, i, p,
                                                     i64, i64, i64, i, i, f, i, i, i, i, i, p, p]
        L.vla_paged_attention_v2_ordered.argtypes = [p, p, p, p, p, p, p, i, i, i, i, i, f, p, p,
                                                     i, i, p, i64, i64, i64, i, i, f, i, i, i, i,
                                                     i, p, p]
        L.vla_paged_attention_v2_noalloc.argtypes = L.vla_paged_attention_v1.argtypes
        L.vla_paged_attention_v2_noalloc_prepare.argtypes = []
        L.vla_paged_attention_kernel_name.argtypes = [C.c_char_p, i, i, i, i, i, i, i, i, i]
        L.vla_reshape_and_cache.argtypes = [p, p, p, p, p, i, i, i, i, i, i64, i64, i, i, f, p]
        L.vla_copy_blocks.argtypes = [p, p, i, p, i, i64, p]
        L.vla_swap_blocks.argtypes = [p, p, p, i, i64, i, p]
        L.vla_convert_fp8.argtypes = [p, p, i64, i, i, f, i, p]
        L.vla_awq_dequantize.argtypes = [p, p, p, p, i, i, i, p]
        L.vla_awq_gemm.argtypes = [p, p, p, p, p, i, i, i, i, p]
        L.vla_gptq_dequantize.argtypes = [p, p, p, p, p, i, i, i, i, p]
        L.vla_gptq_gemm.argtypes = [p, p, p, p, p, p, i, i, i, i, i, i, p]
        L.vla_gptq_shuffle.argtypes = [p, p, p, i, i, i, p]
        L.vla_get_max_shared_memory_per_block.argtypes = [i]
        L.vla_awq_repack.argtypes = [p, p, p, p, i, i, i, p]
        L.vla_q4_gemm_workspace_bytes.argtypes = [i, i, i, i]
        L.vla_q4_gemm_workspace_bytes.restype = i64
        L.vla_q4_gemm_packed.argtypes = [p, p, p, p, p, i, i, i, i, i, p, i64, p]
        L.vla_rms_norm.argtypes = [p, p, p, f, i, i, i, p]
        L.vla_fused_add_rms_norm.argtypes = [p, p, p, f, i, i, i, p]
        L.vla_rotary_embedding.argtypes = [p, p, p, p, i, i, i, i, i, i64, i64, i, i, p]
        L.vla_silu_and_mul.argtypes = [p, p, i, i, i, p]
        L.vla_prefill_attention.argtypes = [p, p, p, p, p, i, i, i, i, i, i64, i64, i64, i64, f, p,
                                            i, p]
        L.vla_prefix_prefill_attention.argtypes = [p, p, p, p, p, p, p, i, p, p, p, i, i, i, i, i, i,
                                                   i64, i64, i64, i64, i64, i64, f, p, i, i, i, f,
                                                   p]
        L.vla_q4_gemm_grouped_workspace_bytes.argtypes = [i, i, i, i, i]
        L.vla_q4_gemm_grouped_workspace_bytes.restype = i64
        L.vla_q4_gemm_grouped.argtypes = [p, p, p, p, p, p, i, i, i, i, i, i, i, p, i64, p]
        L.vla_q4_gemm_grouped_gather.argtypes = [p, p, p, p, p, p, p, i, i, i, i, i, i, i, p, i64, p]
        L.vla_moe_route_topk.argtypes = [p, i, i, i, p, p, i, p]
        L.vla_moe_gate_topk.argtypes = [p, p, i, i, i, i, p, p, p, i, p]
        L.vla_moe_sort_pairs.argtypes = [p, p, i, i, i, i, p, p, p, p, p]
        L.vla_moe_combine.argtypes = [p, p, p, p, p, i, i, i, i, i, p, i, p]
        L.vla_moe_gather_perm.argtypes = [p, p, p, p, p, i, i, i64, p]
        L.vla_ar_region_bytes.argtypes = [i64, p]
        L.vla_ar_alloc_region.argtypes = [i64, p]
        L.vla_ar_free_region.argtypes = [p]
        L.vla_ar_export.argtypes = [p, p]
        L.vla_ar_import.argtypes = [p, i, p]
        L.vla_ar_unimport.argtypes = [p]
        L.vla_ar_create.argtypes = [p, i, i, p, i64, i]
        L.vla_ar_destroy.argtypes = [p]
        L.vla_ar_all_reduce.argtypes = [p, p, p, i64, i, p]
        L.vla_ar_all_reduce_slabs.argtypes = [p, p, p, i, i64, i64, i, p]
        L.vla_ar_all_reduce_add_rms_norm.argtypes = [p, p, p, i, i64, p, p, f, i, i, i, p]
        L.vla_ar_error_count.argtypes = [p, p]
        L.vla_ar_error_count_async.argtypes = [p, p, p]
        L.vla_q4_gemm_packed_partial.argtypes = [p, p, p, p, p, i, i, i, i, i, p, i64, p, p]
        L.vla_q4_gemm_silu_and_mul_supported.argtypes = [i, i, i, i]
        L.vla_q4_gemm_packed_silu_and_mul.argtypes = [p, p, p, p, p, i, i, i, i, i, p]
        L.vla_rotary_reshape_and_cache.argtypes = [p, p, p, p, p, p, p, p, i, i, i, i, i, i64, i64,
                                                   i64, i, i, i, i, i, f, p, i, i64, i64, i, i, i,
                                                   p]
        L.vla_rope_paged_attention_supported.argtypes = [i, i, i, i, i, i, i]
        L.vla_rope_paged_attention.argtypes = [p, p, p, p, p, p, p, i64, i64, i64, p, p, i, i, p, p,
                                               p, p, i, i64, i64, i, i, i, i, i, i, i, i, f, p, p,
                                               i, i, p, i64, i64, i, i, f, p, i, p]
        L.vla_fused_add_rms_norm_splitk.argtypes = [p, i, i64, p, p, p, f, i, i, i, p]
        L.vla_silu_and_mul_splitk.argtypes = [p, i, i64, p, i, i, i, p]
        _lib = L
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().vla_status_string(status).decode()
        if status == 7:
            msg += f" (hipError_t {lib().vla_last_hip_error()})"
        raise RuntimeError(f"{what}: {msg}" if what else msg)


def kv_dtype_code(name: str) -> int:
    code = lib().vla_kv_dtype_from_string(name.encode())
    if code < 0:
        raise RuntimeError(f"Unsupported data type of kv cache: {name}")
    return code
