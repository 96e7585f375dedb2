"""CPU tests of the boundary: the C-ABI library loads without a GPU, exports every symbol that
include/vla_paged_attention.h declares, and reports errors the way the reference does."""
import os
import re

import pytest

from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    from vllm_legacy_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vla_paged_attention.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vla_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)
    assert L.vla_abi_version() == 1


def test_status_strings_and_kv_dtype_names():
    from vllm_legacy_amd import _lib
    L = _lib.lib()
    assert L.vla_status_string(0) == b"ok"
    assert b"head size" in L.vla_status_string(1)
    assert b"block size" in L.vla_status_string(2)
    assert _lib.kv_dtype_code("auto") == 0
    assert _lib.kv_dtype_code("fp8") == 1 and _lib.kv_dtype_code("fp8_e4m3") == 1
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        _lib.kv_dtype_code("fp8_e5m2")


def test_argument_validation_without_gpu():
    # validation happens before any HIP call, so it can be exercised on a CPU-only host
    from vllm_legacy_amd import _lib
    L = _lib.lib()
    one = 0x1000  # never dereferenced: validation fails first
    rc = L.vla_paged_attention_v1(one, one, one, one, 1, 8, 8, 72, 16, 1.0, one, one, 4, 64, None,
                                  576, 9216, 1152, 0, 0, 1.0, 0, 0, 0, 64, 0, None)
    assert rc == 1  # head size
    rc = L.vla_paged_attention_v1(one, one, one, one, 1, 8, 8, 64, 4, 1.0, one, one, 4, 64, None,
                                  512, 8192, 1024, 0, 0, 1.0, 0, 0, 0, 64, 0, None)
    assert rc == 2  # block size
    rc = L.vla_paged_attention_v1(one, one, one, one, 1, 8, 3, 64, 16, 1.0, one, one, 4, 64, None,
                                  512, 8192, 1024, 0, 0, 1.0, 0, 0, 0, 64, 0, None)
    assert rc == 5  # heads not divisible
    rc = L.vla_paged_attention_v1(one, one, one, one, 1, 8, 8, 64, 16, 1.0, one, one, 4, 64, None,
                                  512, 8192, 1024, 0, 0, 1.0, 0, 0, 8, 0, 0, None)
    assert rc == 6  # blocksparse with a non-positive sparse block size
    assert L.vla_awq_gemm(one, one, one, one, one, 4, 128, 32, 128, None) == 8
    assert L.vla_gptq_gemm(one, one, one, one, None, one, 4, 64, 128, 1, 1, 8, None) == 8
    assert L.vla_swap_blocks(one, one, one, 1, 64, 9, None) == 5


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vllm_legacy_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no fallback"):
        _lib.lib()


def test_v1_v2_selection_rules():
    """Host logic of forward_decode: the reference's rule when only query heads are known, the
    workgroup-count rule of the gfx950 kernels (no 8192-token cap) when KV heads are given."""
    from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention as P
    # reference rule (paged_attn.py:120-121)
    assert P.use_v1(512, 1, 32) and P.use_v1(4096, 256, 32)
    assert not P.use_v1(4096, 16, 32) and not P.use_v1(8193, 256, 32)
    # gfx950 rule
    assert P.use_v1(512, 1, 32, 8)                 # a single partition: nothing to split
    assert not P.use_v1(4096, 8, 32, 8) and P.use_v1(4096, 16, 32, 8)
    assert P.use_v1(16384, 32, 32, 8) and not P.use_v1(16384, 8, 32, 8)
    assert not P.use_v1(4096, 64, 8, 1) and P.use_v1(4096, 128, 8, 1)   # one KV head per rank
