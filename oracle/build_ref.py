"""TEST INFRASTRUCTURE ONLY -- build the reference's own CPU kernels into oracle/_ref/.

Compiles, in place under /root/reference, the translation units that hold
the reference's CPU paged-attention, cache and glue kernels
(csrc/cpu/attention.cpp, cache.cpp, layernorm.cpp, pos_encoding.cpp,
activation.cpp) behind our binding shim
(oracle/ref_shim.cpp), plus a tiny C library over the reference's host-side
FNUZ fp8 codec (oracle/ref_fp8_shim.cpp).  Outputs go only to oracle/_ref/
(git-ignored, but shipped to the GPU box with the tree).

The reference CPU kernels hard-require AVX-512 (cmake/cpu_extension.cmake:36-58);
`ref_usable()` says whether the *running* host can execute them.

Nothing here is on the product path: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may use the artefacts.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("VLLM_REFERENCE_ROOT", "/root/reference")
EXT_NAME = "vllm_ref_cpu"


def host_has_avx512() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
    except OSError:
        return False
    need = ("avx512f", "avx512vl", "avx512bw", "avx512dq")
    return all(n in txt for n in need)


def ref_so_path() -> str:
    return os.path.join(OUT, EXT_NAME + ".so")


def fp8_so_path() -> str:
    return os.path.join(OUT, "libref_fp8.so")


def ref_usable() -> bool:
    return os.path.exists(ref_so_path()) and host_has_avx512()


def build(verbose: bool = False) -> bool:
    """Build both artefacts if the reference tree is present. Returns True when
    oracle/_ref/ holds the extension afterwards."""
    if not os.path.isdir(os.path.join(REF, "csrc", "cpu")):
        return os.path.exists(ref_so_path())
    os.makedirs(OUT, exist_ok=True)

    # (1) FNUZ codec: plain g++ shared object, no torch.
    fp8_so = fp8_so_path()
    shim = os.path.join(HERE, "ref_fp8_shim.cpp")
    if (not os.path.exists(fp8_so)
            or os.path.getmtime(fp8_so) < os.path.getmtime(shim)):
        subprocess.check_call([
            "g++", "-O2", "-std=c++17", "-shared", "-fPIC",
            "-I" + os.path.join(REF, "csrc"), shim, "-o", fp8_so
        ])

    # (2) CPU paged attention + cache kernels: torch C++ extension.
    so = ref_so_path()
    srcs = [
        os.path.join(HERE, "ref_shim.cpp"),
        os.path.join(REF, "csrc", "cpu", "attention.cpp"),
        os.path.join(REF, "csrc", "cpu", "cache.cpp"),
        # CPU twins of the glue kernels (RMSNorm, rotary embedding, SiLU-and-mul)
        os.path.join(REF, "csrc", "cpu", "layernorm.cpp"),
        os.path.join(REF, "csrc", "cpu", "pos_encoding.cpp"),
        os.path.join(REF, "csrc", "cpu", "activation.cpp"),
    ]
    if os.path.exists(so) and all(
            os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return True
    if not host_has_avx512():
        # cross-building AVX-512 code is fine, but we could not test it here
        pass
    from torch.utils import cpp_extension
    cpp_extension.load(
        name=EXT_NAME,
        sources=srcs,
        extra_cflags=[
            "-O2", "-fopenmp", "-mavx512f", "-mavx512vl", "-mavx512bw",
            "-mavx512dq", "-DVLLM_CPU_EXTENSION", "-std=c++17"
        ],
        extra_ldflags=["-fopenmp"],
        build_directory=OUT,
        verbose=verbose,
        is_python_module=False,
    )
    return os.path.exists(so)


def load_ref():
    """Import the built extension (None when unusable on this host)."""
    if not ref_usable():
        return None
    import importlib.util

    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(EXT_NAME, ref_so_path())
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    ok = build(verbose="-v" in sys.argv)
    print("oracle/_ref built:", ok, "usable on this host:", ref_usable())
