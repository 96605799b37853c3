"""CPU: the C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/rcdm.h declares
(no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rcdm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rcdm_[a-z0-9_]+)\s*\(", text)))


def test_build_and_exports():
    import __graft_entry__
    lib_path = __graft_entry__.build()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rcdm.h but not exported"
    from rcdms_amd import hip
    assert sorted(hip.SYMBOLS) == names, "ctypes binding and header disagree"
    assert hip.load().rcdm_version() == 0x000300


def test_argument_validation_without_gpu():
    """Entry points reject bad descriptors before touching the device."""
    from rcdms_amd import hip
    lib = hip.load()
    d = hip.GemmDesc(16, 12, 64, 64, 16, 0, 0, 1, 0, 1.0, 1)           # N % 8 != 0
    assert lib.rcdm_gemm(ctypes.byref(d), 8, 8, 0, 0, 0, 8, 0, 0, 0) == -2
    assert lib.rcdm_gemm(None, 8, 8, 0, 0, 0, 8, 0, 0, 0) == -1
    a = hip.AttnDesc(1, 8, 64, 64, 36, 288, 288, 288, 288, 0.1)        # d % 8 != 0
    assert lib.rcdm_flash_attn(ctypes.byref(a), 8, 8, 8, 8, 0) == -2
    g = hip.GroupNormDesc(2, 100, 320, 32, 320, 320, 1e-5, 1)
    assert lib.rcdm_groupnorm_workspace_bytes(ctypes.byref(g)) > 0
    assert lib.rcdm_groupnorm_silu(ctypes.byref(g), 8, 8, 8, 8, 0, 0, 0) == -4  # workspace missing
