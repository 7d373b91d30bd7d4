"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports every
symbol that include/inklayer_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    txt = (ROOT / "include" / "inklayer_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(ink_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from inklayer_amd import build, _lib
    build.build(verbose=False)
    import torch  # noqa: F401  torch's HIP runtime first: see inklayer_amd/_lib.py lib()
    l = ctypes.CDLL(str(_lib.lib_path()))
    syms = _declared_symbols()
    assert len(syms) >= 4
    for s in syms:
        assert hasattr(l, s), f"{s} declared in inklayer_hip.h but not exported"
    hdr = (ROOT / "include" / "inklayer_hip.h").read_text()
    assert l.ink_abi_version() == int(re.search(r"#define\s+INK_ABI_VERSION\s+(\d+)", hdr).group(1))


def test_python_binding_covers_header():
    from inklayer_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_bad_arguments_are_rejected_without_launch():
    # argument validation happens before any HIP call, so it is testable without a GPU
    from inklayer_amd import _lib
    l = _lib.lib()
    p = _lib.InkGemm()
    assert l.ink_gemm_f16(ctypes.byref(p), None) == 1          # null pointers
    p.A = p.W = p.C = 16
    p.M, p.N, p.K = 8, 8, 24                                    # K % 32 != 0
    p.lda = p.ldw = 24
    p.ldc = 8
    assert l.ink_gemm_f16(ctypes.byref(p), None) == 1
    assert l.ink_layernorm_rows(None, 0, None, None, 1e-6, None, 1, 4, None, None, 4, 0, 0, None, 0, None, 0, None) == 1
    assert l.ink_add_split_f16(16, None, 0, 16, 8, 3, None) == 1            # C % 4 != 0


def test_shipped_library_has_no_ablation_gemm_variants():
    """VERDICT r1 / ADVICE: the no-MFMA / no-epilogue / timeline GEMM kernels return garbage fast and must not be
    reachable in the product library (they live in the -DINK_ABLATION build used by tools/), and no environment
    variable may select a GEMM variant."""
    from inklayer_amd import _lib
    l = _lib.lib()
    for v in (21, 22, 23, 43, 44, 46, 48, 49, 50, 51, 52, 7, 99):
        assert l.ink_gemm_set_variant(v) == 1, v
    for v in (0, 10, 45, 445, 454, 455, -2, -1):
        assert l.ink_gemm_set_variant(v) == 0, v
    src = (ROOT / "inklayer_amd" / "csrc" / "gemm.hip").read_text()
    assert "getenv" not in src


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from inklayer_amd import _lib
    monkeypatch.setenv("INKLAYER_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    import pytest
    with pytest.raises(_lib.InkLayerHipError):
        _lib.lib()


def test_graft_entry_build_checks_pass():
    """The driver's build hook: (incremental) hipcc build, ABI version of the .so == the header's, imports."""
    import __graft_entry__ as g
    g.build()
