"""The C-ABI library loads (no GPU needed) and exports every symbol include/maus_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "maus_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(maus_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from adaptive_matrix_solver_amd import _cabi
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/maus_hip.h but not exported"
    assert sorted(_cabi.SYMBOLS) == names, "the ctypes binding and the header disagree"
    assert lib.maus_abi_version() == 1


def test_shipped_library_is_not_a_debug_build():
    """`make EXTRA=-DMAUS_PANEL_CLOCK` adds in-kernel clocks and one extra export (tools/panel_clocks.py); the library in
    the tree -- the one that travels to the GPU box -- must be the plain build."""
    from adaptive_matrix_solver_amd import _cabi
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    assert not hasattr(lib, "maus_debug_panel_clocks"), "libmaus_hip.so was built with -DMAUS_PANEL_CLOCK: run `make` in csrc/"


def test_binding_loads_and_types_every_entry_point():
    from adaptive_matrix_solver_amd import _cabi
    lib = _cabi.load_library()
    for n in _cabi.SYMBOLS:
        assert getattr(lib, n).argtypes is not None


def test_product_fails_loudly_without_device():
    """No CPU fallback: constructing a context without a GPU raises (on the GPU box it succeeds)."""
    import pytest
    from adaptive_matrix_solver_amd import _cabi
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(_cabi.MausHipError):
        _cabi.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "adaptive_matrix_solver_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no CPU fallback", ""), f"{fn} mentions the oracle"
            assert "fake_ctx" not in src or "test seam" in src
