"""The C-ABI shared library loads and exports every symbol include/mira_gpu.h declares.
No compute calls: this runs without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mira_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mira_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from mira_amd import _lib
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from mira_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_product_fails_loudly_without_device():
    """No CPU fallback: on a box without a GPU every compute entry point reports NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mira_amd import _lib
    lib = _lib.load()
    assert lib.c.mira_init(0) == _lib.MIRA_E_NO_DEVICE
    assert b"no CPU fallback" in lib.c.mira_last_error()
    import numpy as np
    from mira_amd import commitment as cm
    with pytest.raises(_lib.MiraError):
        cm.CommitmentKey(0, np.zeros((1, 8), dtype=np.uint64))


def test_missing_library_raises_import_error(tmp_path):
    from mira_amd import _lib
    with pytest.raises(ImportError):
        _lib.MiraLib(str(tmp_path / "libmira_gpu.so"))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mira_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
