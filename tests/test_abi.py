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


def build_c_consumer(lib_path, out_path):
    """tests/abi/consumer.c with the system C compiler against include/mira_gpu.h, linked to `lib_path`"""
    import subprocess
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "abi"),
           os.path.join(ROOT, "tests", "abi", "consumer.c"), "-o", str(out_path), "-L", libdir, f"-l:{libname}", f"-Wl,-rpath,{libdir}"]
    if os.path.isdir("/opt/rocm/lib"):                              # libmira_gpu.so's own dependency (libamdhip64)
        cmd += ["-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return str(out_path)


def test_c_consumer_compiles_and_runs_against_the_emulation(emu_lib, tmp_path):
    """A compiler, not a hand-written ctypes table, checks every signature and both struct layouts of the header;
    the same binary logic runs against libmira_gpu.so under -m gpu (tests/test_gpu_abi_consumer.py)."""
    import subprocess
    exe = build_c_consumer(emu_lib.path, tmp_path / "consumer_emu")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "consumer ok" in res.stdout, res.stdout + res.stderr


def test_ctypes_structs_match_the_layout_the_c_compiler_asserts():
    import ctypes
    from mira_amd._lib import MiraEvalColumn, MiraGraph
    assert ctypes.sizeof(MiraGraph) == 48 and ctypes.sizeof(MiraEvalColumn) == 16       # _Static_asserts of tests/abi/consumer.c
    assert [(f[0], getattr(MiraGraph, f[0]).offset) for f in MiraGraph._fields_] == [
        ("code", 0), ("code_words", 8), ("num_calculations", 16), ("num_constants", 20), ("constants", 24), ("rotations", 32), ("num_rotations", 40), ("reserved", 44)]
    assert [(f[0], getattr(MiraEvalColumn, f[0]).offset) for f in MiraEvalColumn._fields_] == [("d_data", 0), ("kind", 8), ("reserved", 12)]


def test_consumer_vectors_are_current():
    """tests/abi/vectors.h is what tests/abi/make_consumer_vectors.py writes from tests/golden/ref_kats.json"""
    import subprocess, sys
    path = os.path.join(ROOT, "tests", "abi", "vectors.h")
    before = open(path).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "abi", "make_consumer_vectors.py")])
    assert open(path).read() == before
