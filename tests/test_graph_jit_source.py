"""The source generator behind mira_graph_specialize, without a GPU: the HIP text the emulation build writes for a compiled
graph has one statement per instruction of the stream, requests every column the graph reads (selectors as bytes, rotated
rows reduced once per row), and COMPILES for gfx950 with hiprtc -- which needs no device.  That the compiled kernel
computes the interpreter's values is the GPU suite's business (tests/test_gpu_graph_jit.py)."""
import ctypes
import os
import random
import re

import pytest

from graph_cases import gate_like_expression
from mira_amd import _lib
from harness import graph_evaluator as G
from harness import main_gate as MG

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mira_amd", "csrc")


def _hiprtc():
    for name in ("libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            pass
    return None


def _compiles(src):
    rtc = _hiprtc()
    if rtc is None:
        pytest.skip("no libhiprtc.so in this container")
    prog = ctypes.c_void_p()
    assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src.encode(), b"mira_jit.hip", 0, None, None) == 0
    opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-I" + CSRC.encode()]
    rc = rtc.hiprtcCompileProgram(prog, len(opts), (ctypes.c_char_p * len(opts))(*opts))
    n = ctypes.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
    log = ctypes.create_string_buffer(max(1, n.value))
    if n.value > 1:
        rtc.hiprtcGetProgramLog(prog, log)
    assert rc == 0, log.value.decode()[:2000]
    size = ctypes.c_size_t()
    assert rtc.hiprtcGetCodeSize(prog, ctypes.byref(size)) == 0 and size.value > 4096
    rtc.hiprtcDestroyProgram(ctypes.byref(prog))


def _compiles_with_embedded_headers(src):
    """The product library's own path: hiprtc + the kernel headers embedded in libmira_gpu.so (no -I, no files beside the
    library); mira_graph_jit_compile_check needs no device."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmira_gpu.so not built")
    lib = _lib.MiraLib(_lib.LIB_PATH)
    size = ctypes.c_size_t()
    rc = lib.c.mira_graph_jit_compile_check(src.encode(), ctypes.byref(size))
    if rc == _lib.MIRA_E_JIT_UNAVAILABLE:
        pytest.skip("no libhiprtc.so in this container")
    assert rc == 0, lib.c.mira_last_error().decode()[:2000]
    assert size.value > 4096
    bad = lib.c.mira_graph_jit_compile_check(b'#include "field29.cuh"\nthis is not HIP\n', ctypes.byref(size))
    assert bad == _lib.MIRA_E_JIT_FAILED and b"error" in lib.c.mira_last_error()      # a compiler failure is loud and carries the log


def test_generated_source_of_a_gate_like_graph(emu_lib):
    rng = random.Random(5)
    e = gate_like_expression(rng, 6, 7, 12, 3)                # 2 selectors, 3 fixed, 7 advice columns; rotations; 3 challenges
    ev = G.GraphEvaluator.new(e, G.FIELD_FQ)
    cols = [(1, G.COL_BOOL)] * 2 + [(1, G.COL_FIELD)] * 10
    src = ev.jit_source(cols, 3, lib=emu_lib)
    assert "using F = Fq29;" in src and 'extern "C" __global__' in src and "mira_jit_eval" in src
    stmts = re.findall(r"const Fe29<F> t(\d+) = ", src)
    assert [int(x) for x in stmts] == list(range(len(stmts))) and len(stmts) >= 6          # one SSA value per instruction, in order
    assert src.rstrip().endswith("}") and f"f29_pack(t{len(stmts) - 1})" in src           # the last one leaves
    code, consts, rots = ev.flatten()
    used_rot = {int(r) for r in rots if int(r) != 0}
    for r in used_rot:                                                                    # a rotated row is reduced once per row ...
        assert src.count(f"((int64_t)row + ({r})) % (int64_t)nrows") <= 1
    assert src.count("% (int64_t)nrows") <= len(used_rot)                                 # ... never per column read
    assert not ev.is_specialized(3, len(cols), lib=emu_lib)
    with pytest.warns(RuntimeWarning, match="no run-time compiler"):
        assert G.GraphEvaluator.specialize([ev], cols, 3, lib=emu_lib) is False           # the emulation has no run-time compiler, and says so
    # other column kinds, other source: a selector is read as a byte, a field column as 32 bytes
    src_f = ev.jit_source([(1, G.COL_FIELD)] * 12, 3, lib=emu_lib)
    assert "const uint32_t c" not in src_f and ("const uint32_t c" in src) == any(f"cols[{k}].p[" in src for k in (0, 1))
    _compiles(src)
    _compiles_with_embedded_headers(src)


def test_generated_source_of_a_main_gate_point_compiles(emu_lib):
    cg, ctx = MG.compressed_circuit(5, 1)
    plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, G.FIELD_FQ)
    ncol = ctx.num_fixed + 2 * ctx.num_advice
    src = plan.evaluators[2].jit_source([(1, G.COL_FIELD)] * ncol, 2 * ctx.num_challenges, lib=emu_lib)
    assert src.count("__builtin_amdgcn_sched_barrier(0);") >= 40                          # one scheduling region per product
    _compiles(src)


def test_cache_directory_must_be_private(tmp_path):
    """mira_graph_set_cache_dir: code objects found in the directory are executed, so one that others may write (or that is
    not a directory) is refused; "" / NULL switches the files off.  No device needed."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmira_gpu.so not built")
    lib = _lib.MiraLib(_lib.LIB_PATH)
    good, open_dir, plain = tmp_path / "mine", tmp_path / "everyone", tmp_path / "file"
    good.mkdir(mode=0o700); open_dir.mkdir(); os.chmod(open_dir, 0o777); plain.write_text("x")
    assert lib.c.mira_graph_set_cache_dir(os.fsencode(good)) == 0
    assert lib.c.mira_graph_set_cache_dir(os.fsencode(open_dir)) == _lib.MIRA_E_BAD_ARG and b"writable by nobody else" in lib.c.mira_last_error()
    assert lib.c.mira_graph_set_cache_dir(os.fsencode(plain)) == _lib.MIRA_E_IO
    assert lib.c.mira_graph_set_cache_dir(os.fsencode(tmp_path / "absent")) == _lib.MIRA_E_IO
    assert lib.c.mira_graph_set_cache_dir(None) == 0 and lib.c.mira_graph_set_cache_dir(b"") == 0
