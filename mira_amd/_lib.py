"""ctypes binding of the C ABI declared in include/mira_gpu.h.

`load()` only ever opens the in-tree HIP library mira_amd/csrc/libmira_gpu.so and raises if it
is missing: there is no CPU fallback anywhere in this package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmira_gpu.so")

MIRA_OK = 0
MIRA_E_NO_DEVICE, MIRA_E_BAD_ARG, MIRA_E_TOO_LONG, MIRA_E_ALLOC, MIRA_E_UNSUPPORTED, MIRA_E_INVALID_POINT, MIRA_E_IO = -1, -2, -3, -4, -5, -6, -7
MIRA_E_JIT_UNAVAILABLE, MIRA_E_JIT_FAILED = -8, -9
MIRA_MAX_WINDOWS = 64
MIRA_PARTIAL_U64 = MIRA_MAX_WINDOWS * 16

# every symbol include/mira_gpu.h declares (tests check the library exports all of them)
SYMBOLS = [
    "mira_device_count", "mira_init", "mira_set_stream", "mira_last_error",
    "mira_msm_register_bases", "mira_msm_register_bases_device", "mira_msm_unregister", "mira_msm_check_bases", "mira_msm_precompute", "mira_msm_precompute_ex",
    "mira_msm_download_bases", "mira_fold_witness_device", "mira_fold_error_device", "mira_fold_relaxed_witness_device", "mira_g1_mul_add", "mira_g1_lincomb", "mira_g1_fold_commitments", "mira_graph_set_cache_dir", "mira_graph_jit_stats", "mira_graph_eval_device", "mira_graph_compile", "mira_graph_eval_compiled", "mira_graph_eval_batch", "mira_graph_free", "mira_pow_tree_reduce_device", "mira_lincomb_device",
    "mira_msm", "mira_msm_device", "mira_msm_batch", "mira_msm_batch_device", "mira_msm_partial_device", "mira_msm_combine", "mira_msm_set_window_bits", "mira_msm_last_plan",
    "mira_ntt_bn256_fr", "mira_ntt_bn256_fr_device", "mira_fft_bn256_fr", "mira_ifft_bn256_fr",
    "mira_fft_bn256_fr_device", "mira_ifft_bn256_fr_device", "mira_coset_fft_bn256_fr", "mira_coset_ifft_bn256_fr",
    "mira_get_omega_or_inv", "mira_synth_scalars_device", "mira_synth_bases_device",
    "mira_dev_alloc", "mira_dev_free", "mira_dev_upload", "mira_dev_download", "mira_dev_sync",
    "mira_set_timing", "mira_get_timings", "mira_set_tuning",
    "mira_msm_register_bases_file", "mira_msm_save_bases_file", "mira_msm_partial_to_device", "mira_msm_set_handle_window_bits",
    "mira_trim", "mira_dev_mem_info", "mira_msm_plan_window_bits", "mira_lincomb_multi_device", "mira_dev_copy", "mira_msm_last_table_bits",
    "mira_graph_specialize", "mira_graph_is_specialized", "mira_graph_jit_source", "mira_graph_jit_compile_check",
]
TUNE_STAGED_MIN_N, TUNE_TABLE_MIN_N, TUNE_PLAN_HIST_MIN_N, TUNE_NTT_MAX_LOG_LINE, TUNE_NTT_WAVE, TUNE_HOST_CHUNK_MIN_N, TUNE_NTT_SINGLE_TW_LOG, TUNE_NTT_FULL_TW_MAX_LOG, TUNE_TABLE_WIDTH, TUNE_JIT_LOADS_AHEAD, TUNE_MIN_SEGMENT, TUNE_GLV = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
TUNE_REDUCE_PIECES, TUNE_REDUCE_LAMBDA, TUNE_REDUCE_QUAD, TUNE_SHARED_MIN_N, TUNE_PASS_ENTRIES_LOG, TUNE_GLV_AUTO_MAX_LOG, TUNE_WIDTH_TRIALS = 12, 13, 14, 15, 16, 17, 18
TUNE_NTT_GRID = 19
TABLE_GLV = 2   # mira_msm_precompute_ex(handle, MIRA_TABLE_GLV): the endomorphism copy of a key


def _preload_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (same
    SONAME as /opt/rocm's); if libmira_gpu.so pulled in the system copy first and torch (needed
    for torch.distributed / RCCL) were imported later, the process would hold two runtimes and
    the second would see no device.  Loading torch's copy by path first makes both resolve to
    it, in either import order.  Without torch installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


class MiraError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmira_gpu error {code}: {msg}")
        self.code = code


class MiraGraph(ctypes.Structure):
    """struct mira_graph (include/mira_gpu.h)"""
    _fields_ = [("code", ctypes.c_void_p), ("code_words", ctypes.c_size_t), ("num_calculations", ctypes.c_uint32),
                ("num_constants", ctypes.c_uint32), ("constants", ctypes.c_void_p), ("rotations", ctypes.c_void_p),
                ("num_rotations", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class MiraEvalColumn(ctypes.Structure):
    """struct mira_eval_column (include/mira_gpu.h)"""
    _fields_ = [("d_data", ctypes.c_void_p), ("kind", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class MiraLib:
    """One loaded copy of the C ABI."""

    def __init__(self, path, preload_hip=False):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'); "
                "mira_amd has no CPU fallback")
        self.path = path
        if preload_hip:
            _preload_hip_runtime()
        self.c = ctypes.CDLL(path)
        c = self.c
        vp, u64p, sz, u64, u32, i32 = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32
        c.mira_last_error.restype = ctypes.c_char_p
        sig = {
            "mira_device_count": [], "mira_init": [ctypes.c_int], "mira_set_stream": [vp],
            "mira_msm_register_bases": [ctypes.c_int, u64p, sz, vp], "mira_msm_register_bases_device": [ctypes.c_int, vp, sz, vp],
            "mira_msm_unregister": [u64], "mira_msm_check_bases": [u64], "mira_msm_precompute": [u64], "mira_msm_precompute_ex": [u64, i32],
            "mira_msm": [u64, u64p, sz, u64p], "mira_msm_device": [u64, vp, sz, u64p],
            "mira_msm_download_bases": [u64, sz, sz, u64p],
            "mira_fold_witness_device": [ctypes.c_int, vp, vp, vp, u64p, sz], "mira_fold_error_device": [ctypes.c_int, vp, vp, sz, u64p, sz],
            "mira_g1_mul_add": [ctypes.c_int, u64p, u64p, u64p, u64p], "mira_g1_lincomb": [ctypes.c_int, u64p, u64p, u64p, sz, u64p],
            "mira_fold_relaxed_witness_device": [ctypes.c_int, vp, vp, vp, sz, vp, vp, vp, sz, u64p, sz],
            "mira_graph_set_cache_dir": [ctypes.c_char_p], "mira_graph_jit_stats": [vp, vp], "mira_graph_jit_compile_check": [ctypes.c_char_p, vp],
            "mira_g1_fold_commitments": [ctypes.c_int, u64p, u64p, u64p, sz, u64p, u64p, sz, u64p, u64p],
            "mira_graph_eval_device": [ctypes.c_int, vp, vp, u32, u64p, u32, sz, vp],
            "mira_graph_compile": [ctypes.c_int, vp, u32, u32, vp], "mira_graph_eval_compiled": [u64, vp, u32, u64p, u32, sz, vp], "mira_graph_eval_batch": [vp, u32, vp, u32, vp, u32, sz, vp], "mira_graph_free": [u64], "mira_graph_specialize": [vp, u32, vp, u32], "mira_graph_is_specialized": [u64, vp], "mira_graph_jit_source": [u64, vp, u32, vp, sz, vp],
            "mira_pow_tree_reduce_device": [ctypes.c_int, vp, sz, sz, u64p, u32, u64p],
            "mira_lincomb_device": [ctypes.c_int, vp, vp, u64p, sz, sz],
            "mira_msm_batch": [u64, vp, sz, sz, u64p], "mira_msm_batch_device": [u64, vp, sz, sz, sz, u64p],
            "mira_msm_partial_device": [u64, sz, vp, sz, u64p, vp, vp],
            "mira_msm_combine": [ctypes.c_int, u64p, sz, i32, i32, u64p], "mira_msm_set_window_bits": [i32], "mira_msm_last_plan": [vp, vp], "mira_msm_last_table_bits": [vp],
            "mira_ntt_bn256_fr": [u64p, u32, u64p], "mira_ntt_bn256_fr_device": [vp, u32, u64p],
            "mira_fft_bn256_fr": [u64p, u32], "mira_ifft_bn256_fr": [u64p, u32],
            "mira_fft_bn256_fr_device": [vp, u32], "mira_ifft_bn256_fr_device": [vp, u32],
            "mira_coset_fft_bn256_fr": [u64p, u32], "mira_coset_ifft_bn256_fr": [u64p, u32],
            "mira_get_omega_or_inv": [u32, ctypes.c_int, u64p],
            "mira_synth_scalars_device": [ctypes.c_int, sz, u64, u64, ctypes.c_int, vp],
            "mira_synth_bases_device": [ctypes.c_int, sz, u64, u64, vp],
            "mira_dev_alloc": [sz, vp], "mira_dev_free": [vp], "mira_dev_upload": [vp, vp, sz], "mira_dev_download": [vp, vp, sz],
            "mira_dev_sync": [], "mira_set_timing": [ctypes.c_int], "mira_get_timings": [vp, vp, ctypes.c_int],
            "mira_set_tuning": [ctypes.c_int, ctypes.c_int64],
            "mira_msm_register_bases_file": [ctypes.c_int, ctypes.c_char_p, u32, ctypes.c_int, vp], "mira_msm_save_bases_file": [u64, ctypes.c_char_p],
            "mira_msm_partial_to_device": [u64, sz, vp, sz, vp, vp, vp], "mira_msm_set_handle_window_bits": [u64, i32],
            "mira_trim": [sz, vp], "mira_dev_mem_info": [vp, vp], "mira_msm_plan_window_bits": [sz, vp],
            "mira_lincomb_multi_device": [ctypes.c_int, vp, sz, vp, sz, u64p, sz], "mira_dev_copy": [vp, vp, sz],
        }
        for name, args in sig.items():
            fn = getattr(c, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int

    def check(self, rc):
        if rc != MIRA_OK:
            raise MiraError(rc, (self.c.mira_last_error() or b"").decode())
        return rc

    # -- small conveniences shared by the host layer ------------------------------------------
    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        self.check(self.c.mira_dev_alloc(nbytes, ctypes.byref(p)))
        return p.value

    def free(self, ptr):
        self.c.mira_dev_free(ctypes.c_void_p(ptr))

    def upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.c.mira_dev_upload(ctypes.c_void_p(ptr), arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    def download(self, ptr, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        self.check(self.c.mira_dev_download(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), out.nbytes))
        return out

    def copy(self, dst, src, nbytes):
        self.check(self.c.mira_dev_copy(ctypes.c_void_p(dst), ctypes.c_void_p(src), nbytes))

    def tune(self, knob, value):
        """mira_set_tuning; value < 0 restores the default."""
        self.check(self.c.mira_set_tuning(knob, value))

    def trim(self, keep_bytes=0):
        """mira_trim: release the library's grow-only workspaces down to keep_bytes; returns the bytes released."""
        rel = ctypes.c_size_t()
        self.check(self.c.mira_trim(keep_bytes, ctypes.byref(rel)))
        return rel.value

    def mem_info(self):
        """(free, total) bytes of the bound device"""
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        self.check(self.c.mira_dev_mem_info(ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def timings(self):
        names = (ctypes.c_char_p * 32)()
        ms = (ctypes.c_float * 32)()
        n = self.c.mira_get_timings(names, ms, 32)
        return [(names[i].decode(), float(ms[i])) for i in range(min(n, 32))]


_lib = None


def load():
    """The product library.  Raises ImportError when the HIP build is absent."""
    global _lib
    if _lib is None:
        _lib = MiraLib(LIB_PATH, preload_hip=True)
    return _lib
