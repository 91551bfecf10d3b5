"""Host-side mirror of the reference's `CommitmentKey` (src/commitment.rs:26-87) for the GPU path.

Same names, argument meaning and error behaviour as the Rust type: `commit(v)` is the MSM of `v`
against the PREFIX of the key and returns one affine point; a `v` longer than the key raises
`TooLongInput` carrying `input_len` and `limit` (src/commitment.rs:21-24).  Points and scalars are
numpy uint64 arrays in the in-memory layout of halo2curves: scalars (n, 4), points (n, 8),
Montgomery form, identity = zeros.
"""
import ctypes

import numpy as np

from . import _lib

CURVE_BN256 = 0
CURVE_GRUMPKIN = 1


class TooLongInput(Exception):
    """Error::TooLongInput { input_len, limit } (src/commitment.rs:21-24)."""

    def __init__(self, input_len, limit):
        super().__init__(f"Can't commit too long input: input len: {input_len}, but limit is {limit}")
        self.input_len, self.limit = input_len, limit


def _as_u64(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, width)


class CommitmentKey:
    """`CommitmentKey<C>`: a boxed slice of affine bases, resident in HBM for its whole life
    (the reference keeps one key per curve for all fold steps, src/ivc/public_params.rs:50)."""

    def __init__(self, curve, ck=None, *, device_ptr=None, length=None, lib=None):
        self.lib = lib or _lib.load()
        self.curve = curve
        h = ctypes.c_uint64()
        if device_ptr is not None:
            self._len = int(length)
            self.lib.check(self.lib.c.mira_msm_register_bases_device(curve, ctypes.c_void_p(device_ptr), self._len, ctypes.byref(h)))
        else:
            ck = _as_u64(ck if ck is not None else np.zeros((0, 8), np.uint64), 8)
            self._len = len(ck)
            self.lib.check(self.lib.c.mira_msm_register_bases(curve, ck.ctypes.data_as(ctypes.c_void_p), self._len, ctypes.byref(h)))
        self.handle = h.value

    @classmethod
    def synthetic(cls, curve, n, seed=0x42415345, index0=0, lib=None):
        """n deterministic bases P_i = k_i * G generated on the GPU; this is the
        benchmark's stand-in for `setup` (src/commitment.rs:52-76), whose hash-to-curve lives in
        the absent halo2curves crate."""
        lib = lib or _lib.load()
        ptr = lib.alloc(max(n, 1) * 64)
        try:
            lib.check(lib.c.mira_synth_bases_device(curve, n, index0, seed, ctypes.c_void_p(ptr)))
            return cls(curve, device_ptr=ptr, length=n, lib=lib)
        finally:
            lib.free(ptr)                               # register keeps its own resident copy: one copy of the key in HBM

    def __len__(self):
        return self._len

    def len(self):
        return self._len

    def is_empty(self):
        return self._len == 0

    @staticmethod
    def default_value():
        """C::identity() (src/commitment.rs:40-42)"""
        return np.zeros(8, dtype=np.uint64)

    def bases(self):
        """The whole key back in the reference layout (tests)."""
        return self.download()

    # ---- commitment-key cache (src/commitment.rs:96-167): raw dump of [C], 2^k x 64 bytes ----
    def download(self, first=0, n=None):
        """The registered key (or a range) back in the reference layout."""
        n = self._len - first if n is None else n
        out = np.empty((n, 8), dtype=np.uint64)
        self.lib.check(self.lib.c.mira_msm_download_bases(self.handle, first, n, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def save_to_file(self, file_path):
        """`save_to_file` (src/commitment.rs:96-101): the slice as raw bytes (mira_msm_save_bases_file)."""
        self.lib.check(self.lib.c.mira_msm_save_bases_file(self.handle, str(file_path).encode()))

    @classmethod
    def load_from_file(cls, curve, file_path, k, lib=None, validate=False):
        """`load_from_file` (src/commitment.rs:110-127): 2^k points straight from the file into HBM
        (mira_msm_register_bases_file: chunked pread into pinned buffers beside the conversion of the previous
        chunk); validate = the is_on_curve pass of load_or_setup_cache folded into the same sweep.
        A short file raises IOError("failed to fill whole buffer") like read_exact."""
        self = cls.__new__(cls)
        self.lib = lib or _lib.load()
        self.curve, self._len = curve, 1 << k
        h = ctypes.c_uint64()
        rc = self.lib.c.mira_msm_register_bases_file(curve, str(file_path).encode(), k, 1 if validate else 0, ctypes.byref(h))
        if rc == _lib.MIRA_E_IO:
            raise IOError((self.lib.c.mira_last_error() or b"").decode())
        if rc == _lib.MIRA_E_INVALID_POINT:
            raise IOError("Wrong file in cache, some ptr out of curve")
        self.lib.check(rc)
        self.handle = h.value
        return self

    @classmethod
    def load_or_setup_cache(cls, curve, cache_folder, label, k, lib=None):
        """`load_or_setup_cache` (src/commitment.rs:134-166): `{cache_folder}/{label}/{k}.bin`;
        a loaded key is validated point by point (is_on_curve) on the GPU and rejected with the
        reference's message otherwise.  A missing file is generated and stored -- with the
        synthetic generator here, because `setup`'s hash-to-curve lives in the absent halo2curves."""
        import os
        path = os.path.join(cache_folder, label, f"{k}.bin")
        if os.path.exists(path):
            return cls.load_from_file(curve, path, k, lib=lib, validate=True)
        key = cls.synthetic(curve, 1 << k, lib=lib)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        key.save_to_file(path)
        return key

    def precompute(self, window_bits=20):
        """Build the fixed-base window tables in HBM.  20-bit windows (13 x the key size): large
        commits take 13 instead of 16 bucket additions per pair.  16-bit windows (16 x): all windows
        share one bucket set -- the small commits of a fold step lose most of their latency-bound
        tail.  window_bits = _lib.TABLE_GLV: the endomorphism copy of the key (2 x), single commits then split every
        scalar into two 128-bit halves over half the windows.  Results are bit-identical either way."""
        self.lib.check(self.lib.c.mira_msm_precompute_ex(self.handle, window_bits))
        self.precomputed = True
        return self

    def check_on_curve(self):
        """load_or_setup_cache's validation (src/commitment.rs:145-154), on the GPU."""
        self.lib.check(self.lib.c.mira_msm_check_bases(self.handle))

    def commit(self, v):
        """src/commitment.rs:78-87.  v: (n, 4) uint64 Montgomery scalars in host memory."""
        v = _as_u64(v, 4)
        if len(v) > self._len:
            raise TooLongInput(len(v), self._len)
        out = np.empty(8, dtype=np.uint64)
        self.lib.check(self.lib.c.mira_msm(self.handle, v.ctypes.data_as(ctypes.c_void_p), len(v), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def commit_device(self, d_scalars, n):
        """Same with the scalars already in HBM (device pointer)."""
        if n > self._len:
            raise TooLongInput(n, self._len)
        out = np.empty(8, dtype=np.uint64)
        self.lib.check(self.lib.c.mira_msm_device(self.handle, ctypes.c_void_p(d_scalars), n, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def commit_batch(self, vs):
        """`vs.iter().map(|v| ck.commit(v))` for equal-length vectors (the cross-term commits of
        src/nifs/vanilla/mod.rs:124-127) in one submission; returns (count, 8)."""
        vs = [_as_u64(v, 4) for v in vs]
        if not vs:
            return np.zeros((0, 8), dtype=np.uint64)
        n = len(vs[0])
        if any(len(v) != n for v in vs):
            raise ValueError("commit_batch needs vectors of equal length")
        if n > self._len:
            raise TooLongInput(n, self._len)
        ptrs = (ctypes.c_void_p * len(vs))(*[v.ctypes.data for v in vs])
        out = np.empty((len(vs), 8), dtype=np.uint64)
        self.lib.check(self.lib.c.mira_msm_batch(self.handle, ptrs, n, len(vs), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def commit_batch_device(self, d_scalars, n, count, stride=None):
        """Same with the vectors in HBM: vector b starts at element b * stride (default n)."""
        if n > self._len:
            raise TooLongInput(n, self._len)
        out = np.empty((count, 8), dtype=np.uint64)
        self.lib.check(self.lib.c.mira_msm_batch_device(self.handle, ctypes.c_void_p(d_scalars), n, count, n if stride is None else stride,
                                                        out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def commit_partial_device(self, first, d_scalars, n, window_bits=0):
        """Window sums of sum_i v[i] * ck[first + i]; combine with `combine_partials`.  All partials
        of one MSM must be cut with the same `window_bits` (0: the library's length-independent default)."""
        if first + n > self._len:
            raise TooLongInput(first + n, self._len)
        part = np.zeros(_lib.MIRA_PARTIAL_U64, dtype=np.uint64)
        c, w = ctypes.c_int32(window_bits), ctypes.c_int32()
        self.lib.check(self.lib.c.mira_msm_partial_device(self.handle, first, ctypes.c_void_p(d_scalars), n,
                                                          part.ctypes.data_as(ctypes.c_void_p), ctypes.byref(c), ctypes.byref(w)))
        return part, c.value, w.value

    def commit_partial_to_device(self, first, d_scalars, n, d_out, window_bits=0):
        """The same with the MIRA_PARTIAL_U64 words left in device memory at `d_out` (the buffer an RCCL all-gather
        reads); returns (window_bits, num_windows)."""
        if first + n > self._len:
            raise TooLongInput(first + n, self._len)
        c, w = ctypes.c_int32(window_bits), ctypes.c_int32()
        self.lib.check(self.lib.c.mira_msm_partial_to_device(self.handle, first, ctypes.c_void_p(d_scalars), n, ctypes.c_void_p(d_out),
                                                             ctypes.byref(c), ctypes.byref(w)))
        return c.value, w.value

    def set_window_bits(self, c):
        """Window width of every later commit over THIS key (4..16; 0 = planner): mira_msm_set_handle_window_bits."""
        self.lib.check(self.lib.c.mira_msm_set_handle_window_bits(self.handle, c))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.c.mira_msm_unregister(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def combine_partials(curve, partials, window_bits, num_windows, lib=None):
    """Sum per-GPU window sums, Horner over the windows, to_affine."""
    lib = lib or _lib.load()
    partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, _lib.MIRA_PARTIAL_U64)
    out = np.empty(8, dtype=np.uint64)
    lib.check(lib.c.mira_msm_combine(curve, partials.ctypes.data_as(ctypes.c_void_p), len(partials), window_bits, num_windows,
                                     out.ctypes.data_as(ctypes.c_void_p)))
    return out


def concatenate_with_padding(vs, pad_size):
    """src/util.rs:189-193: the MSM scalar vector of a witness: columns zero-padded to pad_size."""
    cols = []
    for v in vs:
        v = _as_u64(v, 4)
        if len(v) < pad_size:
            v = np.concatenate([v, np.zeros((pad_size - len(v), 4), dtype=np.uint64)])
        cols.append(v)
    return np.concatenate(cols) if cols else np.zeros((0, 4), dtype=np.uint64)


def synth_scalars_device(curve, n, seed=0x4D495241, kind=0, index0=0, lib=None):
    """Device buffer of n synthetic scalars (kind 0 uniform, 1 witness-like).  Returns pointer."""
    lib = lib or _lib.load()
    ptr = lib.alloc(max(n, 1) * 32)
    lib.check(lib.c.mira_synth_scalars_device(curve, n, index0, seed, kind, ctypes.c_void_p(ptr)))
    return ptr
