"""Host-side behaviour of the mirrored reference interface (no GPU: emulation library)."""
import numpy as np
import pytest

from mira_amd import commitment as cm
from mira_amd import fft as F
from oracle import cref as C


def test_commit_too_long_input_matches_reference_error(emu_lib):
    # src/commitment.rs:78-87 + :21-24
    key = cm.CommitmentKey(0, C.synth_bases(0, 4), lib=emu_lib)
    with pytest.raises(cm.TooLongInput) as e:
        key.commit(C.synth_scalars(0, 5))
    assert (e.value.input_len, e.value.limit) == (5, 4)
    assert str(e.value) == "Can't commit too long input: input len: 5, but limit is 4"


def test_c_abi_too_long_code(emu_lib):
    import ctypes
    from mira_amd._lib import MIRA_E_TOO_LONG
    key = cm.CommitmentKey(0, C.synth_bases(0, 4), lib=emu_lib)
    sc = C.synth_scalars(0, 5)
    out = np.empty(8, dtype=np.uint64)
    rc = emu_lib.c.mira_msm(key.handle, sc.ctypes.data_as(ctypes.c_void_p), 5, out.ctypes.data_as(ctypes.c_void_p))
    assert rc == MIRA_E_TOO_LONG
    assert b"input len: 5, but limit is 4" in emu_lib.c.mira_last_error()


def test_commit_uses_prefix_of_key(emu_lib):
    bs = C.synth_bases(0, 12)
    sc = C.synth_scalars(0, 7)
    key = cm.CommitmentKey(0, bs, lib=emu_lib)
    assert (key.commit(sc) == C.msm_naive(0, sc, bs[:7])).all()
    assert key.len() == 12 and not key.is_empty()


def test_empty_commit_is_identity(emu_lib):
    key = cm.CommitmentKey(0, C.synth_bases(0, 3), lib=emu_lib)
    assert not key.commit(np.zeros((0, 4), dtype=np.uint64)).any()
    assert not cm.CommitmentKey.default_value().any()
    empty = cm.CommitmentKey(1, np.zeros((0, 8), dtype=np.uint64), lib=emu_lib)
    assert empty.is_empty() and not empty.commit(np.zeros((0, 4), dtype=np.uint64)).any()


def test_all_zero_scalars_and_identity_bases(emu_lib):
    bs = C.synth_bases(1, 9)
    key = cm.CommitmentKey(1, bs, lib=emu_lib)
    assert not key.commit(np.zeros((9, 4), dtype=np.uint64)).any()
    zkey = cm.CommitmentKey(1, np.zeros((9, 8), dtype=np.uint64), lib=emu_lib)
    assert not zkey.commit(C.synth_scalars(1, 9)).any()


def test_concatenate_with_padding():
    # src/util.rs:189-193
    a, b = C.synth_scalars(0, 3, seed=1), C.synth_scalars(0, 4, seed=2)
    out = cm.concatenate_with_padding([a, b], 4)
    assert out.shape == (8, 4)
    assert (out[:3] == a).all() and not out[3].any() and (out[4:] == b).all()


def test_fft_size_asserts(emu_lib):
    a = C.synth_scalars(0, 8)
    with pytest.raises(AssertionError):
        F.fft(a, 4, lib=emu_lib)            # src/fft.rs:65
    with pytest.raises(AssertionError):
        F.get_omega_or_inv(29, False, lib=emu_lib)   # src/fft.rs:13
    with pytest.raises(AssertionError):
        F.coset_fft(a[:6], lib=emu_lib)     # src/fft.rs:179


def test_unknown_handle_and_bad_args(emu_lib):
    import ctypes
    from mira_amd._lib import MIRA_E_BAD_ARG
    out = np.empty(8, dtype=np.uint64)
    sc = C.synth_scalars(0, 1)
    assert emu_lib.c.mira_msm(987654321, sc.ctypes.data_as(ctypes.c_void_p), 1, out.ctypes.data_as(ctypes.c_void_p)) == MIRA_E_BAD_ARG
    assert emu_lib.c.mira_msm_set_window_bits(17) == MIRA_E_BAD_ARG
    h = ctypes.c_uint64()
    assert emu_lib.c.mira_msm_register_bases(7, None, 0, ctypes.byref(h)) == MIRA_E_BAD_ARG


def _two_thread_abi(lib, n, log_n, rounds):
    """Two caller threads, each with its own key (one per curve), interleaving mira_msm and
    mira_fft_bn256_fr through the same library: the ABI promises re-entrancy (include/mira_gpu.h),
    as `cargo test` calls commit from parallel test threads holding only &CommitmentKey."""
    import threading
    errors, results = [], {}

    def worker(cid):
        try:
            bs = C.synth_bases(cid, n, seed=70 + cid)
            key = cm.CommitmentKey(cid, bs, lib=lib)
            a = C.synth_scalars(0, 1 << log_n, seed=80 + cid)
            want_fft = C.fft(a, log_n)
            out = []
            for r in range(rounds):
                sc = C.synth_scalars(cid, n - r, seed=90 + 10 * cid + r, kind=r % 2)
                out.append((key.commit(sc), C.commit(cid, bs, sc)))
                assert (F.fft(a, log_n, lib=lib) == want_fft).all()
            results[cid] = out
        except Exception as e:      # surfaced in the main thread
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(cid,)) for cid in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for cid in (0, 1):
        assert len(results[cid]) == rounds
        for got, want in results[cid]:
            assert (got == want).all()


def test_abi_from_two_threads(emu_lib):
    _two_thread_abi(emu_lib, n=40, log_n=5, rounds=3)
