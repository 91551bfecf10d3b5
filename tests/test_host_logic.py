"""Host-side behaviour of the mirrored reference interface (no GPU: emulation library)."""
import numpy as np
import pytest

from mira_amd import _lib as _lib_mod
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


def test_window_width_is_per_handle(emu_lib):
    """mira_msm_set_handle_window_bits: two keys, two widths, neither sees the other's (the process-wide
    mira_msm_set_window_bits stays the default for keys without one); the point never depends on the width."""
    import ctypes
    lib = emu_lib
    n = 300
    keys = {cid: cm.CommitmentKey(cid, C.synth_bases(cid, n, seed=31 + cid), lib=lib) for cid in (0, 1)}
    d = {cid: cm.synth_scalars_device(cid, n, seed=41 + cid, lib=lib) for cid in (0, 1)}
    want = {cid: keys[cid].commit_device(d[cid], n) for cid in (0, 1)}

    def last_plan():
        c, w = ctypes.c_int32(), ctypes.c_int32()
        lib.check(lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
        return c.value, w.value
    keys[0].set_window_bits(7)
    keys[1].set_window_bits(11)
    for cid, c in ((0, 7), (1, 11), (0, 7)):
        assert (keys[cid].commit_device(d[cid], n) == want[cid]).all()
        assert last_plan() == (c, (256 + c - 1) // c)
    lib.check(lib.c.mira_msm_set_window_bits(9))               # process default: only for keys without a width of their own
    try:
        assert (keys[0].commit_device(d[0], n) == want[0]).all() and last_plan()[0] == 7
        keys[0].set_window_bits(0)
        assert (keys[0].commit_device(d[0], n) == want[0]).all() and last_plan()[0] == 9
        assert (keys[0].commit_batch_device(d[0], n, 1)[0] == want[0]).all() and last_plan()[0] == 9
    finally:
        lib.check(lib.c.mira_msm_set_window_bits(0))
    with pytest.raises(_lib_mod.MiraError):
        keys[0].set_window_bits(3)
    assert lib.c.mira_msm_set_handle_window_bits(424242, 8) == _lib_mod.MIRA_E_BAD_ARG
    for cid in (0, 1):
        lib.free(d[cid]); keys[cid].close()


def test_trim_releases_workspaces_and_calls_recover(emu_lib):
    """mira_trim: grow-only workspaces and cached NTT table sets go back; the next calls re-allocate"""
    lib = emu_lib
    n = 2000
    key = cm.CommitmentKey(0, C.synth_bases(0, n, seed=51), lib=lib)
    sc = C.synth_scalars(0, n, seed=52)
    want = key.commit(sc)
    a = C.synth_scalars(0, 1 << 9, seed=53)
    want_fft = F.fft(a, 9, lib=lib)
    released = lib.trim(0)
    assert released > n * 16 * 2                                # at least the digit buffer of that commit
    assert lib.trim(0) == 0                                     # nothing left to release
    assert (key.commit(sc) == want).all()
    assert (F.fft(a, 9, lib=lib) == want_fft).all()
    assert (F.ifft(F.fft(a, 9, lib=lib), 9, lib=lib) == a).all()
    big = lib.trim(1 << 40)                                     # keep more than there is: nothing happens
    assert big == 0 and (key.commit(sc) == want).all()
    key.close()


def test_partial_to_device_equals_host_partial(emu_lib):
    lib = emu_lib
    n = 700
    key = cm.CommitmentKey(1, C.synth_bases(1, n, seed=61), lib=lib)
    d = cm.synth_scalars_device(1, n, seed=62, lib=lib)
    d_out = lib.alloc(_lib_mod.MIRA_PARTIAL_U64 * 8)
    for first, cnt, c in ((0, n, 9), (100, 333, 11), (n, 0, 0)):       # (width 0 = the 16-bit default of a sharded partial: only on the empty chunk, the emulation is slow at 2^15 buckets)
        part, c1, w1 = key.commit_partial_device(first, d + 0, cnt, window_bits=c)
        c2, w2 = key.commit_partial_to_device(first, d + 0, cnt, d_out, window_bits=c)
        assert (c1, w1) == (c2, w2)
        got = lib.download(d_out, (_lib_mod.MIRA_PARTIAL_U64,))
        assert not got[w2 * 16:].any()                           # words beyond the partial's windows are zero
        # (the projective representation of a window sum depends on the order the sort's atomics left the
        # entries in; the point does not)
        assert (cm.combine_partials(1, got, c2, w2, lib=lib) == cm.combine_partials(1, part, c1, w1, lib=lib)).all()
    with pytest.raises(cm.TooLongInput):
        key.commit_partial_to_device(1, d, n, d_out)
    lib.free(d); lib.free(d_out); key.close()


def test_key_file_errors(emu_lib, tmp_path):
    import ctypes
    h = ctypes.c_uint64()
    assert emu_lib.c.mira_msm_register_bases_file(0, str(tmp_path / "missing.bin").encode(), 4, 1, ctypes.byref(h)) == _lib_mod.MIRA_E_IO
    assert b"missing.bin" in emu_lib.c.mira_last_error()
    (tmp_path / "short.bin").write_bytes(b"\0" * (15 * 64))
    assert emu_lib.c.mira_msm_register_bases_file(0, str(tmp_path / "short.bin").encode(), 4, 0, ctypes.byref(h)) == _lib_mod.MIRA_E_IO
    assert emu_lib.c.mira_last_error() == b"failed to fill whole buffer"
    assert emu_lib.c.mira_msm_register_bases_file(5, str(tmp_path / "short.bin").encode(), 2, 0, ctypes.byref(h)) == _lib_mod.MIRA_E_BAD_ARG
    # the all-zero file is 16 identity points: loads, validates (the identity passes is_on_curve), commits to the identity
    (tmp_path / "zeros.bin").write_bytes(b"\0" * (16 * 64))
    key = cm.CommitmentKey.load_from_file(0, tmp_path / "zeros.bin", 4, lib=emu_lib, validate=True)
    assert not key.commit(C.synth_scalars(0, 16, seed=3)).any()
    key.close()


def test_lincomb_multi_matches_single_lincombs(emu_lib):
    """mira_lincomb_multi_device: M combinations of the same J vectors in one sweep = M mira_lincomb_device calls"""
    import ctypes
    lib, n, J, M, field = emu_lib, 37, 5, 3, 1
    vecs = [C.synth_scalars(0, n, seed=300 + j) for j in range(J)]
    coeffs = C.synth_scalars(0, M * J, seed=400)
    d_v = [lib.alloc(n * 32) for _ in range(J)]
    for p, v in zip(d_v, vecs):
        lib.upload(p, v)
    d_multi = [lib.alloc(n * 32) for _ in range(M)]
    d_one = lib.alloc(n * 32)
    vp = (ctypes.c_void_p * J)(*d_v)
    lib.check(lib.c.mira_lincomb_multi_device(field, (ctypes.c_void_p * M)(*d_multi), M, vp, J, coeffs.ctypes.data_as(ctypes.c_void_p), n))
    for m in range(M):
        cm_ = np.ascontiguousarray(coeffs[m * J:(m + 1) * J])
        lib.check(lib.c.mira_lincomb_device(field, ctypes.c_void_p(d_one), vp, cm_.ctypes.data_as(ctypes.c_void_p), J, n))
        assert (lib.download(d_multi[m], (n, 4)) == lib.download(d_one, (n, 4))).all()
    # an output that aliases an input, too many outputs, a null vector
    bad = (ctypes.c_void_p * M)(d_v[0], d_multi[1], d_multi[2])
    assert lib.c.mira_lincomb_multi_device(field, bad, M, vp, J, coeffs.ctypes.data_as(ctypes.c_void_p), n) == _lib_mod.MIRA_E_BAD_ARG
    assert lib.c.mira_lincomb_multi_device(field, (ctypes.c_void_p * 9)(*([d_one] * 9)), 9, vp, J, coeffs.ctypes.data_as(ctypes.c_void_p), n) == _lib_mod.MIRA_E_BAD_ARG
    twice = (ctypes.c_void_p * M)(d_multi[0], d_multi[1], d_multi[0])        # two outputs in one buffer: last writer would win
    assert lib.c.mira_lincomb_multi_device(field, twice, M, vp, J, coeffs.ctypes.data_as(ctypes.c_void_p), n) == _lib_mod.MIRA_E_BAD_ARG
    assert b"share one buffer" in lib.c.mira_last_error()
    for p in d_v + d_multi + [d_one]:
        lib.free(p)


def test_width_trials_settle_and_never_change_the_point(emu_lib):
    """MIRA_TUNE_WIDTH_TRIALS (default on): the planner's width for a shape of commit is checked against its neighbours on the first
    commits of that shape -- the model's width and its four neighbours, each timed twice -- and the fastest measured is kept.  Every commit returns the same point; the width settles; with the knob off the planner's width is
    used from the first call on; a forced width is never tried against anything."""
    import ctypes
    lib = emu_lib
    cid, n = 1, 1 << 12
    key = cm.CommitmentKey.synthetic(cid, n, seed=71, lib=lib)
    d = cm.synth_scalars_device(cid, n, seed=72, kind=1, lib=lib)
    lib.tune(_lib_mod.TUNE_GLV_AUTO_MAX_LOG, 0)
    c, w = ctypes.c_int32(), ctypes.c_int32()

    def commit():
        pt = key.commit_device(d, n)
        lib.check(lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
        return pt, c.value
    try:
        lib.tune(_lib_mod.TUNE_WIDTH_TRIALS, 0)
        want, c_model = commit()
        assert commit()[1] == c_model
        lib.tune(_lib_mod.TUNE_WIDTH_TRIALS, -1)
        widths = []
        for _ in range(14):                                    # 2 runs x (the model's width + four neighbours) = 10 calls, then settled
            pt, cw = commit()
            assert (pt == want).all()
            widths.append(cw)
        c_stats = widths[0]                                    # (the model's width once the statistics of the first commit exist)
        assert widths[:10] == [c_stats, c_stats, c_stats + 1, c_stats + 1, c_stats - 1, c_stats - 1, c_stats + 2, c_stats + 2, c_stats - 2, c_stats - 2]
        assert len(set(widths[10:])) == 1 and abs(widths[-1] - c_stats) <= 2
        key.set_window_bits(7)
        assert commit()[1] == 7 and commit()[1] == 7
    finally:
        lib.tune(_lib_mod.TUNE_WIDTH_TRIALS, -1); lib.tune(_lib_mod.TUNE_GLV_AUTO_MAX_LOG, -1)
        lib.free(d); key.close()


def test_planner_tables_are_sane(emu_lib):
    """The window width comes from measured wall-time tables (capi.hip: plan_wall_us, glv_wall_us, shared_wall_us) that go stale
    whenever a tail kernel changes (round 4: the mid-round table cost planned 2^17-pair commits 17 %).  This cannot re-measure them;
    it pins what every calibration so far agrees on, so that a mis-typed row shows: widths never shrink as commits grow from 2^18 pairs
    (12 and 13 bits tie below), the fold step's 2^16 .. 2^18-pair commits take 12 or 13 bits, and everything from 2^21 pairs takes the
    full 16 -- a pure function of n (the ranks of a sharded MSM rely on that)."""
    import ctypes

    def plan(n):
        c = ctypes.c_int32()
        emu_lib.check(emu_lib.c.mira_msm_plan_window_bits(n, ctypes.byref(c)))
        return c.value
    widths = [plan(1 << k) for k in range(6, 29)]
    assert all(4 <= c <= 16 for c in widths)
    from18 = widths[18 - 6:]
    assert from18 == sorted(from18), widths
    assert all(plan(1 << k) in (12, 13) for k in (16, 17, 18)), widths
    assert all(plan(1 << k) == 16 for k in range(21, 29)), widths
    assert plan(3 << 20) == plan(3 << 20) and plan(0) == plan(1)
