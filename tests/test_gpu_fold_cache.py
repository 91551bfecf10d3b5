"""GPU parity for SURVEY.md 8(f) rows N2 (RelaxedPlonkWitness::fold, src/plonk/mod.rs:1097-1134)
and N3 (commitment-key cache, src/commitment.rs:96-167)."""
import numpy as np
import pytest

from mira_amd import commitment as cm
from mira_amd import fold as FD
from oracle import cref as C

pytestmark = pytest.mark.gpu
CURVE_OF_FIELD = {1: 0, 0: 1}


@pytest.mark.parametrize("field,n", [(1, 1 << 20), (0, 1 << 17), (1, 12345), (1, 1)])
def test_fold_parity(gpu_lib, field, n):
    cid = CURVE_OF_FIELD[field]
    w1, w2 = C.synth_scalars(cid, n, seed=4, kind=1), C.synth_scalars(cid, n, seed=5)
    r = C.synth_scalars(cid, 1, seed=6)[0]
    assert (FD.fold_witness(field, w1, w2, r) == C.fold_witness(field, w1, w2, r)).all()
    terms = [C.synth_scalars(cid, n, seed=20 + k) for k in range(6)]
    assert (FD.fold_error(field, w1, terms, r) == C.fold_error(field, w1, terms, r)).all()


@pytest.mark.parametrize("field", [0, 1])
def test_fold_relaxed_witness_in_one_submission(gpu_lib, field):
    """mira_fold_relaxed_witness_device at the fold step's shapes (14 / 7 columns of 2^17 rows, 6 / 5 cross terms) and at ragged
    ones, out of place and in place, against the oracle."""
    from test_fold_and_cache import _relaxed_fold_case
    cols, nterms = (14, 6) if field == 1 else (7, 5)
    for n_w, n, k, in_place in ((cols << 17, 1 << 17, nterms, False), (12345, 777, 16, True), (0, 4096, 2, False), (4096, 0, 0, False), (513, 511, 0, False)):
        _relaxed_fold_case(gpu_lib, field, n_w, n, k, in_place)


def test_fold_then_commit_matches_folded_commitment(gpu_lib):
    """Com(W1 + r W2) == Com(W1) + r Com(W2) (src/plonk/mod.rs:547-557) with fold, commits and the
    commitment-side fold all on this library, vectors staying in HBM between the steps."""
    for cid, field in ((0, 1), (1, 0)):
        n = 1 << 16
        key = cm.CommitmentKey.synthetic(cid, n, seed=33)
        d1 = cm.synth_scalars_device(cid, n, seed=34, kind=1)
        d2 = cm.synth_scalars_device(cid, n, seed=35)
        r = C.synth_scalars(cid, 1, seed=36)[0]
        c1, c2 = key.commit_device(d1, n), key.commit_device(d2, n)
        FD.fold_witness_device(field, d1, d1, d2, r, n)          # in place: W1 <- W1 + r W2
        assert (key.commit_device(d1, n) == FD.g1_mul_add(cid, c1, r, c2)).all()


def test_key_cache_roundtrip(gpu_lib, tmp_path):
    cid, k = 1, 12
    key = cm.CommitmentKey.synthetic(cid, 1 << k, seed=77)
    original = key.bases()
    assert (key.download() == original).all()
    assert (original == C.synth_bases(cid, 1 << k, seed=77)).all()
    path = tmp_path / f"{k}.bin"
    key.save_to_file(path)
    loaded = cm.CommitmentKey.load_from_file(cid, path, k)
    loaded.check_on_curve()
    sc = C.synth_scalars(cid, 1 << k, seed=78)
    assert (loaded.commit(sc) == C.commit(cid, original, sc)).all()
    raw = bytearray(path.read_bytes()); raw[64 * 100 + 3] ^= 4; path.write_bytes(bytes(raw))
    folder = tmp_path / "cache" / "grumpkin"
    folder.mkdir(parents=True)
    (folder / f"{k}.bin").write_bytes(bytes(raw))
    with pytest.raises(IOError, match="Wrong file in cache, some ptr out of curve"):
        cm.CommitmentKey.load_or_setup_cache(cid, str(tmp_path / "cache"), "grumpkin", k)


@pytest.mark.parametrize("cid", [0, 1])
def test_key_file_into_hbm_in_chunks(gpu_lib, tmp_path, cid):
    """mira_msm_register_bases_file: a key of several 64 MiB chunks (2^21 + ragged tail is not a power of two, so
    k = 21 reads the first 2^21 points of a longer file) through the two-buffer pipeline; the loaded key commits to
    the oracle's point and exports the file's bytes; validation is folded into the same sweep."""
    k = 21
    n = (1 << k) + 12345                                        # the file is LONGER than 2^k points: load_from_file reads exactly 2^k
    key = cm.CommitmentKey.synthetic(cid, n, seed=91 + cid)
    path = tmp_path / "key.bin"
    key.save_to_file(path)
    assert path.stat().st_size == n * 64
    loaded = cm.CommitmentKey.load_from_file(cid, path, k, validate=True)
    assert len(loaded) == 1 << k
    rng = np.random.default_rng(5)
    for first in (0, 1 << 20, (1 << 21) - 4096, int(rng.integers(0, (1 << 21) - 4096))):      # chunk boundaries and a random spot
        assert (loaded.download(first, 4096) == key.download(first, 4096)).all()
    m = 1 << 16
    sc = C.synth_scalars(cid, m, seed=93)
    assert (loaded.commit(sc) == C.msm_pippenger(cid, sc, key.download(0, m))).all()
    with pytest.raises(IOError, match="failed to fill whole buffer"):
        cm.CommitmentKey.load_from_file(cid, path, k + 1)
    # one flipped bit in the LAST chunk: only the validating load refuses it
    raw = np.memmap(path, dtype=np.uint8, mode="r+")
    raw[((1 << k) - 7) * 64 + 9] ^= 0x10
    raw.flush(); del raw
    cm.CommitmentKey.load_from_file(cid, path, k).close()
    with pytest.raises(IOError, match="Wrong file in cache, some ptr out of curve"):
        cm.CommitmentKey.load_from_file(cid, path, k, validate=True)
    key.close(); loaded.close()


def test_trim_returns_device_memory(gpu_lib):
    """mira_trim(0) after a large commit and transforms: the device's free memory is back where it was before the
    library grew its workspaces."""
    lib = gpu_lib
    lib.trim(0)
    n = 1 << 24
    key = cm.CommitmentKey.synthetic(0, n, seed=7)
    d = cm.synth_scalars_device(0, n, seed=8)
    lib.check(lib.c.mira_dev_sync())
    free0, total = lib.mem_info()
    from mira_amd import fft as F
    p1 = key.commit_device(d, n)
    F.fft_device(d, 24); F.ifft_device(d, 24)                   # two cached table sets of 0.8 GB each + the temporary
    free1, _ = lib.mem_info()
    grown = free0 - free1
    assert grown > (2 * 16 + 4 * 16) * n                        # at least the digit and sorted-entry buffers of that commit
    released = lib.trim(0)
    free2, _ = lib.mem_info()
    assert released >= grown * 0.95 and free0 - free2 < 64 << 20, (free0, free1, free2, released)
    p2 = key.commit_device(d, n)                                 # everything comes back on demand (d now holds ifft(fft(x)) = x)
    assert (p1 == p2).all()
    keep = 1 << 30
    lib.trim(keep)
    free3, _ = lib.mem_info()
    assert free0 - free3 <= keep + (64 << 20)
    lib.free(d); key.close(); lib.trim(0)


def test_window_width_per_handle_gpu(gpu_lib):
    import ctypes
    lib = gpu_lib
    n = 1 << 15
    keys = [cm.CommitmentKey.synthetic(cid, n, seed=50 + cid) for cid in (0, 1)]
    ds = [cm.synth_scalars_device(cid, n, seed=60 + cid) for cid in (0, 1)]
    want = [keys[i].commit_device(ds[i], n) for i in (0, 1)]
    keys[0].set_window_bits(8); keys[1].set_window_bits(14)
    for i, c in ((0, 8), (1, 14), (0, 8)):
        assert (keys[i].commit_device(ds[i], n) == want[i]).all()
        cc, ww = ctypes.c_int32(), ctypes.c_int32()
        lib.check(lib.c.mira_msm_last_plan(ctypes.byref(cc), ctypes.byref(ww)))
        assert cc.value == c
    for i in (0, 1):
        lib.free(ds[i]); keys[i].close()
