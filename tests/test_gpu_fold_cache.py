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
