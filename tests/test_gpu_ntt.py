"""GPU parity tests for src/fft.rs: fft / ifft / coset_fft / coset_ifft / best_fft, bit-exact
against the CPU oracle and the reference's known-answer vector."""
import numpy as np
import pytest

from helpers import ints_to_mont, load_golden, mont_to_ints
from mira_amd import fft as F
from oracle import cref as C
from oracle import pyref as P

pytestmark = pytest.mark.gpu


def test_fft_simple_input_kat(gpu_lib):
    k = load_golden("ref_kats.json")["fft_simple_input_test"]      # src/fft.rs:238-257
    out = F.fft(ints_to_mont(k["input"], P.R_MOD), k["log_n"])
    assert mont_to_ints(out, P.R_MOD) == [int(v) for v in k["output_decimal"]]


def test_fft_random_input_roundtrip(gpu_lib):
    for k in (4, 5, 6, 7, 8):                                        # src/fft.rs:265-279
        a = np.repeat(C.synth_scalars(0, 1, seed=k), 1 << k, axis=0)
        assert (F.ifft(F.fft(a, k), k) == a).all()


@pytest.mark.parametrize("k", [4, 10])
def test_golden_vectors(gpu_lib, k):
    g = load_golden("ntt_vectors.json")[str(k)]
    a = C.synth_scalars(0, 1 << k, seed=g["seed"])
    assert mont_to_ints(F.fft(a, k), P.R_MOD) == [int(v, 16) for v in g["fft"]]
    assert mont_to_ints(F.ifft(a, k), P.R_MOD) == [int(v, 16) for v in g["ifft"]]
    assert mont_to_ints(F.coset_fft(a), P.R_MOD) == [int(v, 16) for v in g["coset_fft"]]
    assert mont_to_ints(F.coset_ifft(a), P.R_MOD) == [int(v, 16) for v in g["coset_ifft"]]


@pytest.mark.parametrize("k", [0, 1, 2, 5, 11, 12, 13, 16, 17, 20, 22])
def test_parity_vs_oracle(gpu_lib, k):
    a = C.synth_scalars(0, 1 << k, seed=300 + k)
    assert (F.fft(a, k) == C.fft(a, k)).all()
    assert (F.ifft(a, k) == C.ifft(a, k)).all()
    w = C.get_omega_or_inv(k, True)
    assert (F.best_fft(a, w, k) == C.best_fft(a, w, k)).all()


def test_omega_derivation(gpu_lib):
    for k in range(0, 29):
        assert (F.get_omega_or_inv(k, False) == C.get_omega_or_inv(k, False)).all()
    assert (F.get_omega_or_inv(5, True) == C.get_omega_or_inv(5, True)).all()


def test_full_size_2p24_properties(gpu_lib):
    """BASELINE config 2 (2^24 coefficients): bit-exact vs the oracle, plus the size-independent
    properties ifft(fft(x)) == x and linearity."""
    k = 24
    n = 1 << k
    a = C.synth_scalars(0, n, seed=77)
    fa = F.fft(a, k)
    assert (F.ifft(fa, k) == a).all()
    assert (fa == C.fft(a, k)).all()
    # fft(e_1) = powers of omega: spot-check entries against the oracle's omega
    e1 = np.zeros((n, 4), dtype=np.uint64)
    e1[1] = C.to_mont(C.FIELD_FR, np.array([1, 0, 0, 0], dtype=np.uint64))[0]
    fe = F.fft(e1, k)
    w = C.get_omega_or_inv(k, False)
    assert (fe[0] == e1[1]).all() and (fe[1] == w).all()
    assert (fe[2] == C.f_mul(C.FIELD_FR, w, w)).all()


@pytest.mark.parametrize("k,grid", [(23, -1), (20, 100), (20, 5), (22, 37)])
def test_block_groups_from_counters(gpu_lib, k, grid):
    """The work distribution of k_ntt_wave (ntt_kernels.cuh: nttw_grab) under real concurrency: 2^23 on the default grid
    (8 192 block-groups of 256- and 128-point lines for 768 workgroups), and grids that are no multiple of the eight ranges --
    100 and 37 workgroups (uneven homes), 5 (three ranges nobody is at home in) -- bit-exact against the oracle."""
    from mira_amd import _lib
    a = C.synth_scalars(0, 1 << k, seed=900 + k)
    want = C.fft(a, k)
    gpu_lib.tune(_lib.TUNE_NTT_GRID, grid)
    try:
        assert (F.fft(a, k) == want).all()
        assert (F.ifft(want, k) == a).all()
    finally:
        gpu_lib.tune(_lib.TUNE_NTT_GRID, -1)


def test_single_line_kernel_up_to_4096_points(gpu_lib):
    """2^10 .. 2^12 points take two passes by default; the single line (one workgroup, up to 128 KiB of LDS) stays reachable."""
    from mira_amd import _lib
    gpu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, 12)
    try:
        for k in (10, 11, 12):
            a = C.synth_scalars(0, 1 << k, seed=1200 + k)
            assert (F.fft(a, k) == C.fft(a, k)).all(), k
            assert (F.ifft(a, k) == C.ifft(a, k)).all(), k
    finally:
        gpu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, -1)


def test_three_pass_2p25_vs_oracle(gpu_lib):
    """log_n 25..28 take three passes of lines (n = n1 n2 n3): 2^25 bit-exact against the oracle,
    forward and inverse."""
    k = 25
    a = C.synth_scalars(0, 1 << k, seed=91)
    fa = F.fft(a, k)
    assert (fa == C.fft(a, k)).all()
    assert (F.ifft(fa, k) == a).all()


@pytest.mark.parametrize("k", [3, 8, 13, 15, 18, 21])
def test_wave_and_workgroup_kernels_agree(gpu_lib, k):
    """The wave-level kernel (default for these sizes) and the workgroup-level kernel give the same
    transform, both equal to the oracle."""
    from mira_amd import _lib
    a = C.synth_scalars(0, 1 << k, seed=500 + k)
    want = C.fft(a, k)
    assert (F.fft(a, k) == want).all()                       # the kernel the size policy picks
    try:
        for mode in (1, 0):                                   # wave-level wherever possible, then never
            gpu_lib.tune(_lib.TUNE_NTT_WAVE, mode)
            assert (F.fft(a, k) == want).all(), mode
            assert (F.ifft(want, k) == a).all(), mode
    finally:
        gpu_lib.tune(_lib.TUNE_NTT_WAVE, -1)
