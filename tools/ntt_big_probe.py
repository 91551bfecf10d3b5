"""Development probe: the largest transforms (three-pass schedule, log_n 25..28) on device-resident
data: timing, ifft(fft(x)) == x on a sample, and fft(e_1) = powers of omega."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, commitment as cm, fft as F
from oracle import cref as C
lib = _lib.load()
for k in (25, 26, 28):
    n = 1 << k
    d = cm.synth_scalars_device(0, n, seed=5)
    head = lib.download(d, (4096, 4)); tail = lib.download(d + (n - 4096) * 32, (4096, 4))
    F.fft_device(d, k)
    t0 = time.perf_counter(); F.fft_device(d, k); dt = time.perf_counter() - t0   # second transform of the data: timing only
    F.ifft_device(d, k); F.ifft_device(d, k)
    ok = (lib.download(d, (4096, 4)) == head).all() and (lib.download(d + (n - 4096) * 32, (4096, 4)) == tail).all()
    # e_1 -> omega^k
    lib.upload(d, np.zeros((4096, 4), dtype=np.uint64))
    import ctypes
    one = C.to_mont(C.FIELD_FR, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    # zero the whole buffer by transforming zeros is wasteful: regenerate via synth of kind "zero" is not available; use memset through upload in chunks
    z = np.zeros((1 << 20, 4), dtype=np.uint64)
    for off in range(0, n, 1 << 20):
        lib.upload(d + off * 32, z)
    lib.upload(d + 32, one)
    F.fft_device(d, k)
    w = C.get_omega_or_inv(k, False)
    got = lib.download(d, (4, 4))
    w2 = C.f_mul(C.FIELD_FR, w, w); w3 = C.f_mul(C.FIELD_FR, w2, w)
    last = lib.download(d + (n - 1) * 32, (1, 4))[0]
    winv = C.get_omega_or_inv(k, True)
    e1_ok = (got[0] == one[0]).all() and (got[1] == w).all() and (got[2] == w2).all() and (got[3] == w3).all() and (last == winv).all()
    print(f"2^{k}: fft {dt * 1e3:.2f} ms = {n / dt / 1e6:.0f} M elements/s; ifft(ifft(fft(fft x))) == x on both ends: {ok}; fft(e_1) = omega^k at k = 0..3, n-1: {e1_ok}", flush=True)
    lib.free(d)
