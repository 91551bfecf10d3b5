"""Development probe: host -> device copy rates on the box, pageable and pinned, by chunk size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mira_amd import _lib
lib = _lib.load()
lib.check(lib.c.mira_init(0))
for mb in (1, 4, 16, 32, 128):
    n = mb << 20
    a = np.ones(n, dtype=np.uint8)
    d = lib.alloc(n)
    lib.upload(d, a)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); lib.upload(d, a); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    pin = torch.empty(n, dtype=torch.uint8).pin_memory()
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
    tp = []
    for _ in range(5):
        t0 = time.perf_counter(); dev.copy_(pin, non_blocking=True); torch.cuda.synchronize(); tp.append(time.perf_counter() - t0)
    tp = sorted(tp)[2]
    src = torch.from_numpy(a)
    tc = []
    for _ in range(5):
        t0 = time.perf_counter(); pin.copy_(src); tc.append(time.perf_counter() - t0)
    tc = sorted(tc)[2]
    print(f"{mb} MiB: pageable hipMemcpy {t * 1e3:.3f} ms ({n / t / 1e9:.1f} GB/s)  pinned {tp * 1e3:.3f} ms ({n / tp / 1e9:.1f} GB/s)  host memcpy into pinned {tc * 1e3:.3f} ms ({n / tc / 1e9:.1f} GB/s)", flush=True)
    lib.free(d)
