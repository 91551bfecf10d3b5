"""Development probe: the 2^k NTT of several library builds on ONE box, alternating, with a digest of the result.

usage: python tools/ntt_variants.py [k] lib1.so lib2.so ...      (child mode: MIRA_PROBE_LIB set, prints one line)
Boxes of the pool differ by ~5 % on ALU-bound kernels, so variants are only comparable inside one call; every variant runs
`rounds` times in rotation and the minimum and median of its per-pass kernel times are printed.
"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(k):
    sys.path.insert(0, ROOT)
    from mira_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
    from mira_amd import commitment as cm, fft as F
    lib = _lib.load()
    n = 1 << k
    d = cm.synth_scalars_device(0, n, seed=5)
    F.fft_device(d, k)
    out = lib.download(d, (n, 4))
    digest = hashlib.sha256(out.tobytes()).hexdigest()[:16]
    lib.check(lib.c.mira_set_timing(1))
    reps = int(os.environ.get("MIRA_PROBE_REPS", "40"))
    for _ in range(10):                                  # let the clock settle under the load before anything is counted
        F.fft_device(d, k)
    tot = []
    acc = {}
    for _ in range(reps):
        F.fft_device(d, k)
        t = dict(lib.timings())
        tot.append(sum(v for a, v in t.items() if a.startswith("ntt_")))
        for a, v in t.items():
            acc.setdefault(a, []).append(v)
    print(json.dumps({"digest": digest, "min": min(tot), "med": sorted(tot)[len(tot) // 2],
                      "passes": {a: round(min(v), 4) for a, v in acc.items() if a.startswith("ntt_")}}))


def main():
    args = sys.argv[1:]
    k = 24
    if args and args[0].isdigit():
        k = int(args.pop(0))
    if os.environ.get("MIRA_PROBE_CHILD"):
        return child(k)
    rounds = int(os.environ.get("MIRA_PROBE_ROUNDS", "3"))
    res = {a: [] for a in args}
    for r in range(rounds):
        for a in args:
            env = dict(os.environ, MIRA_PROBE_LIB=a, MIRA_PROBE_CHILD="1")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not line:
                print("FAILED", a, p.returncode, p.stderr[-400:], flush=True)
                res[a].append(None)
                continue
            res[a].append(json.loads(line[-1]))
    ref = None
    for a in args:
        ok = [x for x in res[a] if x]
        if not ok:
            print("%-40s  no result" % os.path.basename(a)); continue
        ref = ref or ok[0]["digest"]
        print("%-40s  min %.4f  med %.4f ms  passes %s  digest %s%s" % (
            os.path.basename(a), min(x["min"] for x in ok), sorted(x["med"] for x in ok)[len(ok) // 2],
            ok[0]["passes"], ok[0]["digest"], "" if ok[0]["digest"] == ref else "  << DIFFERS"), flush=True)


if __name__ == "__main__":
    main()
