"""Generates tests/golden/*.json with the Python-integer oracle (oracle/pyref.py).

The reference (Rust; no cargo here, arithmetic in un-vendored git dependencies) cannot be run in
this container, so apart from the reference's own known-answer vectors (copied as data into
ref_kats.json with their file:line) these vectors come from the oracle restatement.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref as P  # noqa: E402


def hx(v):
    return hex(v)


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def main():
    # ---- the reference's own vectors (data only) ------------------------------------------
    ref = {
        "fft_simple_input_test": {  # /root/reference/src/fft.rs:238-257
            "source": "src/fft.rs:240-249", "log_n": 3, "input": list(range(8)),
            "output_decimal": [
                "28",
                "68918385373930674424918168212551896122229959265833979749191472831399925654",
                "17631683881184975370165255887551781615748388533673675138856",
                "68918385373930639161550405842601155791718184162270748252414405484049647934",
                "21888242871839275222246405745257275088548364400416034343698204186575808495613",
                "21819324486465344583084855339414673932756646216253763595445789781091758847675",
                "21888242871839275204614721864072299718383108512864252727949815652902133356753",
                "21819324486465344547821487577044723192426134441150200363949012713744408569955"]},
        "fr_modulus_minus_one": {  # /root/reference/src/digest.rs:101-105
            "source": "src/digest.rs:103",
            "decimal": "21888242871839275222246405745257275088548364400416034343698204186575808495616"},
        "g1_scalar_mul": {  # /root/reference/src/digest.rs:98-113: (r-1)*G == -G
            "source": "src/digest.rs:98-113", "statement": "(r-1) * G1::generator() == -G1::generator()"},
        "bn254_g2_generator": {  # /root/reference/src/gadgets/ecc2.rs:156-180
            "source": "src/gadgets/ecc2.rs:159-176",
            "x": ["10857046999023057135944570762232829481370756359578518086990519993285655852781",
                  "11559732032986387107991004021392285783925812861821192530917403151452391805634"],
            "y": ["8495653923123431417604973247489272438418190587263600148770280649306958101930",
                  "4082367875863433681332203403145435568316851327593401208105741076214120093531"]},
    }
    # ref_kats.json also holds reference vectors entered by hand as data (basic_lagrange_test; the `Display` strings of
    # src/polynomial/expression.rs:528-606 and src/polynomial/grouped_poly.rs:287-461): keep them
    path = os.path.join(HERE, "ref_kats.json")
    kats = json.load(open(path)) if os.path.exists(path) else {}
    kats.update(ref)
    json.dump(kats, open(path, "w"), indent=1)

    # ---- field vectors ------------------------------------------------------------------------
    fields = {}
    for name, mod in (("fq", P.P_MOD), ("fr", P.R_MOD)):
        rows = []
        vals = [0, 1, 2, mod - 1, mod - 2, (1 << 253) % mod] + [P.synth_scalar(i, mod, seed=99) for i in range(6)]
        for i, a in enumerate(vals):
            b = vals[(i * 5 + 3) % len(vals)]
            rows.append({"a": hx(a), "b": hx(b), "add": hx((a + b) % mod), "sub": hx((a - b) % mod), "mul": hx(a * b % mod),
                         "inv_a": hx(pow(a, -1, mod) if a else 0), "a_mont": hx(P.to_mont(a, mod))})
        fields[name] = {"modulus": hx(mod), "rows": rows}
    json.dump(fields, open(os.path.join(HERE, "field_vectors.json"), "w"), indent=1)

    # ---- NTT vectors ----------------------------------------------------------------------------
    ntt = {}
    for k in (4, 10):
        a = P.synth_scalars(1 << k, P.R_MOD, seed=1000 + k)
        f = list(a); P.fft(f, k)
        g = list(a); P.ifft(g, k)
        c = list(a); P.coset_fft(c)
        d = list(a); d = P.coset_ifft(d)
        ntt[str(k)] = {"seed": 1000 + k, "fft": [hx(v) for v in f], "ifft": [hx(v) for v in g],
                       "coset_fft": [hx(v) for v in c], "coset_ifft": [hx(v) for v in d]}
    json.dump(ntt, open(os.path.join(HERE, "ntt_vectors.json"), "w"), indent=1)

    # ---- MSM vectors: n in {1, 2, 33, 1000}, both curves, with the edge cases of SURVEY 8(c) ----
    msm = {}
    for cid in (P.CURVE_BN256, P.CURVE_GRUMPKIN):
        cv = P.CURVES[cid]
        cases = []
        for n in (1, 2, 33, 1000):
            sc = P.synth_scalars(n, cv.r, seed=77 + n)
            bs = P.synth_bases(n, cv, seed=55 + n)
            edits = {}
            if n >= 33:
                sc[0] = 0; sc[1] = cv.r - 1; sc[2] = 1; sc[3] = sc[4]          # zero, r-1, one, equal scalars
                bs[6] = bs[5]                                                      # duplicate base
                bs[7] = None                                                       # identity base
                bs[9] = P.ec_neg(bs[8], cv); sc[9] = sc[8]                         # P and -P with equal scalars
                edits = {"scalars": {"0": hx(0), "1": hx(cv.r - 1), "2": hx(1), "3": hx(sc[4]), "9": hx(sc[8])},
                         "bases": {"6": "copy of 5", "7": "identity", "9": "negation of 8"}}
            out = P.msm_naive(sc, bs, cv)
            cases.append({"n": n, "scalar_seed": 77 + n, "base_seed": 55 + n, "edits": edits, "result": pt(out)})
        msm[str(cid)] = cases
    json.dump(msm, open(os.path.join(HERE, "msm_vectors.json"), "w"), indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
