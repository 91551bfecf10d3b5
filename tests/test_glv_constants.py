"""The constants of the GLV split (mira_amd/csrc/glv_consts.h) are the ones tools/glv_constants.py derives and checks against
the oracle's curve arithmetic: beta and lambda with phi(G) = lambda G, a reduced lattice basis, the rounding constants of the
division-free decomposition -- and k = k1 + k2 lambda, k Q = k1 Q + k2 phi(Q), |k1|, |k2| < 2^126 on 2 010 scalars per curve."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_holds_the_derived_constants():
    tool = os.path.join(ROOT, "tools", "glv_constants.py")
    out = subprocess.run([sys.executable, tool, "--cpp"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    header = open(os.path.join(ROOT, "mira_amd", "csrc", "glv_consts.h")).read()
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 12                                    # two specialisations of six lines
    for line in lines:
        assert line in header, line
    checked = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert checked.returncode == 0 and checked.stdout.count("hold") == 2, checked.stdout + checked.stderr[-2000:]
