"""The cut-down Box-Muller functions of csrc/bmmath.hpp (the -2 log, sqrt and sin / cos of 2 pi u every normal of the RNG contract goes
through) against long-double libm: the header is compiled for the host with the hardware's reciprocal / reciprocal-square-root estimates
emulated at 22 good bits (tests/bmmath_check.c) and must stay within the error it documents, end points and quadrant boundaries included."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_box_muller_functions_against_long_double_libm(tmp_path):
    exe = str(tmp_path / "bmmath_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "bmmath_check.c"), "-lm"], check=True)
    out = subprocess.run([exe, "4000000"], check=True, capture_output=True, text=True).stdout
    m = re.search(r"-2log ([\d.]+)\s+sqrt ([\d.]+) \(differs from correctly rounded in (\d+)\)\s+sin ([\d.]+)\s+cos ([\d.]+)\s+normals ([\d.]+)", out)
    assert m, out
    lg, sq, nwrong, sn, cs, nz = float(m.group(1)), float(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))
    assert lg <= 1.0 and sq <= 0.501 and nwrong == 0 and sn <= 1.2 and cs <= 1.2 and nz <= 3.0, out
