"""CPU: path_planner_amd/csrc/pp_cr.h — the double-double sin / cos / atan2 / acos the device's Dubins solver uses — compiled for the
host and compared with the x87 long-double functions rounded to double (11 more bits than a double: conclusive unless the value lies
within 2^-9 ulp of a midpoint) and with glibc's double functions, which is what the reference runs on."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cr_trig_is_correctly_rounded_and_agrees_with_glibc():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "cr_trig_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "cr_trig_check.cpp"), "-o", exe, "-lm"])
        out = subprocess.run([exe, "2000000"], capture_output=True, text=True, check=True).stdout
    print(out)
    for line in out.strip().splitlines():
        w = line.split()
        name, n, wrong, undecided, glibc = w[0], int(w[2]), int(w[4]), int(w[6]), int(w[8])
        assert wrong == 0, f"{name}: {wrong} of {n} values are not the correctly rounded ones"
        assert undecided < n // 200, (name, undecided)                 # ~1 / 256 of the values cannot be judged by 64-bit arithmetic
        assert glibc < n // 300, f"{name}: glibc differs in {glibc} of {n}"     # glibc 2.35: about one in a thousand is not correctly rounded
