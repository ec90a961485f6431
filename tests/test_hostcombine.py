"""CPU test: the 4 x 64-bit host Horner / normalisation of csrc/hostcombine.h against the generic nine-limb implementation the
device code shares (curve.h), built with g++ -- no GPU, no HIP."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hostcombine_matches_generic_implementation():
    src = os.path.join(ROOT, "tests", "native", "hostcombine_test.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "hostcombine_test")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", src, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "hostcombine: ok" in r.stdout, r.stdout + r.stderr
