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


def test_trh_hpp_host_side():
    """include/trh.hpp without a device: its host field arithmetic against hostcombine.h, and the Expression lowering against
    an interpreter of the stack program (random trees, every node type, shared sub-expressions)"""
    src = os.path.join(ROOT, "tests", "native", "trh_hpp_host_test.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "trh_hpp_host_test")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "trh.hpp host side: ok" in r.stdout, r.stdout + r.stderr
