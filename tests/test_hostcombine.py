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


def test_lazy29_domain_matches_canonical():
    """the MSM's signed 29-bit lazy arithmetic (field.h Fy, curve.h XYZZz: merged reductions, carry-free differences) against
    the canonical Montgomery implementation, on the host"""
    src = os.path.join(ROOT, "tests", "native", "lazy29_test.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "lazy29_test")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", src, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "lazy29: ok" in r.stdout, r.stdout + r.stderr


def test_host_side_under_sanitizers():
    """SURVEY section 5 (race / memory checking belongs on the CPU build): the host-side code of the boundary -- hostcombine.h,
    the shared field / curve headers as the host compiles them, trh.hpp's host arithmetic and Expression lowering -- built with
    -fsanitize=address,undefined and run; any report aborts the run (-fno-sanitize-recover)"""
    flags = ["-O1", "-g", "-std=c++17", "-w", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    with tempfile.TemporaryDirectory() as tmp:
        for name, inc, banner in (("hostcombine_test", [], "hostcombine: ok"), ("trh_hpp_host_test", ["-I" + os.path.join(ROOT, "include")], "trh.hpp host side: ok"),
                                  ("lazy29_test", [], "lazy29: ok"),
                                  # the quad-lane group law of csrc/curve_q4.h as a lane-by-lane CPU model (which lane multiplies what, and the
                                  # magnitude bounds of every intermediate) against curve.h's xyzzz_add / xyzzz_dbl
                                  ("q4_model_test", [], "q4 model: ok")):
            exe = os.path.join(tmp, name + "_san")
            subprocess.check_call(["g++", *flags, *inc, os.path.join(ROOT, "tests", "native", name + ".cpp"), "-o", exe])
            r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
            assert r.returncode == 0 and banner in r.stdout, r.stdout + r.stderr
            assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr


def test_device_block_pool_index():
    """csrc/devpool.h, the bookkeeping behind trh_malloc / trh_free, with two made-up device ids: a device gives up ITS largest idle
    block first (round 3 evicted the largest block of the highest-numbered device), classes stay per device, a block is live or idle"""
    src = os.path.join(ROOT, "tests", "native", "devpool_test.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "devpool_test")
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", src, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "devpool: ok" in r.stdout, r.stdout + r.stderr


def test_copy_pool_under_thread_sanitizer():
    """the host threads of the host-pointer staging path (csrc/copypool.h: spin-then-sleep workers handed slices through a generation
    counter): two callers sharing one pool and a third on another, byte-exact copies, no report from -fsanitize=thread; then the same
    under address + undefined"""
    src = os.path.join(ROOT, "tests", "native", "copypool_test.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        for tag, san in (("tsan", "-fsanitize=thread"), ("asan", "-fsanitize=address,undefined")):
            exe = os.path.join(tmp, "copypool_" + tag)
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-w", san, "-fno-omit-frame-pointer", src, "-o", exe, "-pthread"])
            # the test deletes its pools at the end (a context's pools die with it: the destructor stops and joins the workers)
            r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "copypool: ok" in r.stdout, r.stdout + r.stderr
            assert "WARNING: ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_host_helper_under_thread_sanitizer():
    """csrc/hosthelper.h (the thread that takes the second half of a batch's host Horners in msm_finish): jobs back to back and after the
    helper fell asleep, shutdown in both states, two helpers side by side -- no report from -fsanitize=thread, then address + undefined"""
    src = os.path.join(ROOT, "tests", "native", "hosthelper_test.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        for tag, san in (("tsan", "-fsanitize=thread"), ("asan", "-fsanitize=address,undefined")):
            exe = os.path.join(tmp, "hosthelper_" + tag)
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-w", san, "-fno-omit-frame-pointer", src, "-o", exe, "-pthread"])
            r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "hosthelper: ok" in r.stdout, r.stdout + r.stderr
            assert "WARNING: ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
