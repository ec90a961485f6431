"""CPU tests: the C-ABI library loads and exports every symbol include/trh.h declares; the
host-side pieces that need no GPU (argument checks, point combine) behave; without a GPU the
compute entry points fail loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import pasta as o
from tiny_ram_halo2_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "trh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = api.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/trh.h but not exported by libtrh.so"
    # and the Python mirror binds exactly the declared set
    assert sorted(api.EXPORTED_SYMBOLS) == declared


def test_version_and_error_string():
    lib = api.lib()
    assert b"gfx950" in lib.trh_version()
    assert isinstance(lib.trh_last_error(), bytes)


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly():
    with pytest.raises(api.TrhError):
        api.init(0)
    with pytest.raises(api.TrhError):
        api.best_fft("fp", np.zeros((2, 4), np.uint64), np.zeros(4, np.uint64), 1)
    with pytest.raises(api.TrhError):
        api.best_multiexp("pallas", np.zeros((1, 4), np.uint64), np.zeros((1, 8), np.uint64))


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_device_host_pointer_entries_fail_loudly():
    """round 3's host-pointer entries (batch FFT, batched commitments, page-locking helpers, io stats): TRH_ENODEV without a GPU,
    never a silent CPU path"""
    import ctypes
    lib = api.lib()
    col = np.zeros((4, 4), np.uint64)
    with pytest.raises(api.TrhError):
        api.best_fft_batch("fp", [col], np.zeros(4, np.uint64), 2)
    p = ctypes.c_void_p()
    assert lib.trh_host_alloc(ctypes.byref(p), 4096) == -2 and b"trh_init" in lib.trh_last_error()
    assert lib.trh_host_register(col.ctypes.data_as(ctypes.c_void_p), col.nbytes) == -2
    st = api.IoStats()
    assert lib.trh_io_stats(ctypes.byref(st), 0) == -2
    assert lib.trh_ctx_stream(None) is None and lib.trh_group_peer_access() in (0, 1)


def test_best_multiexp_length_mismatch_panics_like_reference():
    with pytest.raises(AssertionError):
        api.best_multiexp("pallas", np.zeros((3, 4), np.uint64), np.zeros((2, 8), np.uint64))
    with pytest.raises(AssertionError):
        api.best_fft("fp", np.zeros((3, 4), np.uint64), np.zeros(4, np.uint64), 2)


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
def test_point_sum_host_combine(curve):
    """trh_point_sum (host-side combine of per-GPU partial MSM results) vs the big-int oracle."""
    import random
    rng = random.Random(11)
    cv = o.CURVES[curve]
    f = cv.base
    pts = [cv.mul(rng.randrange(cv.scalar.m), cv.generator) for _ in range(9)]
    pts[2] = None
    pts[4] = pts[3]
    pts[6] = cv.neg(pts[5])
    rows = []
    for p in pts:
        if p is None:
            rows.append([0] * 12)
            continue
        z = rng.randrange(1, f.m)
        rows.append(f.limbs(p[0] * z * z) + f.limbs(p[1] * z ** 3) + f.limbs(z))
    got = api.point_sum(curve, np.array(rows, np.uint64))
    want = None
    for p in pts:
        want = cv.add(want, p)
    assert cv.affine_from_limbs(got[:8]) == want
    assert [int(v) for v in got[8:]] == f.limbs(1)
    assert (api.point_sum(curve, np.array(rows[5:7], np.uint64)) == 0).all()
    assert (api.point_sum(curve, np.zeros((0, 12), np.uint64)) == 0).all()


# ---- INTEGRATION.md's Rust binding vs include/trh.h -------------------------------------------------------------

def _split_args(s):
    """top-level comma split (function-pointer arguments carry their own parentheses)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def _c_prototypes():
    """name -> (return class, [argument classes]) from include/trh.h; classes: 'ptr', 'int' (any integer scalar), 'void'"""
    text = open(os.path.join(ROOT, "include", "trh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    i = 0
    for m in re.finditer(r"\b(trh_[a-z0-9_]+)\s*\(", text):
        name = m.group(1)
        # the return type: what stands between the previous ';' / '}' / '{' and the name
        start = max(text.rfind(";", 0, m.start()), text.rfind("}", 0, m.start()), text.rfind("{", 0, m.start())) + 1
        ret = text[start:m.start()].strip()
        if not ret or "typedef" in ret or "(" in ret:
            continue  # a function-pointer typedef or a use inside another declaration
        depth, j = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        args = _split_args(text[m.end():j - 1])
        if args == ["void"]:
            args = []
        cls = lambda d: "ptr" if ("*" in d or "[" in d or "(" in d or re.search(r"\btrh_\w+_t\b|\btrh_\w+_fn\b|\btrh_\w+_cb\b", d)) else "int"
        protos[name] = ("void" if ret == "void" else cls(ret + " "), [cls(a) for a in args])
    return protos


def _rust_prototypes():
    """every `fn trh_*` inside an extern "C" block of INTEGRATION.md"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    protos = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        body = re.sub(r"//[^\n]*", "", blk.group(1))
        for m in re.finditer(r"fn\s+(trh_[a-z0-9_]+)\s*\(", body):
            depth, j = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(body[j], 0)
                j += 1
            args = _split_args(body[m.end():j - 1])
            tail = body[j:body.index(";", j)]
            ret = tail.split("->")[1].strip() if "->" in tail else ""
            rcls = lambda t: "ptr" if (t.startswith("*") or "extern" in t or "fn(" in t or t.startswith("Option<")) else "int"
            protos.setdefault(name := m.group(1), ("void" if not ret else rcls(ret), [rcls(a.split(":", 1)[1].strip()) for a in args]))
    return protos


def test_integration_md_rust_block_matches_the_header():
    """the source-only Rust shim of INTEGRATION.md declares libtrh's entry points by hand; nothing compiles it here (no Rust toolchain), so
    this test is its only guard: every `fn trh_*` of every extern "C" block exists in include/trh.h with the same number of arguments,
    pointer vs integer scalar argument by argument, and the same kind of return value"""
    c, rust = _c_prototypes(), _rust_prototypes()
    assert len(rust) >= 30, sorted(rust)
    # the header parser sees what the symbol test sees
    assert set(_declared_symbols()) == set(c) and len(c) >= 90
    for name, (rret, rargs) in sorted(rust.items()):
        assert name in c, f"INTEGRATION.md declares {name}, include/trh.h does not"
        cret, cargs = c[name]
        assert len(rargs) == len(cargs), f"{name}: {len(rargs)} arguments in INTEGRATION.md, {len(cargs)} in trh.h"
        assert rargs == cargs, f"{name}: argument kinds {rargs} in INTEGRATION.md vs {cargs} in trh.h"
        assert rret == cret, f"{name}: returns {rret} in INTEGRATION.md, {cret} in trh.h"


OPTIONS_SCRIPT = r"""
from tiny_ram_halo2_amd import api
lib = api.lib()
assert api.get_option("pool_mb") == 7 and api.get_option("reduce_q4") == 0      # the environment, read once
assert api.get_option("stage_slot_mb") == 16                                     # 9999 is out of range: the default stays
api.set_option("pool_mb", 123)
assert api.get_option("pool_mb") == 123                                          # an explicit call overrides the environment
import os
os.environ["TRH_POOL_MB"] = "55"                                                 # not read again
assert api.get_option("pool_mb") == 123
for name, value in (("no_such_option", "1"), ("bin_sort", "2"), ("bin_sort", "x"), ("copy_threads", "65"), ("trace", "")):
    assert lib.trh_set_option(name.encode(), value.encode()) == -1, (name, value)
    assert name.encode() in lib.trh_last_error()
assert lib.trh_set_option(None, b"1") == -1
names = ["pool_mb", "stage_slot_mb", "copy_threads", "bases_cache", "force_no_peer", "ipa_fold", "trace", "msm_chunk_gb", "sparse", "reduce_q4", "bin_sort", "selftest"]
for n in names:
    api.get_option(n)
print("options ok", len(names))
"""


def test_options_are_parsed_once_and_settable_without_a_device():
    """trh_set_option / trh_get_option (include/trh.h): TRH_<NAME> is read once, an explicit call overrides it, unknown names and values
    out of range are TRH_EINVAL with the name in trh_last_error(); no device is needed.  (That options are fixed while a context exists
    is a -m gpu test: tests/test_gpu_native.py.)"""
    from common import run_with_options
    out = run_with_options(OPTIONS_SCRIPT, {"TRH_POOL_MB": "7", "TRH_REDUCE_Q4": "0", "TRH_STAGE_SLOT_MB": "9999"})
    assert "options ok 12" in out


def test_no_getenv_outside_the_option_parser():
    """the library reads the environment in ONE place (capi.hip opt_load_env): a getenv on a call path would race a setenv elsewhere in a
    multi-threaded host"""
    import glob
    import re
    hits = []
    for f in sorted(glob.glob(os.path.join(ROOT, "tiny-ram-halo2_amd", "csrc", "*"))):
        if not f.endswith((".hip", ".h")):
            continue
        for i, line in enumerate(open(f), 1):
            code = line.split("//")[0]
            if re.search(r"\bgetenv\s*\(", code):
                hits.append((os.path.basename(f), i))
    assert hits == [h for h in hits if h[0] == "capi.hip"] and len(hits) == 1, hits
