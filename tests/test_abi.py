"""CPU checks of the drop-in boundary: libcrgpu.so loads, exports every symbol
include/crgpu.h declares, the ctypes structs match the header's layout, and
without a GPU the library fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from cockroach_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "crgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/crgpu.h but not exported"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(declared) == bound, f"ctypes table out of sync: {set(declared) ^ bound}"


def test_struct_layouts_match_header(tmp_path):
    """sizeof() of every boundary struct as gcc lays it out == the ctypes mirror."""
    import subprocess
    names = [("crgpu_dcol", _lib.DCol), ("crgpu_dev_result", _lib.DevResult), ("crgpu_join_spec", _lib.JoinSpec),
             ("crgpu_agg_desc", _lib.AggDesc), ("crgpu_agg_spec", _lib.AggSpec), ("crgpu_col", _lib.Col),
             ("crgpu_batch", _lib.Batch), ("crgpu_out_col", _lib.OutCol), ("crgpu_out_batch", _lib.OutBatch)]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "crgpu.h"\nint main(){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n, _ in names) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    for (n, ct), sz in zip(names, sizes):
        assert C.sizeof(ct) == sz, n


def test_no_silent_cpu_fallback():
    lib = _lib.load()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-device path cannot be exercised")
    assert lib.crgpu_init(0) == _lib.ERR_NO_DEVICE
    assert b"no HIP device" in lib.crgpu_last_error()
    p = C.c_void_p()
    assert lib.crgpu_dev_malloc(C.byref(p), 16) == _lib.ERR_NO_DEVICE
    spec = _lib.JoinSpec()
    h = C.c_void_p()
    assert lib.crgpu_dev_join_create(C.byref(spec), C.byref(h)) != _lib.OK
    assert lib.crgpu_hj_new(C.byref(spec), C.byref(h)) != _lib.OK


def test_product_does_not_import_the_oracle():
    # the oracle is test infrastructure: nothing under cockroach_amd/, include/, host/ may reference it
    bad = []
    for base in ("cockroach_amd", "include", "host"):
        for dirpath, _d, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".c")):
                    t = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"crdb_oracle|from oracle|import oracle|ora_hash|ora_memhash", t):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
