"""CPU-side checks of the C ABI: the in-tree library builds, loads, and exports
every symbol include/als_hip.h declares.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from collaborative_filtering_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        ge.build()
    return _hip.load()


def test_header_symbols_are_exported(lib):
    from collaborative_filtering_amd import _hip
    text = open(os.path.join(ROOT, "include", "als_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|size_t)\s+(als_\w+)\s*\(", text, flags=re.M))
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)


def test_host_side_queries(lib):
    from collaborative_filtering_amd import layout
    assert lib.als_version() == 103
    for k in (1, 15, 16, 17, 50, 64, 128, 160):
        ld = lib.als_padded_k(k)
        assert ld == layout.padded_k(k)
        perm = [lib.als_perm_index(k, c) for c in range(ld)]
        assert sorted(perm) == list(range(ld))
        assert perm == list(layout.perm_of_col(k))
        kb = ld // 16
        assert lib.als_partial_slot_bytes(k) == (kb * (kb + 1) // 2 * 4 + 2 * kb + 2) * 64 * 4
    assert lib.als_padded_k(0) == -2 and lib.als_padded_k(161) == -2


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirrors of the parameter structs have exactly the C layout: gcc
    compiles the header and prints sizeof / offsetof of every field."""
    import ctypes as C
    import subprocess
    from collaborative_filtering_amd import _hip
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "als_hip.h"', 'int main(void){']
    structs = (("als_row_solve_params", _hip.RowSolveParams), ("als_gs_sweep_params", _hip.GsSweepParams),
               ("als_w_params", _hip.WParams))
    for cname, st in structs:
        src.append(f'printf("%zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            src.append(f'printf("%zu\\n", offsetof({cname}, {fname}));')
    src.append('printf("%zu %zu\\n", sizeof(als_task), sizeof(als_long_row)); return 0;}')
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    vals = iter(int(x) for x in out)
    for _, st in structs:
        assert C.sizeof(st) == next(vals)
        for fname, _ in st._fields_:
            assert getattr(st, fname).offset == next(vals), (st.__name__, fname)
    assert next(vals) == 16 and next(vals) == 16


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from collaborative_filtering_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_hip.HipLibraryMissing):
        _hip.load()


def test_backend_refuses_cpu_device():
    import torch
    from collaborative_filtering_amd.backend import HipBackend
    with pytest.raises(RuntimeError):
        HipBackend(torch.device("cpu"))
