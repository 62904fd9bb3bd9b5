"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports exactly the header's symbols."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dfol_vqa.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfol_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from dfol_vqa_amd import _lib
    return _lib


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 18
    handle = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "libdfolvqa.so does not export %s declared in include/dfol_vqa.h" % n


def test_binding_table_matches_header(lib):
    declared = set(declared_symbols()) - {"dfol_abi_version", "dfol_last_error"}
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    # argument counts agree with the header prototypes
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, argtypes in lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]) == len(argtypes), name


def test_abi_version_and_error_channel(lib):
    h = lib.load()
    assert h.dfol_abi_version() == 3
    # argument errors are reported without touching a device (no GPU needed): NS not a multiple of 4
    rc = h.dfol_filter_fwd_f32(None, None, None, None, None, 0, None, 3, 6, None, None)
    assert rc != 0 and b"filter_fwd" in h.dfol_last_error()


def test_no_cpu_fallback(lib):
    import torch
    with pytest.raises(lib.DfolError):
        lib.quantify_fwd(torch.zeros(2, 4), torch.ones(2), torch.zeros(2, dtype=torch.int32), torch.ones(2, dtype=torch.int32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dfol_vqa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "dfol_oracle" not in src, f
