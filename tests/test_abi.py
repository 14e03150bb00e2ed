"""CPU-only checks of the C ABI: the shared library loads, exports every symbol include/spegnet_hip.h
declares, and the ctypes signatures in spegnet_amd/_lib.py agree with the header."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_header():
    src = open(os.path.join(ROOT, "include", "spegnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|long)\s+(spg_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        sig = ""
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or "spg_stream_t" in a:
                    sig += "p"
                elif re.match(r"(const\s+)?long\b", a):
                    sig += "l"
                elif re.match(r"(const\s+)?float\b", a):
                    sig += "f"
                elif re.match(r"(const\s+)?int\b", a):
                    sig += "i"
                else:
                    raise AssertionError(f"unparsed arg {a!r} in {name}")
        decls[name] = sig
    return decls


def test_library_builds_and_exports_header_symbols():
    import __graft_entry__ as g
    g.build()
    from spegnet_amd import _lib
    lib = _lib.load()
    assert lib.spg_version() == _lib.ABI_VERSION
    decls = parse_header()
    assert len(decls) >= 30
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_signatures_match_header():
    from spegnet_amd import _lib
    decls = parse_header()
    table = dict(_lib.SIGNATURES)
    table.update(_lib._OPTIONAL)
    table.update({k: v[1] for k, v in _lib.QUERIES.items()})
    for name, sig in decls.items():
        if name == "spg_version":
            continue
        assert name in table, f"{name} has no ctypes signature"
        assert table[name] == sig, f"{name}: ctypes {table[name]} != header {sig}"
    for name in list(_lib.SIGNATURES) + list(_lib.QUERIES):
        assert name in decls, f"{name} bound in _lib.py but not declared in the header"


def test_bad_arguments_fail_loudly():
    from spegnet_amd import _lib
    with pytest.raises(RuntimeError, match="K=.*multiple"):
        _lib.call("spg_gemm_nt", _lib.SPG_BF16, None, None, None, None, None, None, None, 8, 8, 7, 8, 8, 0, 0, 0, 0, 0, 0, 0, None)
    with pytest.raises(RuntimeError, match="unsupported head_dim"):
        _lib.call("spg_attn_fwd", _lib.SPG_F32, None, None, None, None, None, 1, 8, 8, 1, 24, 8, None)
