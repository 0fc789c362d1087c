"""CPU: the C-ABI library loads, exports every symbol include/gaviko_hip.h declares, and the ctypes structs in
gaviko_amd/lib.py have exactly the header's field order (no compute calls -- no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "gaviko_hip.h")


def _header():
    return open(HEADER).read()


def _strip_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def test_library_exports_every_declared_symbol():
    from gaviko_amd import lib
    l = lib.load()
    src = _strip_comments(_header())
    decls = re.findall(r"\b(gvk_[a-z0-9_]+)\s*\(", src)
    assert len(decls) >= 20
    bound = set(lib.SIGNATURES) | set(lib.NO_STREAM)
    for name in sorted(set(decls)):
        assert hasattr(l, name), f"{name} declared in gaviko_hip.h but not exported by libgaviko_hip.so"
        assert name in bound, f"{name} has no ctypes signature in gaviko_amd/lib.py"
    assert l.gvk_abi_version() == lib.ABI_VERSION == 12


def test_ctypes_structs_match_header_field_order():
    from gaviko_amd import lib
    src = _strip_comments(_header())
    for cname, cls in lib.STRUCTS.items():
        m = re.search(r"typedef struct " + cname + r"\s*\{(.*?)\}\s*" + cname + r"\s*;", src, flags=re.S)
        assert m, cname
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            typ = re.match(r"(const\s+)?(void|float|int32_t|uint64_t|int64_t)\s*(\*?)", stmt)
            names = re.sub(r"^(const\s+)?(void|float|int32_t|uint64_t|int64_t)", "", stmt)
            for nm in names.split(","):
                nm = nm.strip()
                is_ptr = nm.startswith("*") or typ.group(3) == "*" and nm == names.split(",")[0].strip()
                fields.append((nm.lstrip("* ").strip(), "ptr" if "*" in nm or (typ.group(3) == "*" and nm == names.split(",")[0].strip()) else typ.group(2)))
        want = []
        for name, ct in cls._fields_:
            kind = {ctypes.c_void_p: "ptr", ctypes.c_int32: "int32_t", ctypes.c_float: "float", ctypes.c_uint64: "uint64_t", ctypes.c_int64: "int64_t"}[ct]
            want.append((name, kind))
        assert fields == want, f"{cname}: header {fields} != ctypes {want}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gaviko_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.GavikoHipError):
        lib.load()


def test_no_device_fails_loudly():
    import torch
    from gaviko_amd import lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.GavikoHipError):
        lib.require_device()
