"""The C-ABI library loads and exports every symbol include/glimpse_hip.h declares (CPU, no compute)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "glimpse_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(glh_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from glimpse_amd import _lib

    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    from glimpse_amd import _lib, build

    build.build(verbose=False)
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.glh_version() == 100
    assert lib.glh_stage_count() == len(_lib.stage_names()) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from glimpse_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.GlhError):
        _lib.load()
