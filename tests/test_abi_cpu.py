"""The C-ABI library loads without a GPU and exports every symbol include/orbhip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "orbhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orb_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from orb_slam2_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names


def test_loader_binds_and_reports_errors():
    from orb_slam2_amd import _lib
    L = _lib.lib()
    assert L.orb_last_error() is not None
    # argument validation happens before any device work
    assert L.orb_extractor_get_scale_tables(None, None, None, None, None) == _lib.ERR_ARG
    assert b"null" in L.orb_last_error()
    assert L.orb_match_scratch_bytes(4, 2000) >= 4 * 2000 * 60


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "orb_slam2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower(), os.path.join(dirpath, f)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from orb_slam2_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected ImportError")
