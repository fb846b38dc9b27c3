"""CPU: libcss_hip.so loads without a GPU and exports every symbol that
include/css_hip.h declares; the ctypes table covers the same set; compute entry
points fail loudly (no CPU fallback) when no HIP device is present."""
import ctypes
import re
from pathlib import Path

import pytest

from claude_semantic_search_amd import _native as nat

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "css_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(css_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("css_index_create", "css_index_add", "css_index_search", "css_index_search_dev",
                 "css_merge_topk_dev", "css_encoder_create", "css_encoder_forward", "css_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = nat.lib()
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/css_hip.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(nat.PROTOTYPES) == declared_symbols()


def test_version_and_error_strings():
    lib = nat.lib()
    assert b"css_hip" in lib.css_version()
    assert isinstance(nat.last_error(), str)


def test_fails_loudly_without_device():
    if nat.device_count() > 0:
        pytest.skip("a HIP device is present")
    h = ctypes.c_void_p()
    rc = nat.lib().css_index_create(768, 0, 0, ctypes.byref(h))
    assert rc == nat.CSS_ERR_NO_DEVICE
    assert "no HIP device" in nat.last_error()
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    with pytest.raises(RuntimeError, match="no HIP device"):
        IndexFlatIP(768)
