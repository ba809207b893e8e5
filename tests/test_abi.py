"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/glorie_hip.h declares (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "glorie_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(glorie_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    import glorie_slam_amd.build as b
    lib_path = b.build()
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in glorie_hip.h but not exported"


def test_binding_table_matches_header():
    from glorie_slam_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_string():
    from glorie_slam_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.glorie_version()


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "glorie_slam_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "/root/reference" not in txt.replace("/root/reference/src", "REF"), f


def test_cpu_tensor_rejected():
    import pytest
    import torch
    from glorie_slam_amd import droid_backends as db, _lib
    vol = torch.zeros(1, 2, 2, 2, 2, dtype=torch.float16)
    coords = torch.zeros(1, 2, 2, 2)
    with pytest.raises(_lib.GlorieError):
        db.corr_index_forward(vol, coords, 3)
