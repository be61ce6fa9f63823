"""The C-ABI library must load and export every symbol that include/ader_hip.h declares (no compute calls: this
runs without a GPU).  Also checks that the ctypes table of ader_amd/_lib.py covers exactly those symbols."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ader_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t)\s+(ader_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from ader_amd import build
    return build.build()


def test_header_symbols_are_exported(lib_path):
    import torch  # noqa: F401  (one HIP runtime for the process, see ader_amd/_lib.py)
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libader_hip.so does not export %s" % n


def test_ctypes_table_matches_header(lib_path):
    from ader_amd import _lib
    assert _lib.exported_symbols() == declared_symbols()
    _lib.load()


def test_product_path_fails_loudly_without_gpu():
    import torch
    from ader_amd import _lib
    from ader_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AderHipError):
        Engine(100)


def test_product_code_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ader_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dirpath, f)
