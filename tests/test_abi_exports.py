"""The C-ABI library must load and export every symbol that include/ader_hip.h declares (no compute calls: this
runs without a GPU).  Also checks that the ctypes table of ader_amd/_lib.py covers exactly those symbols."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(xcheck=False):
    """Functions include/ader_hip.h declares: the product ABI, or (xcheck) the block inside #ifdef ADER_XCHECK."""
    src = open(os.path.join(ROOT, "include", "ader_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    blocks = re.findall(r"#ifdef ADER_XCHECK(.*?)#endif", src, flags=re.S)
    src = "".join(blocks) if xcheck else re.sub(r"#ifdef ADER_XCHECK.*?#endif", "", src, flags=re.S)
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


def test_product_library_exports_exactly_the_documented_abi(lib_path):
    """`nm -D libader_hip.so` == the declarations of include/ader_hip.h outside ADER_XCHECK == the list in INTEGRATION.md; the
    cross-check kernels live in libader_xcheck.so only."""
    import subprocess
    from ader_amd import _lib, build

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
        return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("ader_") and " T " in ln})
    assert exported(lib_path) == declared_symbols()
    x = declared_symbols(xcheck=True)
    assert x == sorted(_lib._XSIGS) and len(x) == 3
    assert set(x) <= set(exported(build.XLIB)) and not set(x) & set(exported(lib_path))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = doc.split("## Exported symbols")[1].split("\n## ")[0]
    listed = sorted(set(re.findall(r"`(ader_[a-z0-9_]+)`", "\n".join(ln for ln in section.splitlines() if ln.startswith("* ")))))
    assert listed == declared_symbols(), (sorted(set(declared_symbols()) - set(listed)), sorted(set(listed) - set(declared_symbols())))


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
