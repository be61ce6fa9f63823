"""Markdown list of the product ABI for INTEGRATION.md ("## Exported symbols"), grouped by the section banners of include/ader_hip.h
(dev tool; tests/test_abi_exports.py holds the list in INTEGRATION.md to `nm -D libader_hip.so`)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "include", "ader_hip.h")).read()
src = re.sub(r"#ifdef ADER_XCHECK.*?#endif[^\n]*", "", src, flags=re.S)
out, cur, title = [], [], "conventions / dropout descriptor"
for piece in re.split(r"(/\* ---- .*?\*/)", src, flags=re.S):
    if piece.startswith("/* ----"):
        if cur:
            out.append((title, cur))
        t = re.sub(r"\s+", " ", piece[7:-2]).strip(" -")
        title, cur = re.sub(r"\s*-{3,}.*$", "", t.split(" -- ")[0].split(": ", 1)[0])[:110], []
    else:
        body = re.sub(r"/\*.*?\*/", "", piece, flags=re.S)
        cur += re.findall(r"\b(?:int|size_t)\s+(ader_[a-z0-9_]+)\s*\(", body)
if cur:
    out.append((title, cur))
n = sum(len(c) for _, c in out)
print("## Exported symbols\n")
print("`nm -D ader_amd/libader_hip.so` == the declarations of `include/ader_hip.h` == this list (%d functions; held equal by\n"
      "`tests/test_abi_exports.py`).  The cross-check kernels of the tests (`ader_tab_update`, `ader_tab_update_kd`,\n"
      "`ader_herding_select_generic`) are NOT here: they are built into `libader_xcheck.so` only (`-DADER_XCHECK`).\n" % n)
for t, c in out:
    if c:
        print("* **%s** — %s" % (t, ", ".join("`%s`" % x for x in dict.fromkeys(c))))
