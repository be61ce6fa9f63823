"""No function of the package reads a global name its module does not define (the image ships no linter; tools/lint_names.py walks the
symbol tables).  This is what caught the relative imports and module-level names left behind when engine.py was split into a package."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_globals_in_the_package_and_the_entry_points():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lint_names
    paths = [os.path.join(ROOT, "ader_amd"), os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    bad = [(f, fn, name) for f in sorted(lint_names.files(paths)) for fn, _, name in lint_names.undefined(f)]
    assert not bad, bad


def test_function_level_relative_imports_resolve():
    """`from .x import y` inside a function only fails when the function runs: resolve every one of them now."""
    import ast
    import importlib
    pkg_root = os.path.join(ROOT, "ader_amd")
    for d, _, fs in os.walk(pkg_root):
        for f in fs:
            if not f.endswith(".py"):
                continue
            path = os.path.join(d, f)
            rel = os.path.relpath(path, ROOT)[:-3].replace(os.sep, ".")
            pkg = rel if f == "__init__.py" else rel.rsplit(".", 1)[0]
            if f == "__init__.py":
                pkg = rel[:-len(".__init__")]
            for node in ast.walk(ast.parse(open(path).read())):
                if isinstance(node, ast.ImportFrom) and node.level > 0:
                    base = pkg.split(".")
                    base = base[:len(base) - (node.level - 1)]
                    target = ".".join(base + ([node.module] if node.module else []))
                    if target == "ader_amd.ops":
                        continue                      # (importing it loads the HIP library and registers torch ops)
                    mod = importlib.import_module(target)
                    for a in node.names:
                        if a.name != "*" and not hasattr(mod, a.name):
                            importlib.import_module(target + "." + a.name)     # a submodule
