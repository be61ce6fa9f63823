"""Undefined-name check for the package's Python sources (no linter ships in the image): every name a function reads as a global must be
bound at module level (import / def / class / assignment) or be a builtin.  python tools/lint_names.py [paths...]; exit code 1 on findings.
Also reports module-level imports that nothing in the module uses (--unused)."""
import ast
import builtins
import os
import symtable
import sys


def module_names(tree):
    names = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                names.add((a.asname or a.name).split(".")[0])
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            names.add(node.name)
        elif isinstance(node, (ast.Assign, ast.AugAssign, ast.AnnAssign, ast.For, ast.With, ast.If, ast.Try)):
            for n in ast.walk(node):
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                    names.add(n.id)
    return names


def undefined(path):
    src = open(path).read()
    tree = ast.parse(src, path)
    top = module_names(tree) | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    bad = []

    def walk(tab):
        for s in tab.get_symbols():
            if s.is_global() and s.is_referenced() and s.get_name() not in top:
                bad.append((tab.get_name(), tab.get_lineno(), s.get_name()))
        for c in tab.get_children():
            walk(c)

    walk(symtable.symtable(src, path, "exec"))
    return bad


def unused_imports(path):
    src = open(path).read()
    tree = ast.parse(src, path)
    imported = {}
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                imported[(a.asname or a.name).split(".")[0]] = node.lineno
    used = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name)} | {n.value.id for n in ast.walk(tree)
                                                                        if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name)}
    return [(k, ln) for k, ln in imported.items() if k not in used]


def files(paths):
    for p in paths:
        if os.path.isdir(p):
            for d, _, fs in os.walk(p):
                for f in fs:
                    if f.endswith(".py"):
                        yield os.path.join(d, f)
        else:
            yield p


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ader_amd")]
    rc = 0
    for f in sorted(files(args)):
        for fn, ln, name in undefined(f):
            print("%s:%d: %s reads undefined global %r" % (f, ln, fn, name))
            rc = 1
        if "--unused" in sys.argv and not f.endswith("__init__.py"):
            for name, ln in unused_imports(f):
                print("%s:%d: unused import %r" % (f, ln, name))
    sys.exit(rc)
