"""Drift guard for the binding a maintainer of the reference would actually use: the `foreign import ccall` lines of
haskell/Numeric/LinearAlgebra/Sparse/HIP.hs -- and their excerpt in INTEGRATION.md section 2 -- against include/sla_hip.h: every imported
symbol exists in the header, with the same number of arguments, and every argument / the result has the same C width class (32-bit int,
64-bit int, double, pointer).  No GHC in this image: this is the only mechanical check the Haskell side can get here; it is what
test_cabi_and_host.py::test_library_exports_every_header_symbol is for the ctypes table.  CPU only."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HANDLES = {"sla_ctx_t": "Ctx", "sla_csr_t": "Csr", "sla_vec_t": "Vec", "sla_solver_t": "Solver"}


def _c_class(decl):
    """Width class of one C parameter / result declaration."""
    d = decl.strip()
    d = re.sub(r"/\*.*?\*/", "", d).strip()
    if "*" in d or "[" in d:
        return "ptr"
    words = [w for w in re.split(r"\s+", d) if w not in ("const", "unsigned", "signed", "struct")]
    ty = words[0] if words else ""
    if ty in HANDLES:
        return "ptr"
    if ty in ("int64_t", "uint64_t", "size_t", "long"):
        return "i64"
    if ty in ("int", "int32_t", "uint32_t"):
        return "i32"
    if ty == "double":
        return "f64"
    if ty == "void":
        return "void"
    raise AssertionError(f"unclassified C type in {decl!r}")


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "sla_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*)?)\s*(sla_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ps = [p for p in (q.strip() for q in params.replace("\n", " ").split(",")) if p and p != "void"]
        protos[name] = (_c_class(ret), [_c_class(p) for p in ps])
    return protos


def _hs_class(t):
    t = t.strip()
    if t.startswith(("Ptr", "FunPtr", "(Ptr")) or t in ("CString",):
        return "ptr"
    if t in ("CInt", "Int32", "CUInt"):
        return "i32"
    if t in ("Int64", "CLong", "CLLong", "Word64", "CSize"):
        return "i64"
    if t in ("Double", "CDouble"):
        return "f64"
    raise AssertionError(f"unclassified Haskell FFI type {t!r}")


def _split_arrows(sig):
    """Top-level `->` split of a Haskell type (parentheses respected)."""
    parts, depth, cur = [], 0, ""
    i = 0
    while i < len(sig):
        ch = sig[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if depth == 0 and sig.startswith("->", i):
            parts.append(cur.strip())
            cur = ""
            i += 2
            continue
        cur += ch
        i += 1
    parts.append(cur.strip())
    return parts


def foreign_imports(path):
    out = []
    for line in open(path).read().splitlines():
        m = re.match(r'\s*foreign import ccall (?:safe|unsafe)\s+"(&?)(sla_[a-z0-9_]+)"\s+\w+\s*::\s*(.*?)(?:\s+--.*)?$', line)
        if m:
            out.append((m.group(2), bool(m.group(1)), m.group(3).strip()))
    return out


def _check(path, minimum):
    protos = header_prototypes()
    assert len(protos) >= 60 and "sla_linsolve0" in protos and protos["sla_spmv"] == ("i32", ["ptr", "ptr", "ptr"]), "the header parser lost its footing"
    imports = foreign_imports(path)
    assert len(imports) >= minimum, (path, len(imports))
    for name, by_address, sig in imports:
        assert name in protos, f"{os.path.basename(path)}: {name} is not declared in include/sla_hip.h"
        c_ret, c_args = protos[name]
        if by_address:   # "&sla_x_destroy" :: FunPtr (Ptr X -> IO ()): a ForeignPtr finalizer -- one pointer in, the int result dropped
            m = re.match(r"FunPtr\s*\((.*)\)$", sig)
            assert m, (name, sig)
            parts = _split_arrows(m.group(1))
            assert parts[-1] == "IO ()" and [_hs_class(p) for p in parts[:-1]] == c_args == ["ptr"], (name, sig, c_args)
            continue
        parts = _split_arrows(sig)
        res = parts[-1]
        assert res.startswith("IO "), (name, sig)
        hs_ret = _hs_class(res[3:].strip())
        hs_args = [_hs_class(p) for p in parts[:-1]]
        assert len(hs_args) == len(c_args), f"{name}: {len(hs_args)} arguments in the Haskell import, {len(c_args)} in the header"
        assert hs_args == c_args, f"{name}: argument widths {hs_args} (Haskell) vs {c_args} (header)"
        assert hs_ret == c_ret, f"{name}: result {hs_ret} (Haskell) vs {c_ret} (header)"


def test_haskell_shim_imports_match_the_header():
    _check(os.path.join(ROOT, "haskell", "Numeric", "LinearAlgebra", "Sparse", "HIP.hs"), 28)


def test_integration_md_excerpt_matches_the_header():
    _check(os.path.join(ROOT, "INTEGRATION.md"), 10)


def test_the_guard_catches_a_drifted_import(tmp_path):
    """The guard itself: one argument dropped, one width changed, one unknown symbol -- each must be reported."""
    good = 'foreign import ccall safe "sla_spmv" c_spmv :: Ptr Csr -> Ptr Vec -> Ptr Vec -> IO CInt\n'
    for bad, what in (('foreign import ccall safe "sla_spmv" c_spmv :: Ptr Csr -> Ptr Vec -> IO CInt\n', "arguments"),
                      ('foreign import ccall safe "sla_solver_step" c_step :: Ptr Solver -> Int64 -> IO CInt\n', "widths"),
                      ('foreign import ccall safe "sla_no_such_entry" c_x :: Ptr Ctx -> IO CInt\n', "not declared")):
        p = tmp_path / "Drift.hs"
        p.write_text(good * 12 + bad)
        with pytest.raises(AssertionError, match=what):
            _check(str(p), 10)
