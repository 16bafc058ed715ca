"""Static cross-check of the Julia `ccall` shim (smc.jl_amd/julia/SMCMI.jl) against include/smcmi.h.

Julia is not part of the build image, so the shim has never met a Julia parser here.  What CAN be verified without one: every
`ccall((:name, LIB), ret, (argtypes...), args...)` names a function the header declares, with the same arity, the same class of every
argument (pointer / 32-bit int / 64-bit int / double / struct by reference / string) and the same number of values passed; and the four
structs the shim mirrors have the header's field widths in the header's order (sizes computed with C layout rules from the Julia field
lists against ctypes.sizeof of the binding the GPU tests run on, which tests/test_abi_cpu.py pins against the compiled header).
"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "smc.jl_amd", "julia", "SMCMI.jl")
HDR = os.path.join(ROOT, "include", "smcmi.h")


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _split_top(s, sep=","):
    """split on `sep` at nesting depth 0 of () [] {}; string literals are skipped"""
    out, depth, cur, i = [], 0, [], 0
    while i < len(s):
        ch = s[i]
        if ch == '"':
            j = i + 1
            while j < len(s) and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            cur.append(s[i:j + 1])
            i = j + 1
            continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
        i += 1
    tail = "".join(cur).strip()
    if tail:
        out.append(tail)
    return out


def _balanced(s, start):
    """s[start] == '(' -> index just behind its matching ')'"""
    depth, i = 0, start
    while i < len(s):
        ch = s[i]
        if ch == '"':
            i += 1
            while s[i] != '"':
                i += 2 if s[i] == "\\" else 1
        elif ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise AssertionError("unbalanced parentheses in a ccall")


def header_prototypes():
    """name -> (return class, [argument classes]) for every function include/smcmi.h declares"""
    txt = _strip_c_comments(open(HDR).read())
    txt = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", txt, flags=re.S)
    txt = re.sub(r"typedef[^;]*;", " ", txt)
    txt = re.sub(r"enum\s*\{.*?\}\s*;", " ", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|;|\{)\s*((?:const\s+)?[A-Za-z_][\w\s]*?[\s\*]+)(smcmi_\w+)\s*\(([^;{}]*?)\)\s*(?=;)", txt, flags=re.S | re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        a = [] if args.strip() in ("", "void") else [c_class(x) for x in _split_top(args)]
        protos[name] = (c_class(ret + " x"), a)
    return protos


STRUCTS_C = {"smcmi_config": "Config", "smcmi_run_config": "RunConfig", "smcmi_result": "Result", "smcmi_loop_state": "LoopState",
             "smcmi_stage_stats": "StageStats", "smcmi_host_comm": "HostComm"}


def c_class(decl):
    """class of one C parameter declaration (name included or not)"""
    d = " ".join(decl.replace("*", " * ").split())
    d = re.sub(r"\bconst\b", "", d).strip()
    stars = d.count("*")
    base = d.split("*")[0].split()
    # drop a trailing parameter name
    if stars == 0 and len(base) > 1:
        base = base[:-1]
    btype = " ".join(base)
    if stars:
        if btype == "char":
            return "cstring" if stars == 1 else "ptr"
        if btype in STRUCTS_C and stars == 1:
            return "struct:" + STRUCTS_C[btype]
        return "ptr"
    if btype in ("smcmi_lik_callback",):
        return "ptr"
    return {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64"}[btype]


def jl_class(t):
    t = t.strip()
    if t in ("Handle", "Ptr{Cvoid}") or t.startswith("Ptr{"):
        return "ptr"
    m = re.fullmatch(r"Ref\{(\w+)\}", t)
    if m:
        inner = m.group(1)
        return "struct:" + inner if inner in STRUCTS_C.values() else "ptr"
    return {"Cint": "i32", "Int32": "i32", "UInt32": "u32", "Int64": "i64", "UInt64": "u64", "Float64": "f64", "Cstring": "cstring",
            "Cdouble": "f64", "Clonglong": "i64"}[t]


def julia_ccalls():
    """[(line, name, return class, [argument classes], number of values passed)] for every ccall in the shim"""
    src = open(JL).read()
    # comments: '#' to the end of the line (the shim has no '#' inside string literals that matter; '#=' blocks are not used)
    lines = []
    for ln in src.split("\n"):
        out, in_str, i = [], False, 0
        while i < len(ln):
            ch = ln[i]
            if ch == '"':
                in_str = not in_str
            if ch == "#" and not in_str:
                break
            out.append(ch)
            i += 1
        lines.append("".join(out))
    code = "\n".join(lines)
    calls = []
    for m in re.finditer(r"\bccall\s*\(", code):
        end = _balanced(code, m.end() - 1)
        inner = code[m.end():end - 1]
        parts = _split_top(inner)
        fm = re.fullmatch(r"\(\s*:(\w+)\s*,\s*LIB\s*\)", parts[0])
        assert fm, "ccall target not of the form (:name, LIB): %r" % parts[0]
        ret = jl_class(parts[1])
        at = parts[2].strip()
        assert at.startswith("(") and at.endswith(")"), at
        argt = [jl_class(x) for x in _split_top(at[1:-1]) if x]
        line = code.count("\n", 0, m.start()) + 1
        calls.append((line, fm.group(1), ret, argt, len(parts) - 3))
    return calls


def compatible(jl, c):
    if jl == c:
        return True
    # a struct passed by Ref{T} is a pointer on the C side; Ref{Float64}/Ref{Int32}/Ref{Handle} are out-pointers
    if jl.startswith("struct:"):
        return c == jl
    # signedness of 32-bit scalars is a reinterpretation, not an ABI difference - but the shim must still say 32 bits
    return False


def test_every_ccall_matches_a_header_prototype():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(protos) >= 50, sorted(protos)              # the header parser saw the whole header
    assert len(calls) >= 40                                # ... and the shim's calls were all found
    bad = []
    for line, name, ret, argt, nvals in calls:
        if name not in protos:
            bad.append("SMCMI.jl:%d %s: not declared in include/smcmi.h" % (line, name))
            continue
        cret, cargs = protos[name]
        if not compatible(ret, cret):
            bad.append("SMCMI.jl:%d %s: return %s, header %s" % (line, name, ret, cret))
        if len(argt) != len(cargs):
            bad.append("SMCMI.jl:%d %s: %d argument types, header %d" % (line, name, len(argt), len(cargs)))
            continue
        if nvals != len(argt):
            bad.append("SMCMI.jl:%d %s: %d values passed for %d argument types" % (line, name, nvals, len(argt)))
        for k, (a, b) in enumerate(zip(argt, cargs)):
            if not compatible(a, b):
                bad.append("SMCMI.jl:%d %s: argument %d is %s, header %s" % (line, name, k + 1, a, b))
    assert not bad, "\n".join(bad)


def test_header_parser_agrees_with_the_ctypes_binding():
    """the same classification applied to the binding the GPU tests run through: a parser that mis-read the header would show here"""
    from smc_jl_amd.host import _lib

    protos = header_prototypes()
    names = {n for n, _, _ in _lib.SYMBOLS}
    assert names == set(protos), (sorted(names - set(protos)), sorted(set(protos) - names))

    def ct_class(t):
        if t in (C.c_int, C.c_int32):
            return "i32"
        if t is C.c_uint32:
            return "u32"
        if t is C.c_int64:
            return "i64"
        if t is C.c_uint64:
            return "u64"
        if t is C.c_double:
            return "f64"
        if t is C.c_char_p:
            return "cstring"
        if isinstance(t, type) and issubclass(t, C._Pointer) and issubclass(t._type_, C.Structure):
            return "struct:" + t._type_.__name__
        return "ptr"

    for name, res, args in _lib.SYMBOLS:
        cret, cargs = protos[name]
        got = [ct_class(a) for a in args]
        # (the binding passes raw byte buffers as c_char_p where the header says uint8_t*: both pointers)
        norm = lambda xs: ["ptr" if x == "cstring" else x for x in xs]
        assert norm(got) == norm(cargs), (name, got, cargs)
        assert norm([ct_class(res)]) == norm([cret]), name


JL_WIDTH = {"Int32": (4, C.c_int32), "UInt32": (4, C.c_uint32), "Int64": (8, C.c_int64), "UInt64": (8, C.c_uint64), "Float64": (8, C.c_double)}


def julia_structs():
    src = open(JL).read()
    out = {}
    for m in re.finditer(r"^(?:mutable\s+)?struct\s+(\w+)\s*(?:#[^\n]*)?\n(.*?)^end", src, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name not in STRUCTS_C.values():
            continue
        fields = []
        for ln in body.split("\n"):
            ln = ln.split("#")[0]
            if "new(" in ln or "=" in ln:
                continue
            for f in ln.split(";"):
                fm = re.fullmatch(r"\s*(\w+)::(\w+)\s*", f)
                if fm:
                    fields.append((fm.group(1), fm.group(2)))
        out[name] = fields
    return out


def c_layout_size(types):
    off, amax = 0, 1
    for t in types:
        w = JL_WIDTH[t][0]
        off = (off + w - 1) // w * w
        off += w
        amax = max(amax, w)
    return (off + amax - 1) // amax * amax


@pytest.mark.parametrize("name", ["Config", "RunConfig", "Result", "LoopState"])
def test_mirrored_struct_layouts(name):
    from smc_jl_amd.host import _lib

    jl = julia_structs()
    assert name in jl, sorted(jl)
    ct = getattr(_lib, name)
    assert len(jl[name]) == len(ct._fields_), (name, len(jl[name]), len(ct._fields_))
    for (jn, jt), (cn, ctype) in zip(jl[name], ct._fields_):
        assert JL_WIDTH[jt][1] is ctype, (name, jn, jt, cn, ctype)
        # field names follow the header (the ctypes binding calls `lambda` lam: a Python keyword)
        assert jn == cn or (jn, cn) == ("lambda", "lam"), (name, jn, cn)
    assert c_layout_size([t for _, t in jl[name]]) == C.sizeof(ct), name
    # offsets, field by field, under C layout rules
    off = 0
    for (jn, jt), (cn, _) in zip(jl[name], ct._fields_):
        w = JL_WIDTH[jt][0]
        off = (off + w - 1) // w * w
        assert off == getattr(ct, cn).offset, (name, jn, off, getattr(ct, cn).offset)
        off += w


def test_header_struct_fields_match_the_julia_mirrors():
    """the struct definitions of include/smcmi.h themselves (names, order, widths) against the shim's field lists"""
    txt = _strip_c_comments(open(HDR).read())
    jl = julia_structs()
    cw = {"int32_t": "Int32", "uint32_t": "UInt32", "int64_t": "Int64", "uint64_t": "UInt64", "double": "Float64"}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        cname = m.group(2)
        if STRUCTS_C.get(cname) not in jl:
            continue
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            t, rest = decl.split(None, 1)
            for nm in rest.split(","):
                fields.append((nm.strip(), cw[t]))
        assert fields == jl[STRUCTS_C[cname]], (cname, fields, jl[STRUCTS_C[cname]])
