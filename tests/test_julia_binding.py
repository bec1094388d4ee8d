"""julia/RsysHIP.jl against include/rsys.h without Julia (absent from the image): every `ccall` tuple of the binding is parsed and
its symbol, return type, arity and argument types are checked against the header's prototype; the two struct mirrors are checked
field by field; every entry point of the header must be bound, and none of the test hooks of include/rsys_debug.h.  What this cannot check is
behaviour -- the .jl file has never executed -- only that what it declares is the ABI the library exports."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rsys.h")
JULIA = os.path.join(ROOT, "julia", "RsysHIP.jl")

DEBUG_HEADER = os.path.join(ROOT, "include", "rsys_debug.h")   # test / parity hooks: not part of the boundary, not bound

SCALAR = {"int32_t": "Int32", "int64_t": "Int64", "uint64_t": "UInt64", "uint8_t": "UInt8", "float": "Float32", "double": "Float64",
          "size_t": "Csize_t", "int": "Int32", "char": "UInt8", "void": "Cvoid"}
HANDLES = {"rsys_model", "rsys_optimizer", "rsys_comm"}
STRUCTS = {"rsys_config": "RsysConfig", "rsys_batch": "RsysBatch"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_type_to_julia(ctype):
    """canonical Julia type of a C parameter type (arrays decay to pointers); Ref{T} and Ptr{T} are the same ABI"""
    t = ctype.replace("const", " ").strip()
    depth = t.count("*")
    base = t.replace("*", " ").split()[0]
    if base in HANDLES:
        jt = "Cvoid"
    elif base in STRUCTS:
        jt = STRUCTS[base]
    else:
        jt = SCALAR[base]
    for _ in range(depth):
        jt = f"Ptr{{{jt}}}"
    return jt


def header_prototypes(path=HEADER):
    text = strip_comments(open(path).read())
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int32_t|size_t)\s+(rsys_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[[^\]]*\]\s*$", a)
                if arr:
                    a = a[:arr.start()].strip()
                mm = re.match(r"(.*?)(\w+)$", a)               # the last identifier is the parameter name
                ctype = mm.group(1).strip() if mm and mm.group(1).strip() else a
                if arr:
                    ctype += "*"
                params.append(c_type_to_julia(ctype))
        protos[name] = ("Cstring" if "char" in ret else SCALAR[ret], params)
    return protos


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def norm_julia(t):
    t = t.replace(" ", "")
    t = t.replace("Ref{", "Ptr{")
    t = t.replace("Cstring", "Ptr{UInt8}").replace("Cint", "Int32")
    return t


def julia_ccalls():
    text = re.sub(r"#.*", "", open(JULIA).read())
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\),\s*([\w{}]+),\s*\(", text):
        i = m.end(); depth = 1; j = i
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0); j += 1
        args = text[i:j - 1].strip()
        if args.endswith(","):
            args = args[:-1]
        calls.append((m.group(1), m.group(2), split_top(args) if args else []))
    return calls


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) >= 50, len(protos)
    calls = julia_ccalls()
    assert len(calls) >= 45
    for name, ret, args in calls:
        assert name in protos, f"{name}: not in include/rsys.h"
        want_ret, want_args = protos[name]
        assert norm_julia(ret) == norm_julia(want_ret), (name, ret, want_ret)
        assert len(args) == len(want_args), (name, args, want_args)
        for k, (a, w) in enumerate(zip(args, want_args)):
            assert norm_julia(a) == norm_julia(w), f"{name}: argument {k} is {a}, the header says {w}"


def test_every_entry_point_is_bound_or_listed_as_test_only():
    protos = header_prototypes()
    bound = {c[0] for c in julia_ccalls()}
    missing = sorted(set(protos) - bound)
    assert not missing, missing
    debug_only = set(header_prototypes(DEBUG_HEADER))
    assert len(debug_only) >= 15 and not (debug_only & set(protos))
    assert not (bound & debug_only)


def _c_struct_fields(name):
    text = strip_comments(open(HEADER).read())
    body = re.search(r"typedef\s+struct\s+" + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        mm = re.match(r"((?:const\s+)?\w+)\s*(.*)$", decl)
        base, rest = mm.group(1), mm.group(2)
        for item in rest.split(","):
            item = item.strip()
            stars = item.count("*")
            ident = item.replace("*", "").strip()
            arr = re.search(r"\[(\d+)\]", ident)
            fname = ident[:arr.start()] if arr else ident
            jt = c_type_to_julia(base + "*" * stars)
            if arr:
                jt = f"NTuple{{{arr.group(1)},{jt}}}"
            fields.append((fname, jt))
    return fields


def _julia_struct_fields(name):
    text = re.sub(r"#.*", "", open(JULIA).read())
    body = re.search(r"struct\s+" + name + r"\b(.*?)\nend", text, flags=re.S).group(1)
    return [(f, t.replace(" ", "")) for f, t in re.findall(r"(\w+)::([\w{},\s]+?)(?=;|\n|$)", body)]


def test_struct_mirrors_have_the_headers_fields_in_order():
    for cname, jname in STRUCTS.items():
        c, j = _c_struct_fields(cname), _julia_struct_fields(jname)
        assert [f for f, _ in c] == [f for f, _ in j], (cname, [f for f, _ in c], [f for f, _ in j])
        for (f, ct), (_, jt) in zip(c, j):
            assert norm_julia(ct) == norm_julia(jt), (cname, f, ct, jt)


def test_dtype_constants_agree_with_the_header():
    h = strip_comments(open(HEADER).read())
    m = re.search(r"enum\s*\{\s*RSYS_DTYPE_FP32\s*=\s*(\d+),\s*RSYS_DTYPE_BF16\s*=\s*(\d+),\s*RSYS_DTYPE_FP8\s*=\s*(\d+)\s*\}", h)
    j = re.sub(r"#.*", "", open(JULIA).read())
    mj = re.search(r"const DTYPE_FP32 = Int32\((\d+)\); const DTYPE_BF16 = Int32\((\d+)\); const DTYPE_FP8 = Int32\((\d+)\)", j)
    assert m and mj and m.groups() == mj.groups()
