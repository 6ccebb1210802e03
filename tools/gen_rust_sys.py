"""Generates rust/nexus-hip-sys/src/lib.rs — the `extern "C"` face of include/nexus_hip.h for the reference's Rust host code
 — from the header itself, so the binding cannot drift from the ABI: tests/test_rust_shim_cpu.py re-runs this
generator and fails on any difference, and checks names and arities independently.
  python tools/gen_rust_sys.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nexus_hip.h")
OUT = os.path.join(ROOT, "rust", "nexus-hip-sys", "src", "lib.rs")

PRIM = {"int": "c_int", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "uint8_t": "u8",
        "char": "c_char", "double": "f64", "void": "c_void"}


def strip_comments(t):
    return re.sub(r"/\*.*?\*/", "", t, flags=re.S)


def rust_type(ctype):
    """C type (no declarator name) -> Rust.  Handles const / pointer chains and `struct X`."""
    t = ctype.replace("struct ", "").strip()
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", t)
    base, consts, i = None, [], 0
    pending_const = False
    levels = []                      # constness of the thing each '*' points TO
    cur_const = False
    for tok in toks:
        if tok == "const":
            cur_const = True
        elif tok == "*":
            levels.append(cur_const)
            cur_const = False
        else:
            base = tok
    r = PRIM.get(base, base)
    for c in levels:
        r = ("*const " if c else "*mut ") + r
    if not levels and r == "c_void":
        return "()"
    return r


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_param(p):
    """'const uint32_t alpha[4]' / 'uint32_t* const* d_dst4' / 'nx_ctx* ctx' -> (name, rust type)"""
    p = p.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*\[\s*\d*\s*\]$", p)
    if m:                                            # array parameter decays to a pointer to its element type
        return m.group(2), rust_type(m.group(1).strip() + "*")
    m = re.match(r"^(.*[\*\s])([A-Za-z_][A-Za-z0-9_]*)$", p)
    return m.group(2), rust_type(m.group(1))


def parse_fnptr_field(f):
    m = re.match(r"^(.*?)\(\s*\*\s*([A-Za-z_][A-Za-z0-9_]*)\s*\)\s*\((.*)\)$", f.strip(), flags=re.S)
    ret, name, args = m.group(1).strip(), m.group(2), split_args(m.group(3))
    ps = ", ".join("%s: %s" % parse_param(a) for a in args)
    rt = rust_type(ret)
    return name, "Option<unsafe extern \"C\" fn(%s)%s>" % (ps, "" if rt == "()" else " -> " + rt)


def parse_header(text):
    t = strip_comments(text)
    consts = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(NX_[A-Z0-9_]+)\s+\(?(-?\d+)\)?\s*$", t, flags=re.M)]
    enums = []
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", t, flags=re.S):
        for e in m.group(1).split(","):
            e = e.strip()
            if e:
                k, v = [x.strip() for x in e.split("=")]
                enums.append((k, v))
    opaque = [m.group(1) for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+\1\s*;", t)]
    structs = []
    for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{(.*?)\}\s*(\w+)\s*;", t, flags=re.S):
        name, body, fields = m.group(3), m.group(2), []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            if "(*" in decl:
                fields.append(parse_fnptr_field(decl))
                continue
            parts = [x.strip() for x in decl.split(",")]
            fm = re.match(r"^(.+?)\s*([A-Za-z_]\w*)\s*(?:\[\s*(\d+)\s*\])?$", parts[0], flags=re.S)
            first_type = fm.group(1).strip()
            base = first_type.rstrip("* ").strip()           # later declarators of one declaration share the base type only
            for k, nm in enumerate(parts):
                if k == 0:
                    nm_, ty, arr = fm.group(2), first_type, fm.group(3)
                else:
                    am = re.match(r"^(\**)\s*([A-Za-z_]\w*)\s*(?:\[\s*(\d+)\s*\])?$", nm)
                    nm_, ty, arr = am.group(2), base + am.group(1), am.group(3)
                rt = rust_type(ty)
                if arr:
                    rt = "[%s; %s]" % (rt, arr)
                fields.append((nm_, rt))
        structs.append((name, fields))
    t2 = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", "", t, flags=re.S)
    t2 = re.sub(r"enum\s*\{.*?\}\s*;", "", t2, flags=re.S)
    funcs = []
    for m in re.finditer(r"^([A-Za-z_][\w\s\*]*?)\b(nx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", t2, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [parse_param(a) for a in split_args(args)]
        funcs.append((name, params, rust_type(ret)))
    return consts, enums, opaque, structs, funcs


def generate():
    consts, enums, opaque, structs, funcs = parse_header(open(HEADER).read())
    L = ["// GENERATED by tools/gen_rust_sys.py from include/nexus_hip.h — do not edit; tests/test_rust_shim_cpu.py fails on drift.",
         "// `nexus-hip-sys`: the raw C ABI of libnexus_hip.so (MI355X backend for the Nexus zkVM commit-and-prove path).",
         "// The safe layer — `HipBackend` and its Stwo trait impls — is the sibling crate `nexus-hip` (INTEGRATION.md §2).",
         "#![allow(non_camel_case_types, non_upper_case_globals, clippy::too_many_arguments)]",
         "use std::os::raw::{c_char, c_int, c_void};", ""]
    for k, v in consts:
        L.append("pub const %s: c_int = %s;" % (k, v))
    L.append("")
    for k, v in enums:
        L.append("pub const %s: u32 = %s;" % (k, v))
    L.append("")
    declared = {s[0] for s in structs}
    for o in opaque:
        if o not in declared:
            L.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % o)
    L.append("")
    for name, fields in structs:
        L.append("#[repr(C)]\n#[derive(Copy, Clone)]\npub struct %s {" % name)
        for fn_, ft in fields:
            L.append("    pub %s: %s," % ("type_" if fn_ == "type" else fn_, ft))
        L.append("}")
    L.append("")
    L.append('#[link(name = "nexus_hip")]\nextern "C" {')
    for name, params, ret in funcs:
        ps = ", ".join("%s: %s" % (("type_" if n == "type" else n), t) for n, t in params)
        L.append("    pub fn %s(%s)%s;" % (name, ps, "" if ret == "()" else " -> " + ret))
    L.append("}")
    return "\n".join(L) + "\n"


if __name__ == "__main__":
    src = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != src:
            print("rust/nexus-hip-sys/src/lib.rs is out of date: run python tools/gen_rust_sys.py")
            sys.exit(1)
        print("up to date")
    else:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        open(OUT, "w").write(src)
        print("wrote", OUT, "(%d functions)" % src.count("    pub fn "))
