"""The Rust binding under rust/ cannot be compiled here (no cargo / rustc in the image); what a compiler would catch about the FFI
surface is checked textually instead: the -sys crate is exactly what tools/gen_rust_sys.py generates from include/nexus_hip.h,
names and arities agree with an independent parse of the header, and the hand-written `nexus-hip` crate only calls functions the
-sys crate declares, with the declared number of arguments."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS = os.path.join(ROOT, "rust", "nexus-hip-sys", "src", "lib.rs")
HIP = os.path.join(ROOT, "rust", "nexus-hip", "src", "lib.rs")
HEADER = os.path.join(ROOT, "include", "nexus_hip.h")


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "(<[{"
        depth -= ch in ")>]}"
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def header_functions():
    t = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    t = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", "", t, flags=re.S)   # callbacks inside nx_comm are fields, not functions
    fns = {}
    for m in re.finditer(r"\b(nx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", t, flags=re.S):
        args = m.group(2).strip()
        fns[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    return fns


def rust_extern_functions():
    src = open(SYS).read()
    block = src[src.index('extern "C" {'):]
    return {m.group(1): (0 if not m.group(2).strip() else len(_split_args(m.group(2)))) for m in re.finditer(r"pub fn (nx_\w+)\((.*?)\)(?: -> [^;]+)?;", block)}


def test_sys_crate_is_what_the_generator_emits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_header_function_is_bound_with_the_same_arity():
    h, r = header_functions(), rust_extern_functions()
    assert len(h) >= 80
    assert set(h) == set(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    assert {k: v for k, v in h.items() if r[k] != v} == {}


def test_structs_mirror_the_header_field_for_field():
    src = open(SYS).read()
    t = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, n_fields in (("nx_component_spec", 5), ("nx_pcs_config", 7), ("nx_cinstr", 4), ("nx_air_component", 14), ("nx_comm", 10), ("nx_logup_frac", 6)):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, src, flags=re.S).group(1)
        assert len(re.findall(r"^\s*pub \w+:", body, flags=re.M)) == n_fields, name
    assert "log_constraint_degree_bound" in re.search(r"typedef struct nx_component_spec \{(.*?)\}", t, flags=re.S).group(1)


def test_hand_written_crate_calls_only_declared_functions_with_the_declared_arity():
    decl = rust_extern_functions()
    src = open(HIP).read()
    calls = list(re.finditer(r"sys::(nx_\w+)\(", src))
    assert len(calls) > 40
    for m in calls:
        name = m.group(1)
        assert name in decl, name
        # the argument list: balanced parentheses from the call site
        i, depth = m.end(), 1
        while depth:
            depth += src[i] == "("
            depth -= src[i] == ")"
            i += 1
        args = src[m.end():i - 1].strip()
        n = 0 if not args else len(_split_args(args))
        assert n == decl[name], (name, n, decl[name])
    assert "core/src/lib.rs:22-24" in src
