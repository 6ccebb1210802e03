"""Every `stwo…` / `stwo_constraint_framework…` / `nexus_…` path a set of Rust sources names — in `use` declarations (nested braces
expanded, `self` and `as` handled) and inline (`stwo::prover::poly::twiddles::TwiddleTree<…>`).

  python tools/reference_use_paths.py /root/reference > tests/golden/reference_use_paths.txt      (the fixture: what the REFERENCE names)
  python tools/reference_use_paths.py tools rust                                                  (what this repository's Rust names)

tests/test_rust_shim_cpu.py holds this repository's Rust files (tools/*.rs, rust/**/*.rs) to the fixture: a path they name must be
one the reference itself names, or be listed — with the reason — in rust/UNOBSERVED_PATHS.txt.  The Stwo crate is not under
/root/reference (an un-vendored git dependency), so the reference's own `use` lines are the only evidence here of the crate's module
layout at the pinned revision (tools/dump_reference.rs had been written against a layout the reference does not use)."""
import os
import re
import sys

ROOTS = ("stwo::", "stwo_constraint_framework::", "nexus_")


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return "\n".join(line.split("//")[0] for line in src.splitlines())


def expand(tree, prefix=""):
    """`a::{b, c::{d, self}, e as f}` -> full paths"""
    tree = tree.strip()
    out = []
    depth, start, parts = 0, 0, []
    for i, ch in enumerate(tree):
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        elif ch == "," and depth == 0:
            parts.append(tree[start:i]); start = i + 1
    parts.append(tree[start:])
    for part in parts:
        part = part.strip()
        if not part:
            continue
        m = re.match(r"^([^{]*?)::\s*\{(.*)\}\s*$", part, flags=re.S)
        if m:
            out += expand(m.group(2), prefix + m.group(1).strip() + "::")
            continue
        part = re.sub(r"\s+as\s+\w+$", "", part).strip()
        if part == "self":
            out.append(prefix.rstrip(":"))
        elif part == "*":
            out.append(prefix + "*")
        else:
            out.append(prefix + part)
    return out


def paths_of(src):
    src = strip_comments(src)
    found = set()
    for m in re.finditer(r"\buse\s+([^;]+);", src, flags=re.S):
        for p in expand(re.sub(r"\s+", " ", m.group(1))):
            p = p.replace(" ", "")
            if p.startswith("::"):
                p = p[2:]
            found.add(p)
    for m in re.finditer(r"(?<![:\w])((?:stwo|stwo_constraint_framework|nexus_\w+)(?:::\w+)+)", src):
        found.add(m.group(1))
    return {p for p in found if p.startswith(ROOTS) and "::" in p and not p.startswith(("nexus_hip::", "nexus_hip_sys::"))}


def scan(roots):
    out = set()
    for root in roots:
        if os.path.isfile(root):
            files = [root]
        else:
            files = [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if f.endswith(".rs")]
        for f in files:
            try:
                out |= paths_of(open(f, encoding="utf-8", errors="replace").read())
            except OSError:
                pass
    return out


if __name__ == "__main__":
    for p in sorted(scan(sys.argv[1:] or ["."])):
        print(p)
