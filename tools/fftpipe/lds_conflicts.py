"""LDS bank-conflict model for the pipelined FFT kernels' tile layout (csrc/fft_pipe.cuh).

Rules from /opt/skills/guides/MI355X_MICROARCH.md (LDS section): a wave64 access is serviced in fixed lane groups, one LDS cycle
per group when conflict-free; bank of byte address a = (a/4) mod 64 for ds_read_b64/b128, (a/4) mod 32 for ds_read_b32 and every
ds_write; identical addresses broadcast; N distinct addresses on one bank within a group cost N cycles.
Prints, per access pattern of the kernels, the worst and average cycles per lane group relative to conflict-free (1.0).
"""
import sys

def swz(G):
    return G ^ ((G >> 4) & 1) ^ ((((G >> 3) ^ (G >> 5)) & 1) << 1) ^ (((G >> 6) & 1) << 2)

def swz_none(G):
    return G

def phys(t, f):
    return (f(t >> 2) << 2) | (t & 3)

R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
R128 = R128 + [[l + 32 for l in g] for g in R128]
W128 = [list(range(8 * k, 8 * k + 8)) for k in range(8)]
G32 = [list(range(32)), list(range(32, 64))]

def cost(addrs_words, groups, width_words, nbanks):
    """addrs_words[lane] = first word address of the lane's access."""
    worst, tot = 0, 0
    for g in groups:
        banks = {}
        for l in g:
            for k in range(width_words):
                a = addrs_words[l] + k
                banks.setdefault(a % nbanks, set()).add(a)
        c = max(len(v) for v in banks.values())
        worst = max(worst, c); tot += c
    return worst, tot / len(groups)

_quiet = False


def report(name, fn_addr, kind, f):
    """fn_addr(tid) -> word address (logical t) for all 512 lanes; evaluates every wave."""
    worst, avg, n = 0, 0.0, 0
    for wave in range(8):
        addrs = [phys(fn_addr(wave * 64 + l), f) for l in range(64)]
        if kind == "r32": w, a = cost(addrs, G32, 1, 32)
        elif kind == "w32": w, a = cost(addrs, G32, 1, 32)
        elif kind == "r128": w, a = cost(addrs, R128, 4, 64)
        elif kind == "w128": w, a = cost(addrs, W128, 4, 32)
        worst = max(worst, w); avg += a; n += 1
    if not _quiet:
        print(f"  {name:44s} {kind:5s} worst {worst:2d}x  avg {avg / n:5.2f}x")
    return worst

def row0(tid, bp):
    wl, wh = tid & ((1 << bp) - 1), tid >> bp
    return (wh << (bp + 4)) | wl

def worst_case(f, verbose=False):
    global _quiet
    _quiet = not verbose
    bad = 0
    if True:
        for bp in (2, 4, 5, 6, 7, 8, 9):
            for e in (0, 5, 15):
                bad = max(bad, report(f"round bp={bp} e={e}", lambda tid, bp=bp, e=e: row0(tid, bp) + (e << bp), "r32", f))
        for c in range(4):
            bad = max(bad, report(f"round bp=0 group c={c} (b128)", lambda tid, c=c: 4 * (4 * tid + c), "r128", f))
            bad = max(bad, report(f"round bp=0 group c={c} (b128)", lambda tid, c=c: 4 * (4 * tid + c), "w128", f))
        for m in range(4):
            bad = max(bad, report(f"consecutive groups + {512 * m}", lambda tid, m=m: 4 * (tid + 512 * m), "r128", f))
            bad = max(bad, report(f"consecutive groups + {512 * m}", lambda tid, m=m: 4 * (tid + 512 * m), "w128", f))
        # twiddle slabs of the FIRST passes (plain, unswizzled arrays): lane reads of 8 / 4 / 2 / 1 consecutive words
        ident = lambda g: g
        for bp, cnt, kind in ((0, 4, "r128"), (4, 4, "r128"), (5, 4, "r128"), (2, 4, "r128")):
            bad = max(bad, report(f"slab read bp={bp} ({cnt} words)", lambda tid, bp=bp, cnt=cnt: (row0(tid, bp) >> (bp + (1 if bp else 2))) & ~3, kind, ident))
    return bad


def main():
    for label, f in (("xor swizzle (fft_pipe.cuh)", swz), ("no swizzle", swz_none)):
        print(label)
        print("  worst over all patterns:", worst_case(f, verbose=True))
    # involution / bijection check
    assert sorted(swz(g) for g in range(2048)) == list(range(2048))
    assert all(swz(swz(g)) == g for g in range(2048))

if __name__ == "__main__":
    main()
