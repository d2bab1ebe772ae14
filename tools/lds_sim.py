"""LDS bank-conflict model for the transposes of tfhe_amd/csrc/br_core.h (rules: MI355X_MICROARCH.md, LDS table).
Prints, per access pattern, LDS-array cycles per wave instruction (ideal: 4 for ds_read_b128, 8 for ds_write_b128)."""
import itertools

RD128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
         list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
         list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
         list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
WR128 = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def br(v, n):
    return int(format(v, "0%db" % n)[::-1], 2)


def gray(u):
    return u ^ (u >> 1)


def cycles(slots, groups, nslots):
    """slots[lane] = 16-byte slot index; each lane touches 4 consecutive banks of a (4*nslots)-bank LDS row"""
    tot = 0
    for g in groups:
        per = {}
        for l in g:
            per.setdefault(slots[l] % nslots, set()).add(slots[l])
        tot += max(len(v) for v in per.values())
    return tot


def patterns():
    P = {}
    P["A  x=br6(l)<<3|r"] = lambda l, r: (br(l, 6) << 3) | r
    P["B  x=(l>>3)<<6|r<<3|(l&7)"] = lambda l, r: ((l >> 3) << 6) | (r << 3) | (l & 7)
    P["C  x=r<<6|l"] = lambda l, r: (r << 6) | l
    return P


def gray_pat(l, rho):
    m = (br(rho, 4) << 6) | l
    return gray(m) if m < 512 else gray(1023 - m)


def report(swz):
    P = patterns()
    res = {}
    for name, f in P.items():
        w = sum(cycles([swz(f(l, r)) for l in range(64)], WR128, 8) for r in range(8)) / 8
        rd = sum(cycles([swz(f(l, r)) for l in range(64)], RD128, 16) for r in range(8)) / 8
        res[name] = (w, rd)
    rg = sum(cycles([swz(gray_pat(l, rho)) for l in range(64)], RD128, 16) for rho in range(16)) / 16
    res["G  gray read"] = (None, rg)
    return res


def swz_current(x):
    return x ^ ((x >> 3) & 7) ^ (((x >> 6) & 1) << 3)


if __name__ == "__main__":
    print("current swizzle: (write cycles ideal 8, read cycles ideal 4)")
    for k, v in report(swz_current).items():
        print("  %-32s write %s read %s" % (k, v[0], v[1]))
    print("identity:")
    for k, v in report(lambda x: x).items():
        print("  %-32s write %s read %s" % (k, v[0], v[1]))


def swz_new(x):
    return x ^ ((x >> 6) & 7) ^ (((x >> 7) & 1) << 3)


def igray(u):
    u ^= u >> 1; u ^= u >> 2; u ^= u >> 4; u ^= u >> 8
    return u


def tw_idx(size, j):
    return size // 2 - 4 + j


def twiddle_report(perm=lambda size, j: tw_idx(size, j)):
    out = {}
    def rd(name, f):
        out[name] = cycles([f(l) for l in range(64)], RD128, 16)
    for lv, size in ((3, 64),):
        rd("fwdB L3", lambda l: perm(64, 2 * igray(l & 7) + 1))
    rd("fwdB L4a", lambda l: perm(128, 2 * igray(l & 7) + 1))
    rd("fwdB L4b", lambda l: perm(128, 2 * (8 | (igray(l & 7) ^ 7)) + 1))
    for k, hi in enumerate((0, 1, 3, 2)):
        rd("fwdB L5%d" % k, lambda l, hi=hi, k=k: perm(256, 2 * ((hi << 3) | (igray(l & 7) ^ (7 if k in (1, 2) else 0))) + 1))
    rd("fwdC L6", lambda l: perm(512, 2 * igray(l) + 1))
    rd("fwdC L7a", lambda l: perm(1024, 2 * igray(l) + 1))
    rd("fwdC L7b", lambda l: perm(1024, 2 * (64 | (igray(l) ^ 63)) + 1))
    for k, hi in enumerate((0, 1, 3, 2)):
        rd("fwdC L8%d" % k, lambda l, hi=hi, k=k: perm(2048, 2 * ((hi << 6) | (igray(l) ^ (63 if k in (1, 2) else 0))) + 1))
    rd("invB L4", lambda l: perm(32, (1 << 3) | (l & 7)))
    rd("invB L5", lambda l: perm(64, (3 << 3) | (l & 7)))
    rd("invB L6", lambda l: perm(128, (5 << 3) | (l & 7)))
    base = lambda l: ((l >> 3) << 4) | (l & 7)
    rd("invC L7", lambda l: perm(256, base(l) | 8))
    rd("invC L8", lambda l: perm(512, (1 << 7) | base(l) | 8))
    rd("invC L9", lambda l: perm(1024, (3 << 7) | base(l)))
    rd("inv final", lambda l: perm(2048, (5 << 7) | base(l) | 8))
    return out


if __name__ == "__main__":
    print("new swizzle:")
    for k, v in report(swz_new).items():
        print("  %-32s write %s read %s" % (k, v[0], v[1]))
    print("twiddle reads (ideal 4):", twiddle_report())


def perm_oddfirst(size, j):
    base = size // 2 - 4
    return base + ((j >> 1) if (j & 1) else size // 4 + (j >> 1))


if __name__ == "__main__":
    r = twiddle_report(perm_oddfirst)
    print("odd-first table:", r, sum(r.values()))
    best = None
    # try xor-swizzling the in-table offset: pos = base + (o ^ ((o >> a) & m))
    for a in range(1, 8):
        for m in (1, 3, 7, 15):
            for b in range(0, 8):
                for m2 in (0, 1, 3, 7, 8, 15):
                    def perm(size, j, a=a, m=m, b=b, m2=m2):
                        base = size // 2 - 4
                        o = (j >> 1) if (j & 1) else size // 4 + (j >> 1)
                        o2 = o ^ ((o >> a) & m) ^ (((o >> b) & m2) if m2 else 0)
                        if o2 >= size // 2: return base + o
                        return base + o2
                    # must be a permutation per size
                    ok = all(len({perm(s, j) for j in range(s // 2)}) == s // 2 for s in (32, 64, 128, 256, 512, 1024, 2048))
                    if not ok: continue
                    rr = twiddle_report(perm); tot = sum(rr.values())
                    if best is None or tot < best[0]: best = (tot, a, m, b, m2, rr)
    print("best", best)
