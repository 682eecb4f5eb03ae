#!/usr/bin/env python3
"""Round 4: what the order of a dense k-class view's rows costs in LDS bank conflicts, by the cycle model of
profiles/r04_lds_atomics.txt (a half-wave's 32 ds_min take max(2, lanes on the fullest of 32 banks) cycles; k lanes on ONE cell
2k - 1).  Rows of BASELINE config 3 (five per position, overlap uniform in [0, 60)), k = 31: the view keeps overlap < 30.

  current      the 4-byte rows' interleaved order (memo_interleave.hip, mode 2), filtered
  coloured     round 4's colour_view_kernel (rows taken by A mod 32, one of five places chosen greedily, laid down in that order);
               round 5 (memo_view.hip: view_place_bucket) takes the rows in the order they come: 5.76 against 5.53 by this model

Anywhere (numpy only).  Prints cycles per row instruction and half-wave: first block, second block; at four alignments of the tile's
first group.
"""
import numpy as np

KM1 = 30
rng = np.random.default_rng(1)


def keys(s, ov):
    n = KM1 - ov
    lev = np.floor(np.log2(n)).astype(int)
    return (s - n) & 31, (s - (1 << lev)) & 31, lev


def halfwave(addr):
    worst = 0
    bank = addr & 31
    for b in np.unique(bank):
        _, c = np.unique(addr[bank == b], return_counts=True)
        worst = max(worst, int(np.sum(np.where(c == 1, 1, 2 * c - 1))))
    return max(2, worst)


def cost(s, ov, align=0):
    n = KM1 - ov
    lev = np.floor(np.log2(n)).astype(int)
    first = lev * 1024 + ((s - n) & 1023)
    second = lev * 1024 + ((s - (1 << lev)) & 1023)
    c1 = c2 = cnt = 0
    for g0 in range(align, len(s) // 5 - 32, 32):
        for i in range(5):
            idx = 5 * np.arange(g0, g0 + 32) + i
            c1 += halfwave(first[idx])
            c2 += halfwave(second[idx])
            cnt += 1
    return round(c1 / cnt, 2), round(c2 / cnt, 2)


def interleaved(s, ov):
    out = []
    for b in range(s.max() // 32 + 1):
        m = np.nonzero((s >> 5) == b)[0]
        p = np.lexsort((ov[m] >> 5, ov[m] & 31, s[m]))
        m = m[p]
        chunks = []
        for st in np.unique(s[m]):
            idx = m[s[m] == st]
            chunks += [(q // 4, st, idx[q:q + 4]) for q in range(0, len(idx), 4)]
        chunks.sort(key=lambda c: (c[0], c[1]))
        out.append(np.concatenate([c[2] for c in chunks]))
    return np.concatenate(out)


def coloured(s, ov):
    out_s, out_o = np.empty_like(s), np.empty_like(ov)
    nb = s.max() // 32 + 1
    bounds = np.searchsorted(s >> 5, np.arange(nb + 1))
    for b in range(nb):
        r0, r1 = bounds[b], bounds[b + 1]
        a, o = s[r0:r1], ov[r0:r1]
        A, B, _ = keys(a, o)
        pos = np.arange(r0, r1)
        room = [int(np.sum(pos % 5 == c)) for c in range(5)]
        uA, uA2, uB, uB2, load = [0] * 5, [0] * 5, [0] * 5, [0] * 5, [0] * 5
        members = [[] for _ in range(5)]
        for r in np.argsort(A, kind="stable"):
            best, bestp = 0, None
            for c in range(5):
                if load[c] >= room[c]:
                    continue
                p = 4 * (uA[c] >> A[r] & 1) + 16 * (uA2[c] >> A[r] & 1) + 5 * (uB[c] >> B[r] & 1) + 16 * (uB2[c] >> B[r] & 1)
                p = (p << 8) + load[c]
                if bestp is None or p < bestp:
                    best, bestp = c, p
            c = best
            uA2[c] |= uA[c] & (1 << A[r])
            uB2[c] |= uB[c] & (1 << B[r])
            uA[c] |= 1 << A[r]
            uB[c] |= 1 << B[r]
            load[c] += 1
            members[c].append(r)
        for c in range(5):
            p = pos[pos % 5 == c]
            out_s[p], out_o[p] = a[members[c]], o[members[c]]
    return out_s, out_o


def main():
    s = np.repeat(np.arange(32 * 600), 5)
    ov = rng.integers(0, 60, size=s.size)
    p = interleaved(s, ov)
    s, ov = s[p], ov[p]
    keep = ov < KM1
    s, ov = s[keep], ov[keep]
    order = np.argsort(s >> 5, kind="stable")          # (bucket order; the order inside a bucket is the filtered interleave)
    s, ov = s[order], ov[order]
    print("rows per position in the view: %.2f" % (len(s) / (32 * 600)))
    for name, (a, o) in (("current", (s, ov)), ("coloured", coloured(s, ov))):
        print("%-9s" % name, *(cost(a, o, g) for g in (0, 8, 16, 24)))


if __name__ == "__main__":
    main()
