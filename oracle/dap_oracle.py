"""CPU restatement of the reference's index-row construction (TEST INFRASTRUCTURE).

Restates /root/reference/src/dap_to_bed.py:55-134 (class print_dap_as_mem_bed with --mem,
optionally --order and --overlap; the --ms branch of the reference calls an undefined name and
cannot run).  Written as whole-array NumPy, not as the reference's row loop:

  * a DAP row holds, per non-pivot genome column, the matching statistic at that pivot position;
    --order sorts each row descending first                                   (:85-91)
  * column c has a MEM starting at row p iff p opens its record or lcp[p-1][c] <= lcp[p][c]
                                                                              (:119-125, :130-131)
  * --overlap prints, for each MEM, its overlap with the previous MEM of the same column in the
    same record (start = this MEM's start; end = min of the two ends) if that end >= start
                                                                              (:93-107)
  * after the last row of a record, the pseudo-MEM (len, 2*len) is pushed through the same
    printer for every column                                                  (:126-128, :133-134)
Pinned against tests/golden/dap/*.bed, which tools/make_golden.py made by running the reference.
"""
import numpy as np


def read_fai(path):
    names, lens = [], []
    for line in open(path):
        if line.strip():
            f = line.split()
            names.append(f[0])
            lens.append(int(f[1]))
    return names, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def read_dap(path):
    """(positions [P], lcp [P, C]) from the `nl`-numbered, space separated text of index.sh:83"""
    a = np.loadtxt(path, dtype=np.int64, ndmin=2)
    return a[:, 0], a[:, 1:]


def dap_rows(lcp, rec_begin, overlap, order):
    """rows (record index, start, end, annot) in the reference's print order"""
    lcp = np.asarray(lcp, np.int64)
    if order:
        lcp = -np.sort(-lcp, axis=1)
    P, C = lcp.shape
    pos = np.arange(P)
    rec = np.searchsorted(rec_begin, pos, side="right") - 1
    rel = pos - rec_begin[rec]
    first = rel == 0
    last_of_rec = np.append(rec[1:] != rec[:-1], True) if P else np.zeros(0, bool)
    flag = np.ones((P, C), bool)
    flag[1:] = lcp[:-1] <= lcp[1:]
    flag[first] = True
    end = rel[:, None] + lcp
    # row index of the latest MEM start at or before p, per column (record openers flag every column,
    # so this never reaches back into an earlier record for rows after the opener)
    latest = np.maximum.accumulate(np.where(flag, pos[:, None], -1), axis=0)
    cols = np.arange(C)[None, :]
    out = []                                           # (sort key row, slot, column, rec, start, end)
    if overlap:
        prev = np.vstack([np.full((1, C), -1), latest[:-1]])
        has_prev = (prev >= 0) & ~first[:, None]
        prev_end = end[np.maximum(prev, 0), cols]
        ov_end = np.minimum(prev_end, end)
        emit = flag & has_prev & (ov_end >= rel[:, None])
        p_i, c_i = np.nonzero(emit)
        out.append(np.stack([p_i, np.zeros_like(p_i), c_i, rec[p_i], rel[p_i], ov_end[p_i, c_i]], 1))
        for p in np.nonzero(last_of_rec)[0]:
            L = rec_begin[rec[p] + 1] - rec_begin[rec[p]]
            e = np.minimum(end[latest[p], np.arange(C)], 2 * L)
            c_i = np.nonzero(e >= L)[0]
            out.append(np.stack([np.full_like(c_i, p), np.ones_like(c_i), c_i, np.full_like(c_i, rec[p]),
                                 np.full_like(c_i, L), e[c_i]], 1))
    else:
        p_i, c_i = np.nonzero(flag)
        out.append(np.stack([p_i, np.zeros_like(p_i), c_i, rec[p_i], rel[p_i], end[p_i, c_i]], 1))
        for p in np.nonzero(last_of_rec)[0]:
            L = rec_begin[rec[p] + 1] - rec_begin[rec[p]]
            c_i = np.arange(C)
            out.append(np.stack([np.full_like(c_i, p), np.ones_like(c_i), c_i, np.full_like(c_i, rec[p]),
                                 np.full_like(c_i, L), np.full_like(c_i, 2 * L)], 1))
    rows = np.concatenate(out) if out else np.zeros((0, 6), np.int64)
    rows = rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))]
    return rows[:, 3], rows[:, 4], rows[:, 5], rows[:, 2] + 1


def bed_text(names, rec, start, end, annot):
    return "".join(f"{names[r]}\t{s}\t{e}\t{a}\n" for r, s, e, a in zip(rec, start, end, annot))
