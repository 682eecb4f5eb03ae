#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Runs only in the authoring container (needs /root/reference).  Nothing here is
imported by the product, the tests, smoke() or bench.py: only its OUTPUT (data
files) is committed and travels to the GPU box.

How the reference is executed
-----------------------------
* ``/root/reference/src/memo_query.py`` does ``from numba import jit`` (line 17)
  and numba is not installed here, so a 3-line stand-in module whose ``jit`` is
  the identity decorator is put in ``sys.modules`` first.  The decorated loop
  (memo_query.py:57-63) is plain NumPy slice assignment, which means the same
  thing interpreted as it does under nopython Numba for in-range indices.
* ``sys.dont_write_bytecode`` keeps ``__pycache__`` out of the read-only tree.
* The example index cannot be built with MONI (absent), so matching statistics
  of the 26-bp pivot against the other four example genomes (+ reverse
  complements, the text index.sh:63-65 builds) are brute-forced here, written
  in the dap.txt format of index.sh:83, and then pushed through the
  reference's OWN dap_to_bed.py and parquet_compress_bed.py.

Fixture layout (tests/golden/)
------------------------------
  <index>.parquet          index files written by the reference's compressor
  manifest.json            one entry per query case: args + sha256 + paths
  cases/<name>.out         exact bytes the reference wrote to ``-o`` (kept when
                           <= 4 KiB; every case keeps the sha256 of those bytes)
  cases/<name>.npz         rows = what reference filter_pq returned (uint64 [M,3]),
                           vec  = conservation vector (int64 [L]) or
                           bits = membership matrix, np.packbits(bool[L,N], axis=1)
"""
import contextlib
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")

_numba = types.ModuleType("numba")
_numba.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
sys.modules["numba"] = _numba
sys.path.insert(0, os.path.join(REF, "src"))
import memo_query as ref_q          # noqa: E402  (the reference hot path)
import dap_to_bed as ref_d          # noqa: E402
import parquet_compress_bed as ref_p  # noqa: E402


# --------------------------------------------------------------------------
# example/ index without MONI
# --------------------------------------------------------------------------
def read_fasta(path):
    recs, name, seq = [], None, []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:].split()[0], []
        elif line:
            seq.append(line.upper())
    recs.append((name, "".join(seq)))
    return recs


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def matching_statistics(pivot, texts):
    """MS[i] = length of the longest prefix of pivot[i:] found in any text."""
    out = []
    for i in range(len(pivot)):
        best = 0
        for n in range(1, len(pivot) - i + 1):
            if any(pivot[i:i + n] in t for t in texts):
                best = n
            else:
                break
        out.append(best)
    return out


def build_example(work):
    fas = [os.path.join(REF, "example", f"ref_{i}.fa") for i in range(1, 6)]
    (pname, pseq), = read_fasta(fas[0])
    cols = []
    for fa in fas[1:]:
        texts = []
        for _, s in read_fasta(fa):
            texts += [s, revcomp(s)]
        cols.append(matching_statistics(pseq, texts))
    dap = os.path.join(work, "dap.txt")
    with open(dap, "w") as fh:
        for i in range(len(pseq)):
            fh.write(" ".join(map(str, [i] + [c[i] for c in cols])) + "\n")
    fai = os.path.join(work, "ref_1.fa.fai")
    with open(fai, "w") as fh:      # samtools faidx format: name len offset linebases linewidth
        fh.write(f"{pname}\t{len(pseq)}\t{len(pname) + 2}\t{len(pseq)}\t{len(pseq) + 1}\n")
    out = {}
    for flavour, order in (("example_cons", True), ("example_memb", False)):
        bed = os.path.join(work, flavour + ".bed")
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_d.print_dap_as_mem_bed(ref_d.read_file(dap), ref_d.parse_fai(fai),
                                       True, order).dap_to_mem()
        open(bed, "w").write(buf.getvalue())
        pq = os.path.join(GOLD, flavour + ".parquet")
        ref_p.compress_bed(bed, pq)
        out[flavour] = pq
    shutil.copy(dap, os.path.join(GOLD, "example_dap.txt"))
    return out, len(pseq)


# --------------------------------------------------------------------------
# random indexes (BED text -> reference compressor)
# --------------------------------------------------------------------------
def random_bed(rng, chroms, n_docs, kmax, long_overlap=False):
    """chroms: list of (name, length, rows).  Row = overlap of two consecutive
    MEMs as dap_to_bed.py:93-107 emits them: end >= start, annot in 1..n_docs-1,
    rows start-sorted inside a chromosome, chr-end sentinel rows at the end."""
    lines = []
    for name, length, rows in chroms:
        starts = np.sort(rng.integers(1, length, size=rows))
        for s in starts:
            ln = int(rng.integers(0, 2 * kmax)) if not long_overlap else int(rng.integers(0, 4 * kmax))
            a = int(rng.integers(1, n_docs))
            lines.append(f"{name}\t{int(s)}\t{int(s) + ln}\t{a}")
        for a in range(1, n_docs):
            lines.append(f"{name}\t{length}\t{length}\t{a}")
    return "\n".join(lines) + "\n"


def make_random_index(work, name, text):
    bed = os.path.join(work, name + ".bed")
    open(bed, "w").write(text)
    pq = os.path.join(GOLD, name + ".parquet")
    ref_p.compress_bed(bed, pq)
    return pq


# --------------------------------------------------------------------------
# run the reference on one query
# --------------------------------------------------------------------------
def run_case(manifest, name, pq, k, n, region, memb, work):
    out = os.path.join(work, name + ".out")
    entry = dict(name=name, index=os.path.basename(pq), k=k, n=n, region=region,
                 membership=bool(memb))
    record, se = region.split(":")
    qs, qe = map(int, se.split("-"))
    try:
        rows = ref_q.filter_pq(pq, record, qs, qe + k)
        mem_arr, rec = ref_q.memo_init(rows, k, qs, qe, n, memb)
        rec = ref_q.memo_query(mem_arr, rec, memb)
        ref_q.print_res(rec, out, memb)
    except Exception as exc:                      # e.g. -n too small -> IndexError
        entry["raises"] = type(exc).__name__
        manifest.append(entry)
        return
    import gc
    gc.collect()                                   # print_res never closes its handle
    data = open(out, "rb").read()
    if len(data) <= 4096:                          # big outputs are pinned by sha256 only
        open(os.path.join(GOLD, "cases", name + ".out"), "wb").write(data)
        entry["out"] = "cases/" + name + ".out"
    arrs = dict(rows=np.asarray(rows, dtype=np.uint64))
    if memb:
        arrs["bits"] = np.packbits(rec.astype(np.uint8), axis=1)   # [L, ceil(N/8)], MSB first
    else:
        arrs["vec"] = np.argmax(rec, axis=1).astype(np.int64)
    np.savez_compressed(os.path.join(GOLD, "cases", name + ".npz"), **arrs)
    entry["npz"] = "cases/" + name + ".npz"
    entry["sha256"] = hashlib.sha256(data).hexdigest()
    manifest.append(entry)


def main():
    if os.path.isdir(GOLD):
        shutil.rmtree(GOLD)
    os.makedirs(os.path.join(GOLD, "cases"))
    work = tempfile.mkdtemp(prefix="memo_golden_")
    manifest = []
    ex, plen = build_example(work)

    # --- example index: the README walkthrough + BASELINE config 1 + edge cases
    c, m = ex["example_cons"], ex["example_memb"]
    run_case(manifest, "ex_cons_k3_0_20", c, 3, 5, "ref_1:0-20", False, work)     # example/README.md:19-27
    run_case(manifest, "ex_cons_k31_full", c, 31, 5, f"ref_1:0-{plen}", False, work)  # BASELINE config 1
    run_case(manifest, "ex_cons_k4_3_30", c, 4, 5, "ref_1:3-30", False, work)     # runs past chr end
    run_case(manifest, "ex_cons_k1_full", c, 1, 5, f"ref_1:0-{plen}", False, work)
    run_case(manifest, "ex_cons_k2_full", c, 2, 5, f"ref_1:0-{plen}", False, work)
    run_case(manifest, "ex_cons_k5_7_19", c, 5, 5, "ref_1:7-19", False, work)
    run_case(manifest, "ex_cons_unknown_chr", c, 3, 5, "nochr:0-5", False, work)
    run_case(manifest, "ex_cons_empty_window", c, 3, 5, "ref_1:5-5", False, work)
    run_case(manifest, "ex_cons_n_too_small", c, 3, 2, "ref_1:0-20", False, work)  # IndexError
    run_case(manifest, "ex_cons_n_larger", c, 3, 9, "ref_1:0-20", False, work)
    run_case(manifest, "ex_memb_k3_0_20", m, 3, 5, "ref_1:0-20", True, work)
    run_case(manifest, "ex_memb_k5_2_24", m, 5, 5, "ref_1:2-24", True, work)
    run_case(manifest, "ex_memb_k31_full", m, 31, 5, f"ref_1:0-{plen}", True, work)
    run_case(manifest, "ex_memb_k4_3_30", m, 4, 5, "ref_1:3-30", True, work)
    run_case(manifest, "ex_memb_empty_window", m, 3, 5, "ref_1:5-5", True, work)
    run_case(manifest, "ex_memb_n_too_small", m, 3, 2, "ref_1:0-20", True, work)

    # --- random indexes: several N, two chromosomes, many k / windows
    rng = np.random.default_rng(0x4D454D4F)
    specs = [("rnd_n4", 4, [("chrA", 300, 500), ("chrB", 150, 200)], 8),
             ("rnd_n8", 8, [("chr1", 2000, 6000)], 40),
             ("rnd_n40", 40, [("chr1", 5000, 60000), ("chr2", 700, 900)], 110),
             ("rnd_n70_sparse", 70, [("chr1", 20000, 900)], 110),
             ("rnd_n130", 130, [("chr1", 3000, 30000)], 40)]
    for name, n, chroms, kmax in specs:
        pq = make_random_index(work, name, random_bed(rng, chroms, n, kmax))
        for memb in (False, True):
            tag = "memb" if memb else "cons"
            for k in sorted({1, 2, 3, 5, 21, 31, 64, 65, 101} & set(range(1, kmax * 3))):
                for ci, (cname, clen, _) in enumerate(chroms):
                    wins = [(0, min(clen, 700)),
                            (int(rng.integers(0, clen // 2)), int(rng.integers(clen // 2, clen + 50)))]
                    if k in (3, 31):
                        wins.append((max(0, clen - 40), clen + 37))
                        wins.append((int(clen // 3), int(clen // 3) + 1))
                    for wi, (qs, qe) in enumerate(wins):
                        if (qe - qs) * n > 120_000:     # keep fixtures small
                            qe = qs + 120_000 // n
                        run_case(manifest, f"{name}_{tag}_k{k}_c{ci}w{wi}", pq, k, n,
                                 f"{cname}:{qs}-{qe}", memb, work)
    # --- rows with end < start: never written by dap_to_bed.py (:93-98) but legal input to
    # memo_query.py, where they shade [end-(k-1), start) like any other row -- possibly far more
    # than k-1 positions
    rng2 = np.random.default_rng(0xBEEF)
    lines = []
    starts = np.sort(rng2.integers(1, 3000, size=4000))
    for st_ in starts:
        st_ = int(st_)
        if rng2.random() < 0.08:
            en = st_ - int(rng2.integers(1, 400))
        else:
            en = st_ + int(rng2.integers(0, 70))
        lines.append(f"chrN\t{st_}\t{max(en, 0)}\t{int(rng2.integers(1, 12))}")
    pqn = make_random_index(work, "rnd_negoverlap", "\n".join(lines) + "\n")
    for memb in (False, True):
        tag = "memb" if memb else "cons"
        for k in (1, 2, 31, 101):
            for wi, (qs, qe) in enumerate(((0, 3100), (700, 1900), (2900, 3300))):
                run_case(manifest, f"rnd_negoverlap_{tag}_k{k}_w{wi}", pqn, k, 12, f"chrN:{qs}-{qe}", memb, work)

    # --- index construction (dap_to_bed.py:116-134): DAP rows -> MEMs / MEM-overlap BED rows
    os.makedirs(os.path.join(GOLD, "dap"))
    drng = np.random.default_rng(0xDA9)
    dap_cases = []

    def synth_dap(records, cols, jitter):
        """matching-statistic-like columns: lcp[i+1] >= lcp[i] - 1, random restarts"""
        rows, fai, pos = [], [], 0
        for name, length in records:
            fai.append(f"{name}\t{length}\t0\t60\t61")
            cur = drng.integers(1, 12, cols)
            for i in range(length):
                cur = np.maximum(cur - 1, 0)
                bump = drng.random(cols) < jitter
                cur = np.where(bump, drng.integers(0, 25, cols), cur)
                cur = np.minimum(cur, length - i)
                rows.append(" ".join(map(str, [pos] + cur.tolist())))
                pos += 1
        return "\n".join(rows) + "\n", "\n".join(fai) + "\n"

    dap_specs = [("dap_one", [("chrA", 60)], 3, 0.2), ("dap_multi", [("r1", 40), ("r2", 1), ("r3", 77)], 5, 0.15),
                 ("dap_wide", [("c1", 300), ("c2", 150)], 70, 0.1), ("dap_dense", [("x", 500)], 12, 0.5)]
    example_dap = open(os.path.join(GOLD, "example_dap.txt")).read()
    dap_inputs = [(n,) + synth_dap(r, c, j) for n, r, c, j in dap_specs] + \
        [("dap_example", example_dap, "ref_1\t26\t7\t26\t27\n")]
    for name, dap_text, fai_text in dap_inputs:
        dpath, fpath = os.path.join(GOLD, "dap", name + ".dap.txt"), os.path.join(GOLD, "dap", name + ".fa.fai")
        open(dpath, "w").write(dap_text)
        open(fpath, "w").write(fai_text)
        for overlap in (True, False):
            for order in (True, False):
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    ref_d.print_dap_as_mem_bed(ref_d.read_file(dpath), ref_d.parse_fai(fpath), overlap, order).dap_to_mem()
                tag = f"{name}_{'ovl' if overlap else 'mem'}_{'order' if order else 'doc'}"
                open(os.path.join(GOLD, "dap", tag + ".bed"), "w").write(buf.getvalue())
                dap_cases.append(dict(name=tag, dap=name + ".dap.txt", fai=name + ".fa.fai", overlap=overlap, order=order,
                                      sha256=hashlib.sha256(buf.getvalue().encode()).hexdigest()))
    json.dump(dap_cases, open(os.path.join(GOLD, "dap", "manifest.json"), "w"), indent=0)

    # --- `memo view` binning (plot_conservation.py:46-65): per-bin composition of a conservation
    # vector; plotnine is absent here and only needed for drawing, so it is stubbed for the import
    pn = types.ModuleType("plotnine")
    for nm in ("ggplot aes theme themes element_blank element_line element_text geom_bar ggtitle xlab ylab "
               "scale_y_continuous scale_fill_gradient").split():
        setattr(pn, nm, None)
    po = types.ModuleType("plotnine.options")
    po.figure_size = None
    sys.modules["plotnine"], sys.modules["plotnine.options"] = pn, po
    import plot_conservation as ref_v
    os.makedirs(os.path.join(GOLD, "view"))
    vrng = np.random.default_rng(7)
    view_cases = []
    for vi, (npos, n_docs, n_bins) in enumerate([(20, 5, 4), (1000, 9, 7), (5000, 40, 500), (333, 3, 1), (7, 5, 10), (4096, 100, 64)]):
        vec = vrng.integers(1, n_docs + 1, npos)
        vec[vrng.random(npos) < 0.5] = n_docs
        path = os.path.join(work, f"view{vi}.txt")
        open(path, "w").write("".join(f"{v}\n" for v in vec))
        try:
            df = ref_v.preprocess_data(path, n_docs, n_bins)
        except ZeroDivisionError:                 # an empty bin (more bins than positions)
            np.savez_compressed(os.path.join(GOLD, "view", f"view{vi}.npz"), vec=vec.astype(np.uint16))
            view_cases.append(dict(name=f"view{vi}", n_docs=n_docs, n_bins=n_bins, positions=npos,
                                   raises="ZeroDivisionError"))
            continue
        np.savez_compressed(os.path.join(GOLD, "view", f"view{vi}.npz"), vec=vec.astype(np.uint16),
                            bin=df["bin"].to_numpy(np.int64), genomes=df["No. Genomes"].to_numpy(np.float64),
                            value=df["value"].to_numpy(np.float64))
        view_cases.append(dict(name=f"view{vi}", n_docs=n_docs, n_bins=n_bins, positions=npos))
    json.dump(view_cases, open(os.path.join(GOLD, "view", "manifest.json"), "w"))

    # --- stdout of the reference's bash front end (usage banners; exit status 0 in all three)
    import subprocess
    os.makedirs(os.path.join(GOLD, "cli"))
    for fname, argv in (("memo_usage.txt", []), ("memo_query_usage.txt", ["query"]), ("memo_bogus.txt", ["bogus"])):
        r = subprocess.run(["bash", os.path.join(REF, "src", "memo")] + argv, capture_output=True)
        assert r.returncode == 0
        open(os.path.join(GOLD, "cli", fname), "wb").write(r.stdout)
    json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=0)
    shutil.rmtree(work)
    tot = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(GOLD) for f in fs)
    print(f"{len(manifest)} cases, {tot / 1e6:.2f} MB in {GOLD}")


if __name__ == "__main__":
    main()
