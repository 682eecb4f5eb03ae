"""End to end on sequences: genomes -> matching statistics (brute force) -> DAP -> index rows on the
GPU -> Parquet -> `memo query` on the GPU, checked against what the result MEANS (README.md:3 of the
reference): conservation(p) = number of genomes (pivot included) that contain the k-mer starting at
pivot position p; membership(p, g) = genome g contains it.  The ground truth is an independent
k-mer lookup in the genome texts (each genome plus its reverse complement, as index.sh:63-65 builds
them), so this exercises every stage together rather than stage-by-stage parity."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def _mutate(rng, seq, rate):
    out = []
    for ch in seq:
        r = rng.random()
        if r < rate / 3:
            continue                                   # deletion
        if r < 2 * rate / 3:
            out.append("ACGT"[rng.integers(4)])        # substitution
            continue
        out.append(ch)
        if r > 1 - rate / 3:
            out.append("ACGT"[rng.integers(4)])        # insertion
    return "".join(out)


def _matching_statistics(pivot, text):
    """MS[i] = longest prefix of pivot[i:] occurring in text (binary search on the length)"""
    n, ms = len(pivot), np.zeros(len(pivot), np.int64)
    for i in range(n):
        lo, hi = 0, n - i
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if pivot[i:i + mid] in text:
                lo = mid
            else:
                hi = mid - 1
        ms[i] = lo
    return ms


@pytest.fixture(scope="module")
def pangenome(tmp_path_factory):
    rng = np.random.default_rng(2024)
    work = tmp_path_factory.mktemp("pan")
    records = [("chrA", 1500), ("chrB", 700)]
    pivot = [(name, "".join("ACGT"[x] for x in rng.integers(0, 4, n))) for name, n in records]
    genomes = []                                        # text of every non-pivot genome: records + RC
    for g in range(6):
        recs = [_mutate(rng, seq, 0.02 * (g + 1)) for _, seq in pivot]
        if g == 3:
            recs[1] = ""                                # one genome lacks chrB altogether
        genomes.append("$".join(r + "$" + _revcomp(r) for r in recs))
    dap = os.path.join(work, "dap.txt")
    fai = os.path.join(work, "pivot.fa.fai")
    cols = [np.concatenate([_matching_statistics(seq, text) for _, seq in pivot]) for text in genomes]
    with open(dap, "w") as fh:
        for i in range(sum(n for _, n in records)):
            fh.write(" ".join(map(str, [i] + [int(c[i]) for c in cols])) + "\n")
    with open(fai, "w") as fh:
        for name, n in records:
            fh.write(f"{name}\t{n}\t0\t60\t61\n")
    return dict(work=work, pivot=pivot, genomes=genomes, dap=dap, fai=fai, n_docs=len(genomes) + 1)


@pytest.mark.parametrize("k", [4, 12, 31, 60])
def test_conservation_counts_genomes_containing_the_kmer(pangenome, k):
    from memo_amd import dap_to_bed as D, memo_query as mq, emit_conservation
    P = pangenome
    idx = os.path.join(P["work"], "cons.parquet")
    if not os.path.exists(idx):
        D.dap_to_parquet(P["dap"], P["fai"], idx, order=True)
    for name, seq in P["pivot"]:
        L = len(seq)
        truth = np.ones(L, np.int64)                    # the pivot itself
        for p in range(L - k + 1):
            kmer = seq[p:p + k]
            truth[p] += sum(kmer in text for text in P["genomes"])
        out = os.path.join(P["work"], f"c_{name}_{k}.txt")
        mq.main(mq.parse_arguments(["-b", idx, "-k", str(k), "-n", str(P["n_docs"]), "-r", f"{name}:0-{L}", "-o", out]))
        got = np.loadtxt(out, dtype=np.int64)
        assert np.array_equal(got, truth), (name, k, np.nonzero(got != truth)[0][:10])


@pytest.mark.parametrize("k", [4, 12, 31])
def test_membership_bits_are_kmer_presence(pangenome, k):
    from memo_amd import dap_to_bed as D, memo_query as mq
    P = pangenome
    idx = os.path.join(P["work"], "memb.parquet")
    if not os.path.exists(idx):
        D.dap_to_parquet(P["dap"], P["fai"], idx, order=False)
    for name, seq in P["pivot"]:
        L = len(seq)
        truth = np.zeros((L, P["n_docs"]), np.int64)
        truth[:, 0] = 1
        for p in range(L - k + 1):
            kmer = seq[p:p + k]
            truth[p, 1:] = [kmer in text for text in P["genomes"]]
        out = os.path.join(P["work"], f"m_{name}_{k}.txt")
        mq.main(mq.parse_arguments(["-m", "-b", idx, "-k", str(k), "-n", str(P["n_docs"]), "-r", f"{name}:100-{L}", "-o", out]))
        got = np.loadtxt(out, dtype=np.int64, ndmin=2)
        assert np.array_equal(got, truth[100:]), (name, k)
